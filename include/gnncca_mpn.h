/*
 * gnncca_mpn.h -- C ABI of the MI355X-native GNN-CCA message-passing path (libgnncca_mpn.so).
 *
 * The reference (vpulab/GNN-CCA) is pure Python and has NO native/FFI layer: the
 * interface a caller binds is the Python module surface `models.mpn.MOTMPNet` (models/mpn.py:144-299).
 * This header is therefore the boundary *below* that module: every entry point states which piece of
 * the reference's Python it replaces.  The Python mirror of MOTMPNet (gnn-cca_amd/mpn.py) is the only
 * intended caller and binds these symbols with ctypes.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `gnncca_stream_t` is a hipStream_t passed as void*.
 *   - all device pointers are raw HBM addresses owned by the caller; nothing is allocated or freed here.
 *   - every function returns a gnncca_status (0 = ok).  Nothing synchronises the stream except
 *     gnncca_read_graph_flags().  All kernels are enqueued on the given stream.
 *   - row-major fp32 everywhere at the boundary; `edge_index` is int64 [2][E] exactly as
 *     torch_geometric hands it to MOTMPNet.forward (models/mpn.py:266).
 */
#ifndef GNNCCA_MPN_H
#define GNNCCA_MPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): gnncca_frames_io gained `counters_len` -- gnncca_frames_forward writes [3 N + 1 + G] int32 through `counters` (round 5 grew
 * it from [2 N + 1] without a version change: a caller built against that header would have been overrun) and now refuses a shorter buffer. */
#define GNNCCA_ABI_VERSION 2
#define GNNCCA_MAX_LAYERS 8

#if defined(GNNCCA_BUILD)
#define GNNCCA_API __attribute__((visibility("default")))
#else
#define GNNCCA_API
#endif

typedef void* gnncca_stream_t;

typedef enum gnncca_status {
    GNNCCA_OK = 0,
    GNNCCA_ERR_INVALID_ARG = 1,     /* null pointer, negative size, inconsistent dims            */
    GNNCCA_ERR_UNSUPPORTED = 2,     /* a legal GRAPH_NET_PARAMS this build has no HIP kernel for */
    GNNCCA_ERR_WORKSPACE = 3,       /* workspace_bytes < gnncca_workspace_bytes(...)             */
    GNNCCA_ERR_HIP = 4,             /* a HIP runtime call failed; see gnncca_last_hip_error()    */
    GNNCCA_ERR_NO_DEVICE = 5        /* no gfx950 device visible                                  */
} gnncca_status;

/* Aggregators of models/mpn.py:192-202 (torch_scatter scatter_add / scatter_mean / scatter_max). */
typedef enum gnncca_agg { GNNCCA_AGG_SUM = 0, GNNCCA_AGG_MEAN = 1, GNNCCA_AGG_MAX = 2 } gnncca_agg;

/* One Linear(+BatchNorm1d eval)(+ReLU) block of models/mlp.py:10-24.  Dropout is the identity in eval. */
typedef struct gnncca_layer {
    int32_t in_dim;
    int32_t out_dim;
    int32_t has_bn;   /* models/mlp.py:14  (use_batchnorm and dim != 1) */
    int32_t relu;     /* models/mlp.py:17  (dim != 1)                   */
} gnncca_layer;

typedef struct gnncca_mlp {
    int32_t n_layers; /* 0 = MLP absent (MLPGraphIndependent passes the input through, mpn.py:133-140) */
    gnncca_layer layers[GNNCCA_MAX_LAYERS];
} gnncca_mlp;

/* Everything MOTMPNet.__init__ derives from GRAPH_NET_PARAMS (models/mpn.py:154-247). */
typedef struct gnncca_mpn_dims {
    int32_t abi_version;        /* GNNCCA_ABI_VERSION */
    int32_t node_in;            /* encoder_feats_dict.nodes[arch].node_in_dim (2048 / 512)  */
    int32_t edge_in;            /* encoder_feats_dict.edges.edge_in_dim (4; 2 for ONLY_*)   */
    int32_t node_dim;           /* node_out_dim = H (32) */
    int32_t edge_dim;           /* edge_out_dim = EF (6) */
    int32_t agg;                /* gnncca_agg */
    int32_t num_enc_steps;      /* L, mpn.py:179 */
    int32_t num_class_steps;    /* mpn.py:180 */
    int32_t reattach_nodes;     /* mpn.py:207 */
    int32_t reattach_edges;     /* mpn.py:208 */
    gnncca_mlp enc_node;        /* encoder.node_mlp                (mpn.py:173) */
    gnncca_mlp enc_edge;        /* encoder.edge_mlp                              */
    gnncca_mlp edge_mlp;        /* MPNet.edge_model.edge_mlp       (mpn.py:221) */
    gnncca_mlp node_mlp;        /* MPNet.node_model.node_mlp       (mpn.py:236) */
    gnncca_mlp cls_edge;        /* classifier.edge_mlp             (mpn.py:174) */
} gnncca_mpn_dims;

/* Optional debug taps (all nullable): the latents the golden vectors also hold.  Row-major fp32, edges
 * in the caller's (original) edge order. */
typedef struct gnncca_trace {
    float* h_enc;    /* [N][H]       encoder node output   (mpn.py:270) */
    float* e_enc;    /* [E][EF]      encoder edge output                */
    float* h_steps;  /* [L][N][H]    node latents after each step (mpn.py:288) */
    float* e_steps;  /* [L][E][EF]   edge latents after each step       */
} gnncca_trace;

/* Bits of the per-call graph flag word (device side, read back with gnncca_read_graph_flags). */
#define GNNCCA_GRAPH_UNSORTED 1u   /* `row` was not non-decreasing: the stable device sort ran     */
#define GNNCCA_GRAPH_BAD_INDEX 2u  /* an index outside [0,N): kernels skipped, logits set to NaN   */
#define GNNCCA_GRAPH_IRREGULAR 4u  /* informational: a big batch with a degree above the padded edge-state
                                      stride chosen from E/N; the step kernels used the compact layout */

GNNCCA_API int gnncca_abi_version(void);
GNNCCA_API const char* gnncca_status_string(int status);
GNNCCA_API int gnncca_last_hip_error(void); /* hipError_t of the last failed HIP call on this thread, 0 if none */

/* Number of `float*` entries gnncca_pack_weights expects, and the canonical order: for each MLP in the
 * order enc_node, enc_edge, edge_mlp, node_mlp, cls_edge, for each layer: weight[out][in], bias[out],
 * then if has_bn: bn_weight, bn_bias, bn_running_mean, bn_running_var (each [out]).
 * Replaces: nothing in the reference (it keeps nn.Parameters); this is the state_dict -> HBM layout step
 * behind MOTMPNet.load_state_dict / .cuda() (main.py:78-82). */
GNNCCA_API int gnncca_param_count(const gnncca_mpn_dims* dims);

/* Bytes of the packed weight blob. 0 on invalid dims. */
GNNCCA_API size_t gnncca_packed_weights_bytes(const gnncca_mpn_dims* dims);

/* HOST function: folds eval-mode BatchNorm into the preceding Linear, splits the MPN weights by input
 * block ([W_src|W_dst|W_edge], [W_node|W_edge], cat order of mpn.py:68 and mpn.py:97), builds the
 * per-node projection matrix, and writes the blob into `packed_host` (the caller uploads it, or
 * broadcasts it to the other ranks over RCCL).  `params` are host pointers in the canonical order. */
GNNCCA_API int gnncca_pack_weights(const gnncca_mpn_dims* dims, const float* const* params, int n_params,
                        void* packed_host, size_t packed_bytes);

/* DEVICE variant of gnncca_pack_weights for the tuned (H = 32, EF = 6) family: the same blob, byte for byte, built by
 * one kernel straight from the parameter tensors in HBM -- a training step (train.py:492-494: the optimizer has just
 * rewritten every parameter) or a load_state_dict() never moves the weights through the host.
 *   gnncca_pack_program(dims, program_host, bytes)  HOST: the copy/fold program for `dims`
 *                                                   (gnncca_pack_program_bytes() bytes); upload it once.
 *   gnncca_pack_weights_device(...)                 enqueues the packing on `stream`.  `params_dev`: a HOST array of
 *                                                   DEVICE pointers in the canonical order; `packed_dev` must have
 *                                                   been zero-filled once (padding words are not rewritten).
 * GNNCCA_ERR_UNSUPPORTED for the generic family (use the host packer). */
GNNCCA_API size_t gnncca_pack_program_bytes(void);
GNNCCA_API int gnncca_pack_program(const gnncca_mpn_dims* dims, void* program_host, size_t program_bytes);
GNNCCA_API int gnncca_pack_weights_device(const gnncca_mpn_dims* dims, const float* const* params_dev, int n_params,
                               const void* program_dev, void* packed_dev, size_t packed_bytes, gnncca_stream_t stream);

/* Bytes of device scratch one forward over a graph of N nodes / E edges needs. */
GNNCCA_API size_t gnncca_workspace_bytes(const gnncca_mpn_dims* dims, int64_t n_nodes, int64_t n_edges);

/* Whether this build has HIP kernels for `dims` (GNNCCA_OK) or not (GNNCCA_ERR_UNSUPPORTED). */
GNNCCA_API int gnncca_supported(const gnncca_mpn_dims* dims);

/* Replaces MOTMPNet.forward (models/mpn.py:250-299) in eval mode: encoder (mpn.py:270), L message
 * passing steps (MetaLayer.forward mpn.py:32-54 = EdgeModel 59-69 + NodeModel 71-101 + aggregator
 * 192-202) and the edge classifier on the last num_class_steps steps (mpn.py:290-297).
 *   x          [N][node_in]  fp32      data.x
 *   edge_index [2][E]        int64     data.edge_index (any order; row-sorted graphs take the fast path)
 *   edge_attr  [E][edge_in]  fp32      data.edge_attr
 *   logits_out [n_out][E]    fp32      n_out = gnncca_num_outputs(dims); the list 'classified_edges'
 * Inputs are read-only.  Asynchronous on `stream`. */
GNNCCA_API int gnncca_mpn_forward(const gnncca_mpn_dims* dims, const void* packed_dev, const float* x,
                       const int64_t* edge_index, const float* edge_attr, int64_t n_nodes,
                       int64_t n_edges, void* workspace, size_t workspace_bytes, float* logits_out,
                       const gnncca_trace* trace, gnncca_stream_t stream);

/* Diagnostic twin of gnncca_mpn_forward for bench.py: attaches a start and a stop hipEvent to every kernel dispatch
 * (hipExtLaunchKernelGGL), SYNCHRONISES the stream and returns each kernel's own execution time -- what
 * rocprofv3 --kernel-trace reports.  Never used on the product path. */
#define GNNCCA_PROFILE_MAX 64
enum { GNNCCA_K_PLAN_ROWS = 0, GNNCCA_K_PLAN_SORT = 1, GNNCCA_K_ENC_GEMM = 2, GNNCCA_K_ENC_REDUCE = 3,
       GNNCCA_K_ENC_TAIL = 4, GNNCCA_K_STEP = 5, GNNCCA_K_STEP_LAST = 6 };
typedef struct gnncca_profile {
    uint32_t options;                   /* in: GNNCCA_OPT_* for the profiled forward */
    int32_t count;                      /* launches recorded */
    int32_t kind[GNNCCA_PROFILE_MAX];   /* GNNCCA_K_* */
    float ms[GNNCCA_PROFILE_MAX];       /* hipEventElapsedTime(start, stop) of the dispatch's own events */
} gnncca_profile;
GNNCCA_API int gnncca_mpn_forward_profiled(const gnncca_mpn_dims* dims, const void* packed_dev, const float* x,
                                const int64_t* edge_index, const float* edge_attr, int64_t n_nodes,
                                int64_t n_edges, void* workspace, size_t workspace_bytes, float* logits_out,
                                gnncca_stream_t stream, gnncca_profile* profile);

/* Train-mode Dropout (models/mlp.py:20-21; config keys `dropout_p` of encoder_feats_dict / edge_model_feats_dict /
 * node_model_feats_dict / classifier_feats_dict).  Masks are generated on the device by a counter-based hash of
 * (*seed_dev, tensor, element) and are NOT stored: the backward evaluates the same hash, so both calls of one training
 * iteration must see the same seed word.  p = 0 switches a group off; a null pointer switches everything off. */
typedef struct gnncca_dropout {
    float p_enc;    /* encoder.node_mlp and encoder.edge_mlp */
    float p_edge;   /* MPNet.edge_model.edge_mlp             */
    float p_node;   /* MPNet.node_model.node_mlp (the messages, before aggregation) */
    float p_cls;    /* classifier.edge_mlp                   */
    const uint64_t* seed_dev;   /* DEVICE word */
} gnncca_dropout;

/* gnncca_mpn_forward with options.  GNNCCA_OPT_EDGE_STATE_BF16: keep the edge latents BETWEEN steps as bf16 in HBM
 * (round to nearest even; all arithmetic, the classifier input and every other buffer stay fp32) -- halves the
 * dominant traffic of the step kernels; honoured by the specialised kernels (shipped config shape), ignored elsewhere.
 * Measured effect on the logits: DESIGN.md section 5. */
#define GNNCCA_OPT_EDGE_STATE_BF16 1u
/* GNNCCA_OPT_ENC_SPLIT3: on batches of >= 4096 nodes the first encoder layer (models/mpn.py:131, 2048 -> 128) runs as a
 * split-bf16 MFMA GEMM; by default with the six products that give fp32-level accuracy, with this option with the three
 * leading ones (x0 w0 + x0 w1 + x1 w0): ~2^-17 relative on that layer's pre-activations (encoder output 5e-6 from fp64
 * instead of 3e-7 .. 1e-6), measured logit deviation 1.5e-7 (tolerance 1e-4), GEMM 19-28 % faster.  Off by default. */
#define GNNCCA_OPT_ENC_SPLIT3 2u
/* GNNCCA_OPT_ENC_UNSPLIT: a forward over >= 4096 nodes never splits K in that layer.  By default mid-size batches (a few thousand
 * to a few ten thousand nodes) run it split-K (partial slabs + a tail launch) and the largest ones un-split, so a graph's logits
 * agree across batch sizes within rounding only (<= 2e-6 asserted, 6e-8 measured).  With this option the mid-size batches take an
 * un-split 32-row kernel whose per-element arithmetic is the big un-split kernel's: a graph's logits are then BIT FOR BIT
 * independent of the batch (or the shard of a sharded batch) it is computed in, as long as that batch has >= 4096 nodes.
 * Price: encoder 48 instead of 44 us at 8192 nodes (plan launch included), 87 instead of 70 us at 16 384.
 * Off by default. */
#define GNNCCA_OPT_ENC_UNSPLIT 4u
/* GNNCCA_OPT_COLUMN_RANGES (forwards with >= 2 message-passing steps, specialised kernels): step 1 derives, per source node, whether
 * its target ids form at most two contiguous runs -- true for every graph the reference builds (inference.py:209-216: per camera,
 * cartesian_prod with the detections of the other cameras) and for dense graphs -- and steps 2 ... L then COMPUTE the target ids
 * instead of streaming them (4 of 56 B per edge, and the P_dst gather no longer waits for an id); a forward with any other node
 * streams them on every step, as without the option.  Results are bit for bit the same either way
 * (tests/test_gpu_column_ranges.py).  OFF by default: measured on MI355X it gains nothing -- 1 x dense256 27.6 -> 28.4 us per
 * forward, 64 x dense128 99.9 -> 101.3, 64 x dense256 / 512 x dense128 / 200 x dense256 within noise (profiles/r04_logs/ab_ranges1.log):
 * the id load and the gather behind it are not on these launches' critical path, and step 1 pays for the derivation. */
#define GNNCCA_OPT_COLUMN_RANGES 8u
GNNCCA_API int gnncca_mpn_forward_ex(const gnncca_mpn_dims* dims, const void* packed_dev, const float* x,
                                     const int64_t* edge_index, const float* edge_attr, int64_t n_nodes,
                                     int64_t n_edges, void* workspace, size_t workspace_bytes, float* logits_out,
                                     const gnncca_trace* trace, uint32_t options, gnncca_stream_t stream);

/* len(outputs['classified_edges']) for these dims (mpn.py:277-297). */
GNNCCA_API int gnncca_num_outputs(const gnncca_mpn_dims* dims);

/* ---- the step before the MPN -- graph construction + edge attributes ------------------
 * Replaces the per-frame Python of inference.py:189-279 (duplicated at train.py:257-361 and 616-692).
 * All pointers are device memory; one entry per detection (node) unless noted; frames are concatenated. */
typedef struct gnncca_frames {
    const double* xw;          /* ground-plane x  (data_df['xw'], float64 as pandas holds it)                 */
    const double* yw;          /* ground-plane y                                                              */
    const double* max_dist;    /* [G]   CONFIG['CONV_TO_M'][dataset] of each frame (inference.py:239)        */
    const int32_t* person_id;  /* data_df['id'] (any relabelling that preserves equality)                     */
    const int32_t* cam;        /* data_df['id_cam']                                                           */
    const int32_t* graph_of;   /* frame index of each node                                                    */
    const int32_t* graph_ptr;  /* [G+1] node range of each frame                                              */
    const int32_t* src_order;  /* node ids in the order the reference emits their out-edges (camera-major)     */
    const int32_t* edge_ptr;   /* [N+1] first edge of each position of src_order                              */
} gnncca_frames;
enum { GNNCCA_EDGE_ATTR_FULL = 0, GNNCCA_EDGE_ATTR_ONLY_APPEARANCE = 1, GNNCCA_EDGE_ATTR_ONLY_DIST = 2 };

/* out = x / max(||column||_2, 1e-12): F.normalize(x, p=2, dim=0) of inference.py:189-190.
 * scratch: (ceil(n_rows/64) + 1) * n_cols floats. */
GNNCCA_API int gnncca_normalize_columns(const float* x, int64_t n_rows, int64_t n_cols, float* scratch,
                                        float* out, gnncca_stream_t stream);
/* The same for up to TWO matrices of n_rows <= 4096 rows (the reid and the node embeddings of a batch of frames,
 * inference.py:189-190) in ONE launch; bit for bit gnncca_normalize_columns' results.  x1 may be NULL (n_cols1 = 0).
 * More rows: GNNCCA_ERR_UNSUPPORTED (use gnncca_normalize_columns). */
GNNCCA_API int gnncca_normalize_columns2(const float* x0, int64_t n_cols0, float* out0, const float* x1, int64_t n_cols1,
                                         float* out1, int64_t n_rows, gnncca_stream_t stream);

/* HOST side of the graph construction: the edge enumeration of a batch of frames (inference.py:207-212: per frame, cameras in
 * np.unique order, a camera's nodes in ascending id, each connected to every node of the OTHER cameras) and the staging image that
 * gnncca_build_edges reads, written into host memory `staging` (pinned memory lets the caller upload it without blocking):
 *   f64 xw[n], yw[n], max_dist[g];  i64 ids[n];  i32 person[n], cam[n], graph_of[n], graph_ptr[g+1], src_order[n], edge_ptr[n+1],
 *   edge_ptr_g[g+1]  (first edge of each frame)
 * -- the fields of gnncca_frames at those offsets once uploaded.  All inputs are host arrays, one entry per detection (frames
 * concatenated) or per frame.  Returns the number of edges E >= 0, or -status (graph_sizes that do not sum to n: INVALID_ARG;
 * 2^31 edges or more: UNSUPPORTED).  Replaces the Python list comprehensions of inference.py:199-216. */
GNNCCA_API size_t gnncca_plan_frames_bytes(int64_t n_nodes, int64_t n_frames);
GNNCCA_API int64_t gnncca_plan_frames(const double* xw, const double* yw, const int64_t* ids, const int64_t* id_cam, int64_t n_nodes,
                                      const int64_t* graph_sizes, const double* max_dist, int64_t n_frames, void* staging,
                                      size_t staging_bytes);

/* edge_index [2][E] int64, edge_attr [E][4 or 2] fp32, edge_labels [E] fp32 in the reference's edge order:
 * cartesian_prod per camera (inference.py:207-212), ground-plane L2 and L1 distance / max_dist in float64 then
 * cast (229-242), F.pairwise_distance and F.cosine_similarity of the reid rows (222-226), same-identity labels
 * (262-266); node ids are the batch-global ones Batch.from_data_list produces (269-279). */
GNNCCA_API int gnncca_build_edges(const gnncca_frames* frames, const float* reid, int32_t reid_dim, int64_t n_nodes,
                                  int64_t n_edges, int32_t mode, int64_t* edge_index_out, float* edge_attr_out,
                                  float* edge_labels_out, gnncca_stream_t stream);

/* ---- the step after the MPN -- threshold, pruning, flow counts, identity clusters --------
 * probs = sigmoid(logits), predictions = (probs >= 0.5) as int64 0/1 (inference.py:286-291). */
GNNCCA_API int gnncca_post_threshold(const float* logits, int64_t n_edges, float* probs_out,
                                     int64_t* predictions_out, gnncca_stream_t stream);
GNNCCA_API size_t gnncca_post_workspace_bytes(int64_t n_nodes, int64_t n_edges);
/* pruned_out[k] = predictions[k] && the reverse edge is active too (utils.remove_edges_single_direction,
 * libs/utils.py:387-404); flow_out / flow_in [N] int32 = active pruned edges leaving / entering each node
 * (scatter_add at libs/utils.py:54-55); labels_out [N] int32 = smallest node id of the node's connected component
 * over the pruned edges, *n_clusters_out = number of components incl. isolated nodes (the partition
 * utils.compute_SCC_and_Clusters, libs/utils.py:295-317, returns on the pruned graph).  edge_index may be in any
 * order.  The bridge-based rounding / splitting heuristics stay on the host. */
GNNCCA_API int gnncca_post_prune_cluster(const int64_t* edge_index, const int64_t* predictions, int64_t n_nodes,
                                         int64_t n_edges, void* workspace, size_t workspace_bytes,
                                         int64_t* pruned_out, int32_t* flow_out, int32_t* flow_in,
                                         int32_t* labels_out, int32_t* n_clusters_out, gnncca_stream_t stream);
/* The same over a batch of frame graphs laid out as Batch.from_data_list lays them out (inference.py:279): frame g owns
 * nodes [node_ptr[g], node_ptr[g+1]) and the contiguous edges [edge_ptr[g], edge_ptr[g+1]) (int32 arrays of
 * n_frames + 1 entries in DEVICE memory).  Components never cross frames, so each frame's clustering runs in its own
 * workgroup, frames-wide in parallel. */
GNNCCA_API int gnncca_post_prune_cluster_frames(const int64_t* edge_index, const int64_t* predictions, int64_t n_nodes,
                                                int64_t n_edges, const int32_t* node_ptr_dev, const int32_t* edge_ptr_dev,
                                                int32_t n_frames, void* workspace, size_t workspace_bytes,
                                                int64_t* pruned_out, int32_t* flow_out, int32_t* flow_in,
                                                int32_t* labels_out, int32_t* n_clusters_out, gnncca_stream_t stream);

/* The same with the two TRIGGER bits of the host heuristics per frame (or for the whole graph when no frame ranges are given):
 * triggers_out[g] bit 0 (GNNCCA_POST_TRIGGER_ROUNDING) = a node of frame g has flow_out or flow_in > 3 after the pruning -- the condition
 * under which utils.compute_rounding changes anything (libs/utils.py:58-62); bit 1 (GNNCCA_POST_TRIGGER_SPLITTING) = a cluster of frame g
 * has more than four members -- utils.disjoint_big_clusters' (libs/utils.py:321-322).  A frame with no bit set already has its final
 * ID_pred under ROUNDING / PRUNING / SPLITTING = True: rounding returns [], the second pruning is idempotent, splitting returns at once.
 * `sizes_scratch` [N] int32 and `triggers_out` [max(n_frames, 1)] int32 are zeroed by the call (one memset with the counters when they
 * follow n_clusters_out in memory: flow_out | flow_in | n_clusters | sizes | triggers).  Both null: the call above. */
#define GNNCCA_POST_TRIGGER_ROUNDING 1
#define GNNCCA_POST_TRIGGER_SPLITTING 2
GNNCCA_API int gnncca_post_prune_cluster_frames_ex(const int64_t* edge_index, const int64_t* predictions, int64_t n_nodes,
                                                   int64_t n_edges, const int32_t* node_ptr_dev, const int32_t* edge_ptr_dev,
                                                   int32_t n_frames, void* workspace, size_t workspace_bytes,
                                                   int64_t* pruned_out, int32_t* flow_out, int32_t* flow_in,
                                                   int32_t* labels_out, int32_t* n_clusters_out, int32_t* sizes_scratch,
                                                   int32_t* triggers_out, gnncca_stream_t stream);

/* Row N2, second half, for ONE frame on the HOST ("bridges-based heuristics stay on CPU"): the reference's call sequence
 * inference.py:306-345 after the threshold -- by the three switches of config_inference.yaml:6-8 (all True as shipped):
 * utils.remove_edges_single_direction (libs/utils.py:387-404) -> utils.compute_rounding (25-173) -> remove_edges_single_direction ->
 * utils.disjoint_big_clusters (319-386) -> utils.compute_SCC_and_Clusters (295-317).  Plain host arrays: `src` / `dst` [E] the frame's
 * edges in the reference's edge order with node ids in [node_base, node_base + n_nodes), `probs` [E] the sigmoid values, `predictions`
 * [E] in: the thresholded (or already pruned) 0 / 1 predictions, out: the final ones.  labels_out [n_nodes] (optional) = smallest
 * batch-global node id of the node's final cluster, *n_clusters_out (optional) = number of final clusters, id_pred_out [n_nodes]
 * (optional) = the reference's ID_pred, its label NUMBERING included (networkx's generation order: see csrc/post_host.cpp).
 * Frames are processed one at a time as the reference does (validation batch size 1, main.py:368).  Predictions other than 0 / 1 (inference.py:291
 * thresholds to exactly those) are GNNCCA_ERR_INVALID_ARG since round 6. */
#define GNNCCA_POST_ROUNDING 1
#define GNNCCA_POST_PRUNING 2
#define GNNCCA_POST_SPLITTING 4
GNNCCA_API int gnncca_post_finalize_frame_host(const int64_t* src, const int64_t* dst, int64_t node_base, int64_t n_nodes,
                                               int64_t n_edges, const float* probs, int64_t* predictions, int32_t switches,
                                               int32_t* labels_out, int32_t* n_clusters_out, int64_t* id_pred_out);

/* The same for a list of frames of one batch, dealt to host threads (frames are independent): batch-wide arrays in Batch.from_data_list
 * layout, node_ptr / edge_ptr [G + 1] on the HOST, frames[i] the i-th frame to finalize (the ones whose trigger word is set),
 * clusters_out[i] its final cluster count; n_threads 0 = one per hardware thread, at most 16. */
GNNCCA_API int gnncca_post_finalize_frames_host(const int64_t* src, const int64_t* dst, const int32_t* node_ptr, const int32_t* edge_ptr,
                                                const int32_t* frames, int32_t n_listed, const float* probs, int64_t* predictions,
                                                int32_t switches, int32_t* labels, int32_t* clusters_out, int32_t n_threads);

/* The ASYNCHRONOUS form (round 6): a persistent pool of host threads finalizes batches while the caller enqueues the next batch's GPU
 * chain -- the reference's loop pays its heuristics between two forwards (inference.py:294-345 runs on the host while the GPU waits); here
 * batch k's host pass overlaps batch k + 1's launches.  The caller copies the batch's results to HOST memory (pinned) on a side stream
 * behind an event and submits pointers into that copy: `triggers` [G] (gnncca_post_prune_cluster_frames_ex's words), src / dst [E],
 * probs [E], `predictions` [E] in: the pruned predictions, out: the final ones, `labels` [N] in / out (the device chain's convention: a
 * component's smallest batch-global node id), `n_clusters` [1] in: the device chain's count, out: the final count.  A pool thread waits for
 * `ready_event` (a hipEvent_t recorded behind the copy; null: the data is already there) on device `device`, lists the frames whose
 * trigger word meets the switches and the pool's threads finalize them (gnncca_post_finalize_frame_host per frame; no allocation per
 * frame).  gnncca_post_pool_submit returns a ticket >= 0 (or -status); gnncca_post_pool_wait blocks until that batch is final, writes
 * the finalized frame ids to frames_out [<= G] / their number to *n_frames_out (either may be null), releases the ticket and returns
 * the first non-zero status of any frame.  The buffers must stay valid until the wait returns; `n_clusters` needs `labels` (the correction counts the
 * flagged frames' components in them).  n_threads 0 = hardware threads - 2,
 * at most 16. */
typedef struct gnncca_post_pool gnncca_post_pool;
typedef struct gnncca_post_batch {
    const int64_t* src;
    const int64_t* dst;
    const int32_t* node_ptr;       /* [G + 1] host */
    const int32_t* edge_ptr;       /* [G + 1] host */
    int32_t n_frames;
    int32_t switches;              /* GNNCCA_POST_ROUNDING | GNNCCA_POST_PRUNING | GNNCCA_POST_SPLITTING */
    const int32_t* triggers;       /* [G] */
    const float* probs;
    int64_t* predictions;
    int32_t* labels;
    int32_t* n_clusters;
    void* ready_event;             /* hipEvent_t or null */
    int32_t device;
} gnncca_post_batch;
GNNCCA_API gnncca_post_pool* gnncca_post_pool_create(int32_t n_threads);
GNNCCA_API int32_t gnncca_post_pool_threads(const gnncca_post_pool* pool);
GNNCCA_API void gnncca_post_pool_destroy(gnncca_post_pool* pool);
GNNCCA_API int64_t gnncca_post_pool_submit(gnncca_post_pool* pool, const gnncca_post_batch* batch);
/* gnncca_post_pool_submit for results that still sit in DEVICE memory: enqueues the D2H copy of `nbytes` from device_src to host_dst
 * (pinned) on `stream` -- the stream the batch's chain was enqueued on -- records the event the job waits for behind it and submits
 * `batch`, whose pointers point into host_dst.  One call, no synchronisation. */
GNNCCA_API int64_t gnncca_post_pool_submit_copy(gnncca_post_pool* pool, const gnncca_post_batch* batch, const void* device_src,
                                                void* host_dst, size_t nbytes, int32_t device, gnncca_stream_t stream);
GNNCCA_API int gnncca_post_pool_wait(gnncca_post_pool* pool, int64_t ticket, int32_t* frames_out, int32_t* n_frames_out);
/* diagnostics: gnncca_post_pool_wait plus the job's life in microseconds -- times_us_out[0..3] = submit -> picked up by a pool thread,
 * -> its event had completed, -> its last frame was final, -> this call returned (negative: the caller arrived before the job was done). */
GNNCCA_API int gnncca_post_pool_wait_timed(gnncca_post_pool* pool, int64_t ticket, int32_t* frames_out, int32_t* n_frames_out,
                                           double* times_us_out);

/* ---- rows N1 + the path + N2 in ONE call: a batch of frames from the uploaded staging image to identity clusters -------------------
 * The per-batch body of inference.py:189-345 (normalise the embeddings, build the graph, MOTMPNet.forward, sigmoid / threshold,
 * prune, flow counts, clusters) as the same launches gnncca_normalize_columns2 / gnncca_build_edges / gnncca_mpn_forward_ex /
 * gnncca_post_threshold / gnncca_post_prune_cluster_frames make, issued from one native call: what a per-batch loop pays for at this
 * size is host time per launch, and the glue between five calls is a third of it.  Every pointer is device memory; outputs are
 * caller-allocated.  `staged_dev` is the image gnncca_plan_frames wrote, uploaded as is.  Batches of more than 4096 detections:
 * GNNCCA_ERR_UNSUPPORTED (use the separate entry points). */
typedef struct gnncca_frames_io {
    const void* staged_dev;        /* gnncca_plan_frames' staging image on the device                                  */
    int64_t n_nodes, n_frames, n_edges;
    const float* node_embeds;      /* [N][node_in]  (raw; normalised into node_norm when `normalize`)                  */
    const float* reid_embeds;      /* [N][reid_dim]                                                                    */
    int32_t reid_dim, mode, normalize;
    float* node_norm;              /* [N][node_in]  out (normalize != 0): the MPN's x                                  */
    float* reid_norm;              /* [N][reid_dim] out (normalize != 0)                                               */
    int64_t* edge_index;           /* [2][E] out                                                                       */
    float* edge_attr;              /* [E][4 or 2] out                                                                  */
    float* edge_labels;            /* [E] out                                                                          */
    float* logits;                 /* [n_out][E] out                                                                   */
    float* probs;                  /* [E] out: sigmoid of the last classified step                                     */
    int64_t* predictions;          /* [E] out                                                                          */
    int64_t* pruned;               /* [E] out                                                                          */
    int32_t* counters;             /* [3 N + 1 + G] out: flow_out | flow_in | n_clusters | cluster sizes (scratch) | triggers [G] */
    int32_t* labels;               /* [N] out                                                                          */
    int64_t counters_len;          /* int32 words the caller allocated behind `counters`: < 3 N + 1 + G -> GNNCCA_ERR_INVALID_ARG  */
} gnncca_frames_io;
GNNCCA_API int gnncca_frames_forward(const gnncca_mpn_dims* dims, const void* packed_dev, const gnncca_frames_io* io,
                                     void* mpn_workspace, size_t mpn_workspace_bytes, void* post_workspace,
                                     size_t post_workspace_bytes, uint32_t options, gnncca_stream_t stream);

/* ---- backward pass (training through the HIP kernels, train.py:454-494) -----------------
 * Supported (GNNCCA_OK from gnncca_backward_supported): the MFMA family (both reattach flags, all three
 * aggregators), two-layer node encoder, L >= 1, BatchNorm nowhere or only between the classifier's two layers -- i.e.
 * both shipped config shapes (config_training.yaml:94-181, config_inference.yaml:76-163).  `saved` holds the latents written by gnncca_mpn_forward's trace taps for the same
 * inputs and weights; `params_dev` / `grads_dev` are DEVICE pointers to the raw parameters / their gradients in the
 * canonical order of gnncca_param_count (row-major, un-split, exactly the nn.Parameter layouts).  grads are
 * overwritten.  grad_logits: [n_out][E]. */
GNNCCA_API int gnncca_backward_supported(const gnncca_mpn_dims* dims);
GNNCCA_API size_t gnncca_backward_workspace_bytes(const gnncca_mpn_dims* dims, int64_t n_nodes, int64_t n_edges);
GNNCCA_API int gnncca_mpn_backward(const gnncca_mpn_dims* dims, const float* const* params_dev, int n_params,
                                   const float* x, const int64_t* edge_index, const float* edge_attr, int64_t n_nodes,
                                   int64_t n_edges, const gnncca_trace* saved, const float* cls_bn_stat,
                                   const float* grad_logits, float* const* grads_dev, void* workspace,
                                   size_t workspace_bytes, gnncca_stream_t stream);
/* gnncca_mpn_backward with options.  GNNCCA_BWD_GRADS_ZEROED: the caller has already zero-filled every buffer of
 * `grads_dev` (e.g. they are views of one flat buffer cleared by a single fill), so the per-parameter clears are
 * skipped -- sixteen fewer enqueues per training step for the shipped configs. */
#define GNNCCA_BWD_GRADS_ZEROED 1u
GNNCCA_API int gnncca_mpn_backward_ex(const gnncca_mpn_dims* dims, const float* const* params_dev, int n_params,
                                      const float* x, const int64_t* edge_index, const float* edge_attr, int64_t n_nodes,
                                      int64_t n_edges, const gnncca_trace* saved, const float* cls_bn_stat,
                                      const float* grad_logits, float* const* grads_dev, void* workspace,
                                      size_t workspace_bytes, uint32_t options, gnncca_stream_t stream);
/* Train-mode classifier when a BatchNorm1d sits between its two layers (the shipped inference config): recomputes the
 * logits of every classified step from the saved edge latents with BATCH statistics over the E edges, updates
 * running_mean / running_var in place (momentum 0.1, unbiased variance, as torch.nn.BatchNorm1d), and returns per
 * step and hidden unit (mean, 1/sqrt(var + eps)) in bn_stat_out [n_out][C1][2] for gnncca_mpn_backward (cls_bn_stat).
 * scratch: 2 * C1 doubles. */
GNNCCA_API int gnncca_classifier_train(const gnncca_mpn_dims* dims, const float* const* params_dev, int n_params,
                                       const float* e_steps, int64_t n_edges, void* scratch, float* bn_stat_out,
                                       float* logits_out, gnncca_stream_t stream);

/* Train-mode variants with Dropout (train.py:454-494 with `dropout_p` > 0 somewhere in GRAPH_NET_PARAMS): the same three
 * calls with a gnncca_dropout; `trace` is required (the post-dropout latents it receives are what the backward reads, and
 * they carry the ReLU x Dropout masks of every saved activation: y_saved > 0 <=> kept and positive).  A null `dropout`
 * makes each of them its plain counterpart. */
GNNCCA_API int gnncca_mpn_forward_train(const gnncca_mpn_dims* dims, const void* packed_dev, const float* x,
                                        const int64_t* edge_index, const float* edge_attr, int64_t n_nodes, int64_t n_edges,
                                        void* workspace, size_t workspace_bytes, float* logits_out, const gnncca_trace* trace,
                                        const gnncca_dropout* dropout, gnncca_stream_t stream);
GNNCCA_API int gnncca_classifier_train_dropout(const gnncca_mpn_dims* dims, const float* const* params_dev, int n_params,
                                               const float* e_steps, int64_t n_edges, void* scratch, float* bn_stat_out,
                                               float* logits_out, const gnncca_dropout* dropout, gnncca_stream_t stream);
GNNCCA_API int gnncca_mpn_backward_train(const gnncca_mpn_dims* dims, const float* const* params_dev, int n_params,
                                         const float* x, const int64_t* edge_index, const float* edge_attr, int64_t n_nodes,
                                         int64_t n_edges, const gnncca_trace* saved, const float* cls_bn_stat,
                                         const float* grad_logits, float* const* grads_dev, void* workspace,
                                         size_t workspace_bytes, uint32_t options, const gnncca_dropout* dropout,
                                         gnncca_stream_t stream);

/* Layer-by-layer training engine (SURVEY.md 8f row N3 remainder; train.py:454-494 through models/mpn.py:250-299 and
 * models/mlp.py:4-28 op for op): EVERY legal GRAPH_NET_PARAMS in train mode -- BatchNorm1d with batch statistics in any MLP
 * (running_mean / running_var updated in place with momentum 0.1, as torch.nn.BatchNorm1d; the caller bumps num_batches_tracked),
 * Dropout behind any ReLU (masks from `dropout`, as above), any widths / depths, sum | mean | max, both reattach flags.
 * `params_dev`: the gnncca_pack_weights order (per layer: weight, bias, [BatchNorm weight, bias, running_mean, running_var]).
 * `tape` (gnncca_train_tape_bytes) receives what autograd would keep and is read back by gnncca_train_backward, which ADDS
 * d loss / d parameter into `grads_dev` (same order; entries of buffers may be null; zero them first) given
 * `grad_logits` [n_out][E].  The shipped shapes have the fused pair gnncca_mpn_forward_train / gnncca_mpn_backward_train;
 * this engine is the one that covers everything else (one launch per op, correctness first). */
GNNCCA_API size_t gnncca_train_tape_bytes(const gnncca_mpn_dims* dims, int64_t n_nodes, int64_t n_edges);
/* Where the tape of gnncca_train_forward keeps the latents a forward hook on the reference's containers would see (models/mpn.py:270,
 * 288: the outputs of `encoder` and of every `MPNet` call, train-mode values: batch-statistics BatchNorm, Dropout applied): byte
 * offsets into the tape, offsets_out[0] = encoder node output [N][node_dim], [1] = encoder edge output [E][edge_dim], then per step s
 * (0-based) [2 + 2 s] = node latents after the step [N][node_dim], [3 + 2 s] = edge latents after the step [E][edge_dim]; -1 where
 * the MLP has no layer (its input passes through).  n_offsets must be 2 + 2 * num_enc_steps.  Pure function of (dims, N, E). */
GNNCCA_API int gnncca_train_tape_latents(const gnncca_mpn_dims* dims, int64_t n_nodes, int64_t n_edges, int64_t* offsets_out,
                                         int n_offsets);
GNNCCA_API int gnncca_train_forward(const gnncca_mpn_dims* dims, float* const* params_dev, int n_params, const float* x,
                                    const int64_t* edge_index, const float* edge_attr, int64_t n_nodes, int64_t n_edges,
                                    void* tape, size_t tape_bytes, float* logits_out, const gnncca_dropout* dropout,
                                    gnncca_stream_t stream);
GNNCCA_API int gnncca_train_backward(const gnncca_mpn_dims* dims, float* const* params_dev, int n_params, const float* x,
                                     const int64_t* edge_index, const float* edge_attr, int64_t n_nodes, int64_t n_edges,
                                     void* tape, size_t tape_bytes, const float* grad_logits, float* const* grads_dev,
                                     const gnncca_dropout* dropout, gnncca_stream_t stream);

/* Stand-alone calls of the sub-modules, which the reference allows (models/mlp.py:26-28 MLP.forward; models/mpn.py:128-142
 * MLPGraphIndependent.forward, :59-69 EdgeModel.forward, :71-101 NodeModel.forward, :32-54 MetaLayer.forward): eval semantics
 * (BatchNorm from the running statistics, Dropout = identity), one launch per op.  MOTMPNet.forward does not use them.
 *   gnncca_mlp_eval     : out[rows][out_dim] = mlp(in[rows][in_dim]); params in the order weight, bias, [BN weight, bias, mean, var]
 *   gnncca_gather_cat   : out[r] = cat(a[ia[r]], b[ib[r]], c[ic[r]]) (null ids: row r itself; width 0: segment absent) -- the
 *                         x[row], x[col] gathers and the torch.cat of mpn.py:48,68,97 without a torch kernel
 *   gnncca_aggregate    : out[i] = sum | mean | max over {k : edge_index[0][k] == i} of messages[k], empty -> 0 (mpn.py:99,192-202) */
GNNCCA_API size_t gnncca_mlp_eval_workspace_bytes(const gnncca_mlp* mlp, int64_t rows);
GNNCCA_API int gnncca_mlp_eval(const gnncca_mlp* mlp, const float* const* params_dev, int n_params, const float* in, int64_t rows,
                               float* out, void* workspace, size_t workspace_bytes, gnncca_stream_t stream);
GNNCCA_API int gnncca_gather_cat(const float* a, const int64_t* ia, int wa, int64_t rows_a, const float* b, const int64_t* ib, int wb,
                                 int64_t rows_b, const float* c, const int64_t* ic, int wc, int64_t rows_c, int64_t rows, float* out,
                                 gnncca_stream_t stream);
GNNCCA_API size_t gnncca_aggregate_workspace_bytes(int64_t n_nodes, int64_t n_edges);
GNNCCA_API int gnncca_aggregate(const float* messages, const int64_t* edge_index, int64_t n_nodes, int64_t n_edges, int width, int agg,
                                float* out, void* workspace, size_t workspace_bytes, gnncca_stream_t stream);

/* One launch that copies a frame (x [N][node_in], edge_index [2][E] int64, edge_attr [E][edge_in]) into buffers of a canonical shape
 * (x_pad [n_real_max + n_dummy][node_in], edge_index_pad [2][e_pad], edge_attr_pad [e_pad][edge_in]) and writes the padding: zero rows,
 * and e_pad - E self loops with zero attributes on the n_dummy extra nodes behind the real ones (a disjoint dummy component: the logits of
 * the frame's own edges are those of the frame alone, models/mpn.py has no cross-component term; rows stay sorted).  What lets ONE captured
 * HIP graph serve every frame of the per-frame loop of inference.py:173-283 (gnn_cca_amd.inference.GraphedForward(pad_to=...)). */
GNNCCA_API int gnncca_pad_frame(const float* x, int64_t n_nodes, const int64_t* edge_index, const float* edge_attr, int64_t n_edges,
                                float* x_pad, int64_t n_real_max, int n_dummy, int64_t* edge_index_pad, float* edge_attr_pad,
                                int64_t e_pad, int node_in, int edge_in, gnncca_stream_t stream);

/* Synchronises `stream` and returns the flag word of the last forward that used `workspace`. */
GNNCCA_API int gnncca_read_graph_flags(const void* workspace, uint32_t* flags_out, gnncca_stream_t stream);
/* The same plus, in flags_out[1], the column-range verdict of that forward: 0 = every node's target ids were <= 2 contiguous runs (or
 * the forward never asked: no GNNCCA_OPT_COLUMN_RANGES, L < 2, general kernels), 1 = some node's were not and every step streamed them. */
GNNCCA_API int gnncca_read_graph_flags2(const void* workspace, uint32_t flags_out[2], gnncca_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNCCA_MPN_H */
