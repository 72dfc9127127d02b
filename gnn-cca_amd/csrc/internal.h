// internal.h -- layouts shared by the host packer (pack.cpp) and the HIP kernels (mpn_kernels.hip).
// Not part of the public ABI (include/gnncca_mpn.h is).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "gnncca_mpn.h"

namespace gnncca {

extern thread_local int g_last_hip_error;  // hipError_t of the last failed HIP call on this thread

// Diagnostic switches (the A/B experiments of DESIGN.md 5; listed in DESIGN.md 11).  They change kernel SELECTION, hence
// summation order, so the shipped library ignores every one of them unless GNNCCA_DIAG=1 is set in the same environment:
// a stray variable in one rank's environment cannot silently change its numerics.
const char* diag_env(const char* name);
int diag_env_int(const char* name, int fallback, int lo, int hi);

constexpr uint32_t kBlobMagic = 0x4D504E36u;  // "MPN6": bumped with every change of the blob layout (a blob is only
                                              // valid for the library build that packed it; load_packed_blob checks)
constexpr int kH = 32;        // node latent width the MFMA step kernel is built for (node_out_dim)
constexpr int kEF = 6;        // edge latent width (edge_out_dim): 3 k-steps of v_mfma_f32_32x32x2_f32
constexpr int kProjOut = 48;  // per-node projection slots: [0,6) P_dst, [8,14) P_src+b_e, [16,48) Q+b_n
constexpr int kPdStride = 8;  // floats per node in the P_dst gather table (32 B rows)
constexpr int kPsQStride = 40;  // floats per node in the (P_src | Q) table: slots 8..47 of the projection
constexpr int kMaxCls = 8;    // widest hidden layer of the edge classifier handled in registers
constexpr int kMaxEdgeIn = 16;

// Families of GRAPH_NET_PARAMS this build has kernels for.
enum Family : uint32_t {
    kFamilyNone = 0,
    kFamilyMfma32x6 = 1,  // H = 32, EF = 6, single-layer edge/node MLPs (both shipped configs)
    kFamilyGeneric = 2,   // any other legal GRAPH_NET_PARAMS: op-for-op kernels, correctness first
};

// Header of the packed weight blob.  All offsets are in floats from the start of the blob and are
// multiples of 4 (16-byte aligned).  BatchNorm (eval) is already folded into weight and bias.
struct BlobHeader {
    uint32_t magic, abi_version, family, total_floats;
    int32_t enc_node_layers;
    int32_t enc_node_w[GNNCCA_MAX_LAYERS];  // [out][in] row-major
    int32_t enc_node_b[GNNCCA_MAX_LAYERS];  // [out]
    int32_t enc_last_wT;                    // last encoder layer transposed: [in][32]
    int32_t enc_edge_w, enc_edge_b;         // [6][edge_in], [6]
    int32_t wee;                            // [6][ef*6]   edge-feature block of the edge MLP
    int32_t wne_b;                          // [3][64]     MFMA B operand: Wne[lane&31][2s + (lane>>5)]
    int32_t proj_wT;                        // [nf*32][48] per-node projection, transposed
    int32_t proj_b;                         // [48]        biases b_e (slots 8..13) and b_n (16..47)
    int32_t cls_layers, cls_hidden;         // 1: Linear(6,1);  2: Linear(6,C1)+ReLU, Linear(C1,1)
    int32_t cls_w1, cls_b1, cls_w2, cls_b2;
    int32_t fast_consts;                    // [kFastConsts] contiguous copy of the per-step scalars (see below), or 0
    int32_t enc_w3;                         // first encoder weight as 3 bf16 pieces, [in/32][3][out][32], or 0
    int32_t wne_bf16;                       // [9][64] dwords: W_ne as packed bf16 piece pairs, the B operands of msg_bf16.cuh
    int32_t enc_w2h;                        // first encoder weight as 2 fp16 pieces (w = w0 + w1 / 2048), the B-operand image of enc_f16.cuh:
                                            // [in/32][2 pieces][4 column tiles][2 k-steps][64 lanes][8] halfs (pack.cpp: w2h_index); or 0
    int32_t enc_w2h_bad;                    // [kW2hBadWords] words: non-zero where a weight does not fit fp16 (|w| >= 65520): the fp16 GEMM then
                                            // hands every tile to its bf16 arm
    int32_t pad[3];
};
constexpr int kW2hBadWords = 64;            // one per pack block (pack_device_kernel's grid is 64 wide): no atomics, nothing to reset
constexpr float kF16Limit = 65520.0f;       // the smallest magnitude that fp16 round-to-nearest-even turns into infinity
constexpr float kF16Tiny = 0.0009765625f;   // 2^-10.  The fp16 pieces carry 2^-36 ABSOLUTE precision below fp16's normal range (6.1e-5): per dot product
                                            // sqrt(K) |w| 2^-36 ~ 1e-11 -- far below an fp32 ulp of the biased sum it feeds, but as RELATIVE error of
                                            // the product alone sqrt(K) 2^-36 / |x| (2.5e-6 at a largest |x| of 2^-10, 1e-5 at 1e-6).  A tile whose
                                            // LARGEST |x| is under this (and not zero) therefore takes the range arm like a tile beyond 65520 does
                                            // (ADVICE r5).  Column-normalised embeddings sit at 1 / sqrt(N): a million-node batch still has tile
                                            // maxima of 3.7e-3, so no ordinary input pays for the guard.

// Layout of the `fast_consts` block (floats): the per-step scalars mpn_step_fast_kernel reads into SGPRs.
// Present when edge_in == 4, no reattach flags and the classifier is Linear(6,4)+ReLU+Linear(4,1).
// The three matrices are stored TRANSPOSED ([in][out]) so that the weights of two adjacent outputs are one aligned
// SGPR pair, the operand form of v_pk_fma_f32.
constexpr int kFcEncW = 0;     // [4][6]  enc_edge_w^T
constexpr int kFcEncB = 24;    // [6]
constexpr int kFcWee = 32;     // [6][6]  wee^T
constexpr int kFcCw1 = 68;     // [6][4]  cls_w1^T
constexpr int kFcCb1 = 92;     // [4]
constexpr int kFcCw2 = 96;     // [4]
constexpr int kFcCb2 = 100;    // [1]
constexpr int kFcProjB = 104;  // [48]
constexpr int kFastConsts = 152;
bool fast_consts_ok(const gnncca_mpn_dims* d);

// Device-side packing (gnncca_pack_weights_device): the blob as a list of strided copies out of the parameter
// tensors, built once per GRAPH_NET_PARAMS by pack_program() and interpreted by pack_device_kernel -- the same
// folding / splitting as gnncca_pack_weights, bit for bit, without the parameters ever visiting the host.
struct PackSeg {
    int32_t kind;               // 0: weight element, 1: bias element, 2: weight element split into 3 bf16 planes,
                                // 3: W_ne element (row = channel, column = k < 6) as bf16 pieces in the MsgB lane layout,
                                // 4: weight element split into 2 fp16 pieces in the swizzled chunk image of BlobHeader::enc_w2h
                                //    (dst = enc_w2h, unit0 unused, plane = float offset of the kW2hBadWords flag words)
    int32_t dst;                // float offset in the blob (kind 2: offset of plane 0)
    int32_t param;              // index of the Linear's weight (kind 0, 2) or bias (kind 1) in the parameter list
    int32_t bn;                 // index of the BatchNorm weight (gamma; beta, mean, var follow), or -1
    int32_t src_off;            // element offset inside the parameter
    int32_t rows, cols;         // rows = output units (the BatchNorm channel of element (r, c) is unit0 + r)
    int32_t unit0;
    int32_t drs, dcs, srs, scs; // destination / source strides per row and per column
    int32_t plane;              // kind 2: out * 32, the piece stride inside a 32-deep k-chunk of [in/32][3][out][32]
    int32_t pad[3];
};
constexpr int kMaxPackSegs = 96;
struct PackProgram {
    BlobHeader header;
    int32_t n_segs;
    int32_t pad[3];
    PackSeg segs[kMaxPackSegs];
};
bool pack_program(const gnncca_mpn_dims* d, PackProgram* out);
bool enc_split_ok(const gnncca_mpn_dims* d);  // first encoder layer eligible for the split-bf16 MFMA GEMM

// Blob of the generic family: every layer of every MLP as W[out][in] + b[out], BatchNorm folded.
// MLP index: 0 encoder.node, 1 encoder.edge, 2 MPNet.edge_model, 3 MPNet.node_model, 4 classifier.edge
struct GenBlobHeader {
    uint32_t magic, abi_version, family, total_floats;
    int32_t w[5][GNNCCA_MAX_LAYERS];   // transposed, padded: [in][ceil8(out)]
    int32_t b[5][GNNCCA_MAX_LAYERS];   // [ceil8(out)]
    int32_t enc0_rowmajor;             // first node-encoder weight also as [out][in] (input of the MFMA GEMM), or 0
    int32_t step_w, step_w_floats;     // the fused step's per-edge weights in the order and shape it stages them in LDS (generic_fused.cuh), or 0
    int32_t pad[1];
};
int gen_enc0_ksplit(const gnncca_mpn_dims* d, int64_t n_nodes);  // split-K factor of that GEMM (0: not used)
bool gen_blob_header(const gnncca_mpn_dims* d, GenBlobHeader* out);
const gnncca_mlp& mlp_by_index(const gnncca_mpn_dims* d, int i);

struct GenWorkspace {
    size_t flags, blockflags, seg_ptr, col32, perm, cursor, row32o, col32o;
    size_t node[3], h0, edge[4], e0, partial, tab[2], total;   // tab: per-node projection tables of the fused step (generic_fused.cuh)
    int64_t node_w, edge_w;  // floats per row of the node / edge scratch buffers
};
GenWorkspace carve_generic(const gnncca_mpn_dims* d, int64_t n, int64_t e);

// gnncca_build_edges that also zeroes `zero_n` int32 words at `zero_ptr` (graph_build.hip; gnncca_frames_forward hands it the post stage's counters)
int build_edges_zeroing(const gnncca_frames* fr, const float* reid, int32_t reid_dim, int64_t n_nodes, int64_t n_edges, int32_t mode,
                        int64_t* edge_index_out, float* edge_attr_out, float* edge_labels_out, int32_t* zero_ptr, int64_t zero_n,
                        gnncca_stream_t stream);

Family classify(const gnncca_mpn_dims* d);
bool blob_header(const gnncca_mpn_dims* d, BlobHeader* out);  // false if unsupported
bool dims_valid(const gnncca_mpn_dims* d);
int mlp_param_count(const gnncca_mlp& m);

// Workspace carve-up (byte offsets, all multiples of 256).
struct Workspace {
    size_t flags, blockflags, seg_ptr, col32, perm, cursor, h0, act, partial, pd[2], psq[2], e, e0, rng, total;
    int ksplit;        // split-K factor of the first encoder GEMM
    int64_t e_stride;  // floats between two feature planes of the edge state
    int ell_S;         // padded edge-state layout: slots per node (multiple of 32), 0 = compact CSR order only
};
// Padded edge-state stride for big, nearly regular batches: a pure function of (dims, N, E) -- the host never reads the
// graph back; the plan kernel validates the degrees against it on the device (GNNCCA_GRAPH_IRREGULAR).
int ell_stride(const gnncca_mpn_dims* d, int64_t n, int64_t e);
// Split-K factor of the 256-row split-bf16 encoder GEMM (N >= 16 384): 1 = un-split with the fused epilogue, else 2 / 4 / 8
// partial slabs for the tail kernel.  One workgroup per CU (144 KB of LDS), so the cost is counted in ROUNDS of 256 workgroups.
int enc_lds_ksplit(int64_t n_nodes, int K);
Workspace carve(const gnncca_mpn_dims* d, int64_t n, int64_t e);

}  // namespace gnncca
