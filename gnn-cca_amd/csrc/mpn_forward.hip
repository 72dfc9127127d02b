// mpn_kernels.hip -- gfx950 (MI355X / CDNA4) kernels and the C-ABI forward of the GNN-CCA message-passing path.
//
// Algebra (derived from models/mpn.py:48,68-69,97-99): with the edge-MLP weight split by the
// cat order [x[row] | x[col] | e] and the node-MLP weight by [x[row] | e'],
//     P_src = h W_src^T + b_e,  P_dst = h W_dst^T,  Q = h W_nx^T + b_n            (per node, tiny)
//     e'[k] = ReLU(P_src[row k] + P_dst[col k] + W_ee e[k])                         (per edge, VALU)
//     m[k]  = ReLU(Q[row k] + W_ne e'[k])                                           (per edge, MFMA 32x32x2 f32)
//     h'[i] = agg_{k : row k = i} m[k]                                              (in-register, per segment)
// so no [E,70] / [E,38] concatenation is ever materialised.  The aggregation index is `row` (the SOURCE node),
// exactly as the reference does it (mpn.py:99).
//
// Data layout in HBM (all fp32):
//   edge state   e      : 6 feature planes [6][E_pad]  in ROW-SORTED edge order  -> coalesced 256-B wave loads
//   gather table Pd     : [N][8]   (P_dst, 32-B rows)                             -> L1/L2-resident random reads
//   segment table PsQ   : [N][40]  (P_src | pad | Q)                              -> wave-uniform reads
//   topology     seg_ptr: [N+1] int32 CSR offsets by source node;  col32 [E] int32 (sorted order)
// One wave owns (a share of) one source node's contiguous edge segment, so the per-destination reduction needs
// no atomics and is bitwise reproducible.
#include <cstdlib>

#include "common.cuh"
#include "wave_reduce.cuh"
#include "plan.cuh"
#include "encoder.cuh"
#include "enc_rows32.cuh"
#include "msg_bf16.cuh"
#include "step_general.cuh"
#include "step_fast.cuh"
#include "step_pipe.cuh"
#include "enc_f16.cuh"
#include "enc_f16_slices.cuh"
#include "generic_fused.cuh"
#include "generic.cuh"
#include "postprocess.cuh"
#include "backward.cuh"
#include "pack_device.cuh"
#include "train_generic.cuh"

using namespace gnncca;

extern "C" {

#ifdef GNNCCA_STAMPS
__attribute__((visibility("default"))) int gnncca_debug_set_stamps(void* dev_buf) {
    unsigned long long* p = static_cast<unsigned long long*>(dev_buf);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)));
    return GNNCCA_OK;
}
#endif

int gnncca_last_hip_error(void) { return g_last_hip_error; }

int gnncca_read_graph_flags(const void* workspace, uint32_t* flags_out, gnncca_stream_t stream) {
    if (!workspace || !flags_out) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(flags_out, workspace, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GNNCCA_OK;
}

int gnncca_read_graph_flags2(const void* workspace, uint32_t flags_out[2], gnncca_stream_t stream) {
    if (!workspace || !flags_out) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(flags_out, workspace, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GNNCCA_OK;
}

}  // extern "C"

static int forward_impl(const gnncca_mpn_dims* d, const void* packed_dev, const float* x, const int64_t* edge_index,
                        const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* workspace, size_t workspace_bytes,
                        float* logits_out, const gnncca_trace* trace, gnncca_stream_t stream, Profiler* prof,
                        uint32_t options, const gnncca_dropout* dropout = nullptr) {
    if (!dims_valid(d) || n_nodes < 0 || n_edges < 0) return GNNCCA_ERR_INVALID_ARG;
    const Family fam = classify(d);
    if (fam == kFamilyNone) return GNNCCA_ERR_UNSUPPORTED;
    if (n_nodes >= (1ll << 31) - 64 || n_edges >= (1ll << 31) - 64) return GNNCCA_ERR_UNSUPPORTED;
    if (n_nodes == 0) return n_edges == 0 ? GNNCCA_OK : GNNCCA_ERR_INVALID_ARG;
    if (!packed_dev || !x || !workspace) return GNNCCA_ERR_INVALID_ARG;
    if (n_edges > 0 && (!edge_index || !edge_attr || !logits_out)) return GNNCCA_ERR_INVALID_ARG;
    DropCfg drop;
    std::memset(&drop, 0, sizeof(drop));
    if (dropout && (dropout->p_enc > 0.f || dropout->p_edge > 0.f || dropout->p_node > 0.f || dropout->p_cls > 0.f)) {
        // train-mode Dropout lives in the general (traced) kernels of the MFMA family only
        if (fam != kFamilyMfma32x6 || !trace || !trace->h_enc || !trace->e_enc || !trace->h_steps || !trace->e_steps ||
            !dropout->seed_dev || d->enc_node.n_layers != 2)
            return fam == kFamilyMfma32x6 ? GNNCCA_ERR_INVALID_ARG : GNNCCA_ERR_UNSUPPORTED;
        const float ps[4] = {dropout->p_enc, dropout->p_edge, dropout->p_node, dropout->p_cls};
        for (float q : ps)
            if (!(q >= 0.f && q < 1.f)) return GNNCCA_ERR_INVALID_ARG;
        drop.p_enc = dropout->p_enc, drop.p_edge = dropout->p_edge, drop.p_node = dropout->p_node, drop.p_cls = dropout->p_cls;
        drop.seed = reinterpret_cast<const unsigned long long*>(dropout->seed_dev);
    }
    const bool dropping = drop.seed != nullptr;
    if (fam == kFamilyGeneric)
        return forward_generic(d, packed_dev, x, edge_index, edge_attr, n_nodes, n_edges, workspace, workspace_bytes,
                               logits_out, trace, static_cast<hipStream_t>(stream));
    const Workspace ws = carve(d, n_nodes, n_edges);
    if (workspace_bytes < ws.total) return GNNCCA_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(workspace);
    const int N = (int)n_nodes, E = (int)n_edges;
    const float* blob = static_cast<const float*>(packed_dev);

    // The blob header is a pure function of dims: recompute it on the host instead of reading it back.
    BlobHeader hdr;
    if (!blob_header(d, &hdr)) return GNNCCA_ERR_UNSUPPORTED;

    unsigned* flags = reinterpret_cast<unsigned*>(base + ws.flags);
    int* seg_ptr = reinterpret_cast<int*>(base + ws.seg_ptr);
    int* col32 = reinterpret_cast<int*>(base + ws.col32);
    int* perm = reinterpret_cast<int*>(base + ws.perm);
    int* cursor = reinterpret_cast<int*>(base + ws.cursor);
    unsigned* blockflags = reinterpret_cast<unsigned*>(base + ws.blockflags);
    float* h0 = reinterpret_cast<float*>(base + ws.h0);
    float* act = reinterpret_cast<float*>(base + ws.act);
    float* part = reinterpret_cast<float*>(base + ws.partial);
    float* pd[2] = {reinterpret_cast<float*>(base + ws.pd[0]), reinterpret_cast<float*>(base + ws.pd[1])};
    float* psq[2] = {reinterpret_cast<float*>(base + ws.psq[0]), reinterpret_cast<float*>(base + ws.psq[1])};
    float* ebuf = reinterpret_cast<float*>(base + ws.e);
    float* e0buf = reinterpret_cast<float*>(base + ws.e0);

    // big, nearly regular batches keep the edge state in the padded layout (StepParams::ell_S); the decision is a pure
    // function of (dims, N, E) made in carve(); the plan validates the degrees against it on the device
    const bool use_ell = ws.ell_S > 0 && hdr.fast_consts != 0 && d->edge_in == 4 && !trace && d->agg != GNNCCA_AGG_MAX &&
                         (reinterpret_cast<uintptr_t>(edge_attr) & 15) == 0 && E > 0;
    // ---- node encoder -------------------------------------------------------------------------------------
    const int nl = d->enc_node.n_layers;
    const int n_gemm = nl == 1 ? 1 : nl - 1;
    const float* cur_in = x;
    int ks_last = 1;
    bool fused_tail = false;   // the GEMM launch already produced h0 and the step-1 projections
    const bool split3 = (options & GNNCCA_OPT_ENC_SPLIT3) != 0;
    // graphs and batches whose GEMM ran split-K: the tail on the matrix pipe, 32 nodes per workgroup, from 2560 nodes (round 3 lowered this
    // from 6144: 8.7 -> 6.9 us at N = 4096, 10.3 -> 7.5 at 5120, 10.4 -> 7.0 at 4000, 9.2 -> 7.2 at 3000, 8.6 -> 6.4 at 3072; equal at 2048
    // (6.2), the register-resident tail ahead at 1024 (5.0 vs 7.6); profiles/r03_logs/r3_thresh1.log, r3_thresh2.log)
    static const bool no_mfma_tail = diag_env("GNNCCA_NO_MFMA_TAIL") != nullptr;  // diagnostics: A/B the two tails
    static const int kTailMfmaMin = diag_env("GNNCCA_TAIL_MFMA_MIN") ? std::atoi(diag_env("GNNCCA_TAIL_MFMA_MIN")) : 2560;
    const bool tail_mfma_ok = !dropping && !no_mfma_tail && N >= kTailMfmaMin && nl == 2 && d->enc_node.layers[0].out_dim == 128 &&
                              !d->reattach_nodes && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
    for (int g = 0; g < n_gemm; ++g) {
        const gnncca_layer& l = d->enc_node.layers[g];
        const int K = l.in_dim, O = l.out_dim;
        const int ks = g == 0 ? ws.ksplit : 1;
        int kslice = (K + ks - 1) / ks;
        kslice = (kslice + 63) / 64 * 64;
        // From 384 nodes on the first encoder layer runs on the split-bf16 MFMA GEMM, the plan riding in its launch.  (Round 3 lowered this
        // from 4096 to 1024 -- f32 MFMA GEMM + plan 19.5 us at dense1024, 43 at dense2048; 128-row split kernel + a plan launch 11.0 + 5.8
        // and 17.4 + 17.1, r3_gemm_split_min.log -- and, once the plan rode along, to 384: GEMM 11.5 / 12.1 / 17.7 / 20.8 -> 9.3 / 9.2 /
        // 10.5 / 13.2 us at 384 / 512 / 768 / 896 nodes, 3 x dense256 17.0 -> 10.0; at 256 nodes the f32 form stays ahead, 6.6 vs 7.6;
        // r3_split_min2.log)
        static const int split_min = diag_env("GNNCCA_GEMM_SPLIT_MIN") ? std::atoi(diag_env("GNNCCA_GEMM_SPLIT_MIN")) : 384;   // diagnostics
        // round 6: below 4096 nodes the first layer runs on the fp16-split GEMM in 32-row tiles with K split over one round of workgroups
        // (enc_f16_slices.cuh): a fraction of the slabs of the two forms above (dense1024: 8 x 0.5 MB instead of 16 x 0.5 MB written and read
        // back; dense256: 16 instead of 32) and half the matrix work of the six-product form.  nks: powers of two while one round of
        // workgroups holds the tiles (nrt * nks <= 384), a slice stays >= 128 deep and the workspace has the slabs.
        static const int slices_min = diag_env_int("GNNCCA_GEMM_SLICES_MIN", 1, 0, 0x7FFFFFFF);
        static const int slices_max = diag_env_int("GNNCCA_GEMM_SLICES_MAX", 4095, 0, 0x7FFFFFFF);
        static const bool slices_bf16 = diag_env("GNNCCA_GEMM_BF16") != nullptr;
        const bool use_slices = g == 0 && hdr.enc_w2h != 0 && O == 128 && K >= 64 && (K & (K - 1)) == 0 && K / kF16SlMaxKs <= ws.ksplit && N >= slices_min && N <= slices_max && !split3 &&
                                !slices_bf16 && (options & GNNCCA_OPT_ENC_UNSPLIT) == 0 && (reinterpret_cast<uintptr_t>(cur_in) & 15) == 0;
        const bool split = !use_slices && g == 0 && hdr.enc_w3 != 0 && N >= split_min && (reinterpret_cast<uintptr_t>(cur_in) & 15) == 0;
        EncPlanParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.in = cur_in;
        ep.W = blob + hdr.enc_node_w[g];
        ep.part = part;
        ep.M = N;
        ep.K = K;
        ep.O = O;
        ep.kslice = kslice;
        ep.vec_ok = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(cur_in) & 15) == 0);
        ep.nrt = (N + 31) / 32;
        ep.nks = ks;
        ep.gemm_blocks = split ? 0 : ep.nrt * ks * ((O + 127) / 128);
        int plan_blocks = 0;
        bool plan_launched = false;
        if (g == 0 && E > 0) {  // the graph plan rides in the first GEMM launch (small graphs) or gets its own (batches)
            ep.ei = reinterpret_cast<const long long*>(edge_index);
            ep.seg_ptr = seg_ptr;
            ep.col32 = col32;
            ep.blockflags = blockflags;
            ep.E = E;
            ep.N = N;
            ep.ell_S = use_ell ? ws.ell_S : 0;
            plan_blocks = plan_num_blocks(E);
        }
        int ks_split = 1;
        if (use_slices) {
            const int nrt = (N + 31) / 32;
            // slices: at least K / 256 (a workgroup's x tile is 32 KB of LDS), then powers of two while one round of workgroups holds the tiles, a
            // slice stays >= 64 deep and the workspace has the slabs (dense1024 -> 8, dense256 and below -> 16: 32 slices of 64 measured 0.7 us slower per dense256 forward, 8 of 256 1.4 us)
            static const int nks_cap = diag_env_int("GNNCCA_GEMM_SLICES_NKS", 16, 1, 32);       // diagnostics: cap of the slice count
            static const int wg_cap = diag_env_int("GNNCCA_GEMM_SLICES_WGS", 256, 1, 1 << 20);    // diagnostics: workgroups one round may hold
            int nks = 1;
            while (K / nks > kF16SlMaxKs) nks *= 2;
            while (nks * 2 <= std::min(nks_cap, ws.ksplit) && nrt * nks * 2 <= wg_cap && K / (nks * 2) >= 64 && (K / (nks * 2)) % 64 == 0) nks *= 2;
            EncF16SlicesParams q;
            std::memset(&q, 0, sizeof(q));
            q.x = cur_in;
            q.w2h = reinterpret_cast<const unsigned short*>(blob + hdr.enc_w2h);
            q.w_bad = reinterpret_cast<const unsigned*>(blob + hdr.enc_w2h_bad);
            q.w32 = blob + hdr.enc_node_w[0];
            q.part = part;
            q.M = N, q.K = K, q.nrt = nrt, q.nks = nks, q.Ks = K / nks;
            static const int sl_force_arm = diag_env_int("GNNCCA_GEMM_F16_ARM", 0, 0, 1);   // diagnostics / tests: every wave on the fp32 arm
            q.force_arm = sl_force_arm;
            EncPlanParams pl = ep;
            int ride = 0;
            if (plan_blocks > 0) {   // the plan rides in this launch (extra workgroups beyond the GEMM tiles)
                pl.plan_span = plan_span(edge_index, E);
                ride = pl.plan_span > 1 ? (plan_blocks + pl.plan_span - 1) / pl.plan_span : plan_blocks;
                ride = (ride + 1) / 2;   // two plan blocks per 512-thread workgroup
            } else {
                pl.E = 0;
            }
            const dim3 sgrid((unsigned)(nrt * nks + ride));
            if (q.Ks == 256)
                GNNCCA_LAUNCH(enc_gemm_f16_slices8_kernel, sgrid, dim3(kF16SlThreads), 0, st, q, pl);
            else if (q.Ks == 128)
                GNNCCA_LAUNCH(enc_gemm_f16_slices4_kernel, sgrid, dim3(kF16SlThreads), 0, st, q, pl);
            else
                GNNCCA_LAUNCH(enc_gemm_f16_slices2_kernel, sgrid, dim3(kF16SlThreads), 0, st, q, pl);
            HIP_TRY(hipGetLastError());
            PROF_MARK(GNNCCA_K_ENC_GEMM);
            plan_launched = true;
            ep.gemm_blocks = 0;
            plan_blocks = 0;
            ks_split = nks;
        }
        if (split) {  // big batches: split-bf16 MFMA GEMM; the plan gets its own launch
            const unsigned short* w3 = reinterpret_cast<const unsigned short*>(blob + hdr.enc_w3);
            static const bool force_direct = diag_env("GNNCCA_GEMM_DIRECT") != nullptr;  // diagnostics: A/B the GEMMs
            static const bool no_fuse = diag_env("GNNCCA_NO_FUSE") != nullptr;             // A/B against GEMM + tail launch
            static const int lds_min = diag_env("GNNCCA_GEMM_LDS_MIN") ? std::atoi(diag_env("GNNCCA_GEMM_LDS_MIN")) : 6144;
            const bool fusable = !no_fuse && !dropping && nl == 2 && d->enc_node.layers[1].in_dim == 128 && d->enc_node.layers[1].out_dim == kH &&
                                 !d->reattach_nodes && hdr.proj_wT != 0;
            const bool use_lds = N >= lds_min && O == 128 && !force_direct;
            if (use_lds) ks_split = std::min(enc_lds_ksplit(N, K), ws.ksplit);
            // mid-size batches, where the 256-row form would run split-K: 32-row workgroups, un-split, fused epilogue (enc_rows32.cuh)
            // GNNCCA_OPT_ENC_UNSPLIT: batches of >= 4096 nodes never split K -- where the 256-row form would, 32-row workgroups run
            // un-split with the same fused epilogue (enc_rows32.cuh), so a node's encoder output is bit for bit independent of the batch
            // around it (a shard of a sharded batch reproduces the union's logits exactly).  Not the default: one wave per SIMD at
            // N <= 8192 (N = 8192: 41-43.5 us against 28 + 9 for split-K + tail; enc_rows32.cuh says where the time goes)
            static const int r32_nst = diag_env("GNNCCA_GEMM_R32_NST") ? std::atoi(diag_env("GNNCCA_GEMM_R32_NST")) : 0;
            const bool use_r32 = (options & GNNCCA_OPT_ENC_UNSPLIT) != 0 && fusable && O == 128 && K % 256 == 0 && !force_direct && N >= 4096 &&
                                 !(use_lds && ks_split == 1);
            if (use_r32) ks_split = 1;
            // round 5: mid-size batches on the fp16-split form take 32-row workgroups that split K over their WAVES and finish the encoder in
            // the epilogue (enc_f16.cuh: enc_gemm_f16_rows32_kernel) -- no slabs, no tail launch.  Range: 4096 ... 8192 nodes, i.e. while the
            // row tiles fit ONE round of workgroups: every 32-row workgroup pulls all 1 MB of W through its CU's L2 path (~70 GB/s: 14 us), which
            // a second round doubles (plan + GEMM + tail, same box: 25.3 / 26.4 / 27.4 / 28.6 / 30.4 us at 4096 / 5120 / 6144 / 7168 / 8192 nodes
            // against 26.6 / 35.0 / 31.2 / 33.0 / 38.3 for the split-K forms; 47.4 against 39.2 at 9216; profiles/r05_logs/ab_r32f_2.log)
            static const int r32f_min = diag_env_int("GNNCCA_GEMM_R32F_MIN", 4096, 0, 0x7FFFFFFF);
            static const int r32f_max = diag_env_int("GNNCCA_GEMM_R32F_MAX", 8192, 0, 0x7FFFFFFF);
            static const bool gemm_bf16_early = diag_env("GNNCCA_GEMM_BF16") != nullptr;
            const bool use_r32f = !use_r32 && fusable && O == 128 && K % 256 == 0 && hdr.enc_w2h != 0 && !split3 && !gemm_bf16_early && !force_direct &&
                                  (options & GNNCCA_OPT_ENC_UNSPLIT) == 0 && N >= r32f_min && N <= r32f_max && !(use_lds && ks_split == 1);
            if (use_r32f) ks_split = 1;
            // un-split and the shipped encoder shape (2048 -> 128 -> 32, no reattach): the rest of the encoder, the step-1
            // projections and the plan fold run in the GEMM's epilogue, on the tile while it is on chip
            fused_tail = (use_lds && fusable && ks_split == 1) || use_r32 || use_r32f;
            EncFuseParams fp;
            std::memset(&fp, 0, sizeof(fp));
            static const int gemm_x_l2 = diag_env_int("GNNCCA_GEMM_X_L2_ROWS", 0, 1, 256);   // diagnostics: x served from L2 (timing only)
            fp.diag_x_rows = gemm_x_l2;
            // k rotation of the big-batch GEMMs (encoder.cuh / enc_f16.cuh): OFF.  A bare streaming loop of this access shape gains 20 % from it
            // (tools/ubench_xring.hip: 5.0 -> 6.2 TB/s), the kernels themselves 0-3 % at N = 65 536 and LOSE 4 % at 32 768 (their waves are
            // not in lockstep across CUs the way the microbenchmark's are; profiles/r05_logs/ab_krot{1,2}.log) -- and it makes a node's encoder
            // output depend on WHICH row block of the batch it sits in (the summation starts elsewhere), which the un-rotated kernels do not
            // (tests/test_gpu_fuzz.py: 512 copies of one graph, bitwise).  Kept behind GNNCCA_GEMM_KROT=1 for the record.
            static const bool k_rot = diag_env("GNNCCA_GEMM_KROT") != nullptr;
            fp.k_rotate = ((options & GNNCCA_OPT_ENC_UNSPLIT) == 0 && k_rot) ? 1 : 0;
            if (fused_tail) {
                fp.b1 = blob + hdr.enc_node_b[0];
                fp.W2 = blob + hdr.enc_node_w[1];
                fp.b2 = blob + hdr.enc_node_b[1];
                fp.projwT = blob + hdr.proj_wT;
                fp.projb = blob + hdr.proj_b;
                fp.h0 = h0;
                fp.trace_h = trace ? trace->h_enc : nullptr;
                fp.pd_out = pd[0];
                fp.psq_out = psq[0];
                fp.relu_prev = l.relu;
                // the plan goes FIRST here, so that one extra workgroup of the GEMM launch can fold its findings (and repair
                // an unsorted graph): no tail launch is left in this regime
                fp.ei = reinterpret_cast<const long long*>(edge_index);
                fp.seg_ptr = seg_ptr;
                fp.col32 = col32;
                fp.perm = perm;
                fp.cursor = cursor;
                fp.flags = flags;
                fp.blockflags = blockflags;
                fp.E = E;
                if (plan_blocks > 0) {
                    ep.plan_span = plan_span(edge_index, E);
                    if (ep.plan_span > 1) plan_blocks = (plan_blocks + ep.plan_span - 1) / ep.plan_span;
                    GNNCCA_LAUNCH(plan_only_kernel, dim3(plan_blocks), dim3(256), 0, st, ep);
                    HIP_TRY(hipGetLastError());
                    PROF_MARK(GNNCCA_K_PLAN_ROWS);
                }
                plan_launched = true;
            }
            // The fp16-split form (enc_f16.cuh) is the default of the 256-row regime; the bf16 form stays for GNNCCA_OPT_ENC_UNSPLIT (whose
            // bitwise batch independence is stated on the bf16 arithmetic of all three un-split kernels), for GNNCCA_OPT_ENC_SPLIT3 (an option
            // of that form) and as the A/B reference (GNNCCA_GEMM_BF16).  K / ks_split is a multiple of 32 on this path.
            static const bool gemm_bf16 = diag_env("GNNCCA_GEMM_BF16") != nullptr;
            static const int f16_force_arm = diag_env_int("GNNCCA_GEMM_F16_ARM", 0, 0, 1);   // diagnostics / tests: every tile on the bf16 arm
            // x stream of the fp16-split GEMMs: NON-TEMPORAL from 128 MB of x on (N >= 16 384 at K = 2048).  Inside a real forward the caches are full
            // of what the previous forward's step kernels left (dirty edge state), and a default-policy x stream fights it for every line: BASELINE
            // config 4 in situ 174 -> 137 us for this launch (0.516 -> 0.485 ms per forward), while an encoder timed alone on clean caches hides
            // the difference (121 us either way); at the 8 192-node share the policy costs 1 us, so small batches keep the default
            // (profiles/r05_logs/ab_config4_nt.log).  GNNCCA_GEMM_F16_XNT = 0 / 1 forces it off / on.
            static const int f16_x_nt_force = diag_env_int("GNNCCA_GEMM_F16_XNT", -1, -1, 1);
            const int f16_x_nt = f16_x_nt_force >= 0 ? f16_x_nt_force : ((double)N * K * 4.0 >= 128.0 * 1024 * 1024 ? 1 : 0);
            const bool use_f16 = use_lds && !use_r32 && !use_r32f && hdr.enc_w2h != 0 && !split3 && (options & GNNCCA_OPT_ENC_UNSPLIT) == 0 && !gemm_bf16 &&
                                 gemm_x_l2 == 0 && K % 32 == 0 && (K / ks_split) % 32 == 0;
            static thread_local int attr_dev = -1;  // once per device and thread: the attribute is per device
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            if (attr_dev != dev) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(enc_gemm_f16_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kF16LdsBytes));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(enc_gemm_f16_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kF16LdsBytes));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(enc_gemm_f16_rows32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kF16R32LdsBytes));
                const void* fns[8] = {reinterpret_cast<const void*>(enc_gemm_split_lds_kernel<false, false>),
                                      reinterpret_cast<const void*>(enc_gemm_split_lds_kernel<true, false>),
                                      reinterpret_cast<const void*>(enc_gemm_split_lds_kernel<false, true>),
                                      reinterpret_cast<const void*>(enc_gemm_split_lds_kernel<true, true>),
                                      reinterpret_cast<const void*>(enc_gemm_split_lds_kernel<false, false, true>),
                                      reinterpret_cast<const void*>(enc_gemm_split_lds_kernel<true, false, true>),
                                      reinterpret_cast<const void*>(enc_gemm_split_lds_kernel<false, true, true>),
                                      reinterpret_cast<const void*>(enc_gemm_split_lds_kernel<true, true, true>)};
                for (const void* fn : fns)
                    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsGemmBytes));
                attr_dev = dev;
            }
            // (Round 4 measured a 64-ROW split-K tiling for the mid-size batches -- 128 row blocks at 8192 nodes, split-K 4, two 64-KB-LDS
            // workgroups per CU, half the slabs: correct, and slower at every size: 32.1 vs 27.8 us at N = 8192 with a 7.0 instead of 9.3 us
            // tail, 59 vs 46 at 16 384, 129 vs 96 at 32 768 -- 4x the W traffic from L2 and one accumulation chain per wave; the kernel and
            // the log are kept under profiles/r04_logs/: enc_rows64.cuh.txt, ab_enc64_1.log.)
            if (use_r32f) {
                EncF16Params q;
                std::memset(&q, 0, sizeof(q));
                q.x = cur_in;
                q.w2h = reinterpret_cast<const unsigned short*>(blob + hdr.enc_w2h);
                q.w_bad = reinterpret_cast<const unsigned*>(blob + hdr.enc_w2h_bad);
                q.w3 = w3;
                q.M = N, q.K = K, q.kslice = K;
                q.k_rotate = fp.k_rotate;
                q.force_arm = f16_force_arm;
                q.x_nt = f16_x_nt;
                GNNCCA_LAUNCH(enc_gemm_f16_rows32_kernel, dim3((unsigned)((N + 31) / 32) + 1), dim3(kF16R32Threads), kF16R32LdsBytes, st, q, fp);
            } else if (use_r32) {
                const dim3 rgrid((unsigned)((N + 31) / 32) + 1);
                const int nst = r32_nst == 4 || r32_nst == 8 ? r32_nst : ((N + 31) / 32 <= 256 ? 8 : 4);
                if (nst == 8 && split3)
                    GNNCCA_LAUNCH((enc_gemm_rows32_fused_kernel<true, 8>), rgrid, dim3(256), 0, st, cur_in, w3, N, K, fp);
                else if (nst == 8)
                    GNNCCA_LAUNCH((enc_gemm_rows32_fused_kernel<false, 8>), rgrid, dim3(256), 0, st, cur_in, w3, N, K, fp);
                else if (split3)
                    GNNCCA_LAUNCH((enc_gemm_rows32_fused_kernel<true, 4>), rgrid, dim3(256), 0, st, cur_in, w3, N, K, fp);
                else
                    GNNCCA_LAUNCH((enc_gemm_rows32_fused_kernel<false, 4>), rgrid, dim3(256), 0, st, cur_in, w3, N, K, fp);
            } else if (use_lds && use_f16) {
                // round 5: the fp16-split GEMM (enc_f16.cuh) -- three piece products, x and W by LDS-DMA rings; same grids, same epilogues
                EncF16Params q;
                std::memset(&q, 0, sizeof(q));
                q.x = cur_in;
                q.w2h = reinterpret_cast<const unsigned short*>(blob + hdr.enc_w2h);
                q.w_bad = reinterpret_cast<const unsigned*>(blob + hdr.enc_w2h_bad);
                q.w3 = w3;
                q.out = part;
                q.M = N, q.K = K, q.kslice = K / ks_split;
                q.k_rotate = fp.k_rotate;
                q.force_arm = f16_force_arm;
                static const int f16_prio = diag_env_int("GNNCCA_GEMM_F16_PRIO", 0, 0, 1);   // diagnostics: static priority of waves 4-7 (measured: 122.9 vs 121 us at N = 65 536 -- off)
                q.prio_late_half = f16_prio;
                q.x_nt = f16_x_nt;
                const dim3 fgrid((N + 255) / 256 + 1, 1), sgrid((N + 255) / 256, ks_split);
#ifdef GNNCCA_F16_ABLATIONS
                static const int f16_diag = diag_env_int("GNNCCA_GEMM_F16_DIAG", 0, 0, 7);
                if (fused_tail && f16_diag) {
                    typedef void (*kfn)(const EncF16Params, const EncFuseParams);
                    static const kfn fns[8] = {nullptr, enc_gemm_f16_fused_diag1_kernel, enc_gemm_f16_fused_diag2_kernel, enc_gemm_f16_fused_diag3_kernel,
                                               enc_gemm_f16_fused_diag4_kernel, enc_gemm_f16_fused_diag5_kernel, enc_gemm_f16_fused_diag6_kernel, enc_gemm_f16_fused_diag7_kernel};
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fns[f16_diag]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kF16LdsBytes));
                    GNNCCA_LAUNCH(fns[f16_diag], fgrid, dim3(512), kF16LdsBytes, st, q, fp);
                } else
#endif
                if (fused_tail)
                    GNNCCA_LAUNCH(enc_gemm_f16_fused_kernel, fgrid, dim3(512), kF16LdsBytes, st, q, fp);
                else
                    GNNCCA_LAUNCH(enc_gemm_f16_split_kernel, sgrid, dim3(512), kF16LdsBytes, st, q, fp);
            } else if (use_lds) {
                // 256-row workgroups, both operands through LDS (144 KB: one workgroup per CU, 8 waves); split-K by whole
                // rounds of 256 workgroups (internal.h: enc_lds_ksplit)
                // (The plan does not ride in this launch.  As extra workgroups: every workgroup reserves the 144 KB of dynamic LDS, so the
                // plan's queue one per CU behind the GEMM tiles, 29 -> 45 us at N = 8192.  Inside the GEMM waves, a 1024-edge block per wave
                // under the first operands' round trip: the block's own two dependent round trips are then exposed in every workgroup,
                // 29.5 + 6.4 -> 37.6 us.  profiles/r03_logs/r3_ride2.log, r3_ride4.log.)
                // (Round 4 re-measured the plan BESIDE this GEMM on a side stream, fork / join by events, now under HIP-graph replay: 64 x dense128
                // 98 -> 108 us per forward, 96 x dense128 130 -> 140, 64 x dense256 239 -> 244 -- both kernels slow each other down (GEMM 26.5 ->
                // 29, plan 5.9 -> 10 us) and the cross-queue hand-offs cost more than the plan's launch; round 1 had found the same with eager
                // launches.  profiles/r04_logs/ab_planfork1.log)
                // (Round 4 also let the plan ride in the TAIL launch of this regime -- plan workgroups behind enc_tail_mfma_kernel's, the last one
                // to arrive folding: the bare co-run, with no arrival count at all, takes 14.4 us against 5.9 + 9.7 at N = 8192 and 35-37 against
                // 17.6 + 14.8 at 64 x dense256 (the plan's stream at the tail's 192-VGPR occupancy); the arrival count costs ~15 ns per plan
                // workgroup (one address, eight L2s: 22.9 us) and the release fence in front of it ~95 ns (an L2 write-back each: 117-120 us).
                // profiles/r04_logs/ab_tailride{1,2,3}.log)
                const dim3 fgrid((N + 255) / 256 + 1, 1), sgrid((N + 255) / 256, ks_split);
                static const bool gemm_pipe = diag_env("GNNCCA_GEMM_NOPIPE") == nullptr;   // diagnostics: A/B against the barrier-per-chunk form (encoder.cuh: PIPE)
                if (gemm_pipe && fused_tail && split3)
                    GNNCCA_LAUNCH((enc_gemm_split_lds_kernel<true, true, true>), fgrid, dim3(512), kLdsGemmBytes, st, cur_in, w3, part, N, K, O, K, fp);
                else if (gemm_pipe && fused_tail)
                    GNNCCA_LAUNCH((enc_gemm_split_lds_kernel<true, false, true>), fgrid, dim3(512), kLdsGemmBytes, st, cur_in, w3, part, N, K, O, K, fp);
                else if (gemm_pipe && split3)
                    GNNCCA_LAUNCH((enc_gemm_split_lds_kernel<false, true, true>), sgrid, dim3(512), kLdsGemmBytes, st, cur_in, w3, part, N, K, O,
                                  K / ks_split, fp);
                else if (gemm_pipe)
                    GNNCCA_LAUNCH((enc_gemm_split_lds_kernel<false, false, true>), sgrid, dim3(512), kLdsGemmBytes, st, cur_in, w3, part, N, K, O,
                                  K / ks_split, fp);
                else if (fused_tail && split3)
                    GNNCCA_LAUNCH((enc_gemm_split_lds_kernel<true, true>), fgrid, dim3(512), kLdsGemmBytes, st, cur_in, w3, part, N, K, O, K, fp);
                else if (fused_tail)
                    GNNCCA_LAUNCH((enc_gemm_split_lds_kernel<true, false>), fgrid, dim3(512), kLdsGemmBytes, st, cur_in, w3, part, N, K, O, K, fp);
                else if (split3)
                    GNNCCA_LAUNCH((enc_gemm_split_lds_kernel<false, true>), sgrid, dim3(512), kLdsGemmBytes, st, cur_in, w3, part, N, K, O,
                                  K / ks_split, fp);
                else
                    GNNCCA_LAUNCH((enc_gemm_split_lds_kernel<false, false>), sgrid, dim3(512), kLdsGemmBytes, st, cur_in, w3, part, N, K, O,
                                  K / ks_split, fp);
            } else {
                // 128-row workgroups (a wave = 32 rows x 128 columns); split-K until >= 512 workgroups are in flight
                static const int direct_wg = diag_env("GNNCCA_GEMM_DIRECT_WG") ? std::atoi(diag_env("GNNCCA_GEMM_DIRECT_WG")) : 512;   // diagnostics
                while (ks_split < ws.ksplit && ((N + 127) / 128) * ks_split < direct_wg && (K / (ks_split * 2)) % 32 == 0) ks_split *= 2;
                const int rb = (N + 127) / 128;
                // the plan rides in this launch (extra workgroups beyond the GEMM tiles)
                EncPlanParams pl = ep;
                int ride = 0;
                static const bool no_ride_d = diag_env("GNNCCA_NO_RIDE") != nullptr;   // diagnostics: A/B against a plan launch of its own
                if (plan_blocks > 0 && (N < 4096 || !no_ride_d)) {
                    pl.plan_span = plan_span(edge_index, E);   // narrow (0) or pair form by the edge count and alignment
                    ride = pl.plan_span > 1 ? (plan_blocks + pl.plan_span - 1) / pl.plan_span : plan_blocks;
                    plan_launched = true;
                } else {
                    pl.E = 0;
                }
                const dim3 dgrid((unsigned)(rb * ks_split + ride));
                if (split3)
                    GNNCCA_LAUNCH(enc_gemm_split_direct_kernel<true>, dgrid, dim3(256), 0, st, cur_in, w3, part, N, K, O, K / ks_split, rb, ks_split, pl);
                else
                    GNNCCA_LAUNCH(enc_gemm_split_direct_kernel<false>, dgrid, dim3(256), 0, st, cur_in, w3, part, N, K, O, K / ks_split, rb, ks_split, pl);
            }
            HIP_TRY(hipGetLastError());
            PROF_MARK(GNNCCA_K_ENC_GEMM);
        }
        if (ep.gemm_blocks + plan_blocks > 0 && !plan_launched) {
            if (ep.gemm_blocks == 0 && plan_blocks > 0) {   // a plan-only launch (big batches): several blocks per workgroup
                ep.plan_span = plan_span(edge_index, E);
                if (ep.plan_span > 1) plan_blocks = (plan_blocks + ep.plan_span - 1) / ep.plan_span;
            }
            if (ep.gemm_blocks == 0)
                GNNCCA_LAUNCH(plan_only_kernel, dim3(plan_blocks), dim3(256), 0, st, ep);
            else
                GNNCCA_LAUNCH(enc_gemm_plan_kernel, dim3(ep.gemm_blocks + plan_blocks), dim3(256), 0, st, ep);
            HIP_TRY(hipGetLastError());
            PROF_MARK(split ? GNNCCA_K_PLAN_ROWS : GNNCCA_K_ENC_GEMM);
        }
        ks_last = (split || use_slices) ? ks_split : ks;
        if (g < n_gemm - 1) {
            float* dst = act + (size_t)(g & 1) * N * O;
            GNNCCA_LAUNCH(reduce_bias_act_kernel, grid1((size_t)N * O, 256), dim3(256), 0, st, (const float*)part,
                               blob + hdr.enc_node_b[g], dst, N, O, ks, l.relu);
            HIP_TRY(hipGetLastError());
            PROF_MARK(GNNCCA_K_ENC_REDUCE);
            cur_in = dst;
        }
    }
    const int nf = d->reattach_nodes ? 2 : 1;
    const int hin = nf * kH;
    {
        const gnncca_layer& lprev = d->enc_node.layers[n_gemm - 1];
        TailParams tp;
        std::memset(&tp, 0, sizeof(tp));
        tp.blob = blob;
        tp.part = part;
        tp.h0 = h0;
        tp.trace_h = trace ? trace->h_enc : nullptr;
        tp.pd_out = pd[0];
        tp.psq_out = psq[0];
        tp.off_prev_b = hdr.enc_node_b[n_gemm - 1];
        tp.off_lastWT = hdr.enc_last_wT;
        tp.off_last_b = hdr.enc_node_b[nl - 1];
        tp.off_projwT = hdr.proj_wT;
        tp.off_projb = hdr.proj_b;
        tp.ks = ks_last;
        tp.F = lprev.out_dim;
        tp.N = N;
        tp.has_last = nl >= 2;
        tp.relu_prev = lprev.relu;
        tp.reatt_n = d->reattach_nodes;
        tp.hin = hin;
        tp.vec_reduce = (tp.F % 4 == 0) && (tp.F / 4 <= 64) && (64 % (tp.F / 4) == 0);
        const size_t lds = ((size_t)hin * kProjOut + (tp.has_last ? (size_t)tp.F * kH : 0) + 4 * (size_t)tp.F + 4 * 256) * sizeof(float);
        if (lds > 160 * 1024) return GNNCCA_ERR_UNSUPPORTED;
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(enc_tail_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        tp.ei = reinterpret_cast<const long long*>(edge_index);
        tp.seg_ptr = seg_ptr;
        tp.col32 = col32;
        tp.perm = perm;
        tp.cursor = cursor;
        tp.flags = flags;
        tp.blockflags = blockflags;
        tp.E = E;
        tp.drop = drop;
        const unsigned blocks = (unsigned)std::min<size_t>(((size_t)N + 3) / 4, 2048) + 1;  // + plan-repair workgroup
        // register-resident tail in the latency-bound regime only: on big batches it is VALU-bound (readlane traffic) and
        // measured 15 % slower than the LDS form (72 vs 62 us at N = 65 536)
        const bool tail_mfma = tail_mfma_ok && !fused_tail;
        const bool tail_fast = !tail_mfma && !dropping && N < 4096 && tp.F == 128 && tp.has_last && !tp.reatt_n && tp.trace_h == nullptr && tp.vec_reduce &&
                               (reinterpret_cast<uintptr_t>(part) & 15) == 0;
        if (fused_tail) {
            // nothing: h0, the projections and the plan's flag word all came out of the GEMM launch
        } else if (tail_fast) {
            static const int tail_npw = diag_env("GNNCCA_TAIL_NPW") ? std::atoi(diag_env("GNNCCA_TAIL_NPW")) : 0;   // diagnostics
            // nodes per workgroup: one up to 320 nodes, two up to 640, four beyond (a block of 200 forwards in one HIP graph: dense256 28.07 ->
            // 27.4 us per forward with one, dense64 24.7 -> 24.1; dense512 39.97 -> 39.5-39.8 with two; dense1024 best with four;
            // profiles/r03_logs/r3_tailnpw1.log)
            tp.npw = tail_npw == 1 || tail_npw == 2 || tail_npw == 4 ? tail_npw : (N <= 320 ? 1 : (N <= 640 ? 2 : 4));
            const unsigned fblocks = (unsigned)std::min<size_t>(((size_t)N + tp.npw - 1) / tp.npw, 2048) + 1;
            GNNCCA_LAUNCH(enc_tail_fast_kernel, dim3(fblocks), dim3(256), 0, st, tp);
        }
        else if (tail_mfma)
            GNNCCA_LAUNCH(enc_tail_mfma_kernel, dim3((unsigned)((N + 31) / 32) + 1), dim3(256), 0, st, tp,
                          blob + hdr.enc_node_w[nl - 1]);
        else
            GNNCCA_LAUNCH(enc_tail_kernel, dim3(blocks), dim3(256), std::max<size_t>(lds, 4096), st, tp);
        HIP_TRY(hipGetLastError());
        if (!fused_tail) PROF_MARK(GNNCCA_K_ENC_TAIL);
    }
    if (E == 0) return GNNCCA_OK;

    // ---- message passing steps ----------------------------------------------------------------------------
    const int L = d->num_enc_steps;
    const int first_cls = L - d->num_class_steps + 1;  // models/mpn.py:277
    const long long avg_deg = (E + (long long)N - 1) / N;
    const int chunks = (int)((avg_deg + 63) / 64);
    StepParams sp;
    std::memset(&sp, 0, sizeof(sp));
    sp.blob = blob;
    sp.seg_ptr = seg_ptr;
    sp.col32 = col32;
    sp.perm = perm;
    sp.flags = flags;
    sp.edge_attr = edge_attr;
    sp.e = ebuf;
    sp.e0 = e0buf;
    sp.h0 = h0;
    sp.e_stride = ws.e_stride;
    sp.off_wee = hdr.wee;
    sp.off_wneb = hdr.wne_b;
    sp.off_projwT = hdr.proj_wT;
    sp.off_projb = hdr.proj_b;
    sp.off_encw = hdr.enc_edge_w;
    sp.off_encb = hdr.enc_edge_b;
    sp.off_cw1 = hdr.cls_w1;
    sp.off_cb1 = hdr.cls_b1;
    sp.off_cw2 = hdr.cls_w2;
    sp.off_cb2 = hdr.cls_b2;
    sp.off_fast = hdr.fast_consts;
    sp.off_wnebf = hdr.wne_bf16;
    static const bool step_r2 = diag_env("GNNCCA_STEP_R2") != nullptr;   // diagnostics: A/B against round 2's step kernel
    // which arithmetic the node message uses (StepParams::msg_f32): the traced (general) and the fast kernels follow ONE rule
    sp.msg_f32 = (N <= 512 || step_r2 || !step_pipe_fits(N, E, ws.e_stride, ws.total)) ? 1 : 0;
    sp.cls_hidden = hdr.cls_hidden;
    sp.N = N;
    sp.E = E;
    sp.edge_in = d->edge_in;
    sp.attr_vec = d->edge_in == 4 && (reinterpret_cast<uintptr_t>(edge_attr) & 15) == 0;
    sp.agg = d->agg;
    sp.reatt_n = d->reattach_nodes;
    // waves per source-node segment: split a segment over 2 or 4 waves only while that is needed to put ~4 waves on
    // every SIMD (small graphs are latency-bound); big batches keep one wave per node, which amortises the
    // per-node prologue / projection epilogue over all of the node's chunks
    int npw_first = 1, npw_later = 1;   // nodes per wave of step 1 / of the later message steps (step_pipe.cuh: NPW)
    {
        // waves per node that would fill the chip; graphs whose nodes average >= 32 chunks (degree ~2000+) get four times that: the
        // tail of a launch is then a few long segments (dense3000: 0.4315 -> 0.4039 ms with four waves per node, dense2048 equal;
        // profiles/r03_logs/r3_wps1.log)
        const long long want = (chunks >= 32 ? 16384 : 4096) / (long long)N;
        int wps = chunks >= 4 ? 4 : (chunks >= 2 ? 2 : 1);
        while (wps > 1 && wps > want) wps >>= 1;
        // GNNCCA_OPT_ENC_UNSPLIT promises logits that do not depend on the batch around a graph for batches of >= 4096 nodes: the
        // cross-wave combine sums in another order with another wave count, and `want` above depends on N (a shard of 2 x dense2048
        // would get four waves per node, the union of 8 one), so the option pins one wave per node there, the rule of every round
        // before the four-wave one (tests/test_gpu_sharded.py: dense2048 shard against its union, bitwise)
        if ((options & GNNCCA_OPT_ENC_UNSPLIT) != 0 && N >= 4096) wps = 1;
        static const int force_wps = diag_env_int("GNNCCA_WPS", 0, 1, 4);  // diagnostics
        if (force_wps == 1 || force_wps == 2 || force_wps == 4) wps = std::min(force_wps, chunks >= 4 ? 4 : (chunks >= 2 ? 2 : 1));
        sp.wps = wps;
        // two nodes per wave (step_pipe.cuh: NPW): batches whose nodes average one round (<= 128 edges) and that still fill the chip with
        // half as many waves
        static const int force_npw = diag_env_int("GNNCCA_NPW", 0, 1, 2);   // diagnostics: 1 / 2 = never / whenever eligible
        // thresholds: from 16 384 nodes (step 1 at 8192: 14.9 -> 15.7 us; at 16 384: 29.8 -> 28.2).  The later steps alone from 8192 nodes, with
        // the classification deferred, move 64 x dense128 from 17.2 / 15.3 / 6.4 to 14.5 / 14.6 / 8.9 us per step: a tie per forward
        // (profiles/r04_logs/ab_npw{6,7}.log), so one threshold serves both
        static const int npw_min_n = diag_env_int("GNNCCA_NPW_MIN_N", 16384, 0, 0x7FFFFFFF);
        static const int npw_min_n_first = diag_env_int("GNNCCA_NPW_MIN_N_FIRST", 16384, 0, 0x7FFFFFFF);
        static const int npw_max_chunks = diag_env_int("GNNCCA_NPW_MAX_CHUNKS", 2, 1, 64);   // diagnostics: average 64-edge chunks per node up to which it is used
        npw_later = (wps == 1 && chunks <= npw_max_chunks && (force_npw == 2 || (force_npw == 0 && N >= npw_min_n))) ? 2 : 1;
        npw_first = (npw_later == 2 && (force_npw == 2 || N >= npw_min_n_first)) ? 2 : 1;
        sp.npw = npw_later;
    }
    sp.hin = hin;
    // column ranges instead of the col32 stream on steps 2 ... L of the specialised kernels (StepParams::rng)
    static const bool step_norange = diag_env("GNNCCA_STEP_NORANGE") != nullptr;   // diagnostics: A/B against streaming col32 on every step
    static const int range_max_e = diag_env_int("GNNCCA_RANGE_MAX_E", 0x7FFFFFFF, 0, 0x7FFFFFFF);   // diagnostics: edge count up to which the buffer-addressed kernel uses them
    sp.rng = (L >= 2 && !step_norange && (options & GNNCCA_OPT_COLUMN_RANGES) != 0 && (sp.msg_f32 || E <= range_max_e)) ? reinterpret_cast<int*>(base + ws.rng) : nullptr;
    sp.ws_base = base;
    sp.ws_bytes = ws.total;
    sp.so_e = (unsigned)ws.e, sp.so_col = (unsigned)ws.col32, sp.so_perm = (unsigned)ws.perm;
    // The P_dst gather table staged whole in LDS: OFF by default since round 4.  Rounds 1-2 staged it for every N <= 1024; round 3 found
    // gathers straight from L2 ahead below 801 nodes (dense32 ... dense768 -1.3 ... -4 % per forward, r3_pdlds2.log, r3_pdlds3.log) and
    // kept it for 801 ... 1024; with four waves per node and the kernels as they are now the table loses there as well -- dense1024,
    // L = 8 (BASELINE config 5): 120.4 -> 118.3 us per forward with the fp32 edge state, 112.8 -> 110.1 with bf16
    // (profiles/r04_logs/ab_config5.log; with ONE wave per node the table still wins, 141 vs 151 us, but that form is slower anyway).
    // The switches keep the variant reachable for A/B runs: GNNCCA_PD_LDS_MIN / _MAX name the node range that stages it.
    static const int pd_lds_max = diag_env_int("GNNCCA_PD_LDS_MAX", 1024, 0, 1024);   // diagnostics
    static const int pd_lds_min = diag_env_int("GNNCCA_PD_LDS_MIN", 0x7FFFFFFF, 0, 0x7FFFFFFF);
    sp.pd_lds = (N >= pd_lds_min && N <= pd_lds_max) && d->num_enc_steps > 0;
    sp.e_bf16 = (options & GNNCCA_OPT_EDGE_STATE_BF16) != 0;  // honoured by the specialised kernels only
    sp.ell_S = use_ell ? ws.ell_S : 0;
    sp.drop = drop;
    {
        static const bool no_nt = diag_env("GNNCCA_NO_NT") != nullptr;  // diagnostics: A/B the cache policy
        const double state_bytes = (double)(sp.e_bf16 ? kEF / 2 : kEF) * (double)ws.e_stride * 4.0;
        sp.nt_store = !no_nt && state_bytes > 150e6;
        sp.nt_load = !no_nt && state_bytes > 256e6;
        static const int force_nt = diag_env_int("GNNCCA_STEP_NT", -1, -1, 2);   // diagnostics: 0 / 1 / 2 = default policy / nt stores / nt loads too
        if (force_nt >= 0) sp.nt_store = force_nt >= 1, sp.nt_load = force_nt >= 2;
    }
    const bool re = d->reattach_edges != 0;
    int out_idx = 0;
    if (L == 0) {  // models/mpn.py:295-297: classify the encoded edge features once
        sp.first = 1;
        sp.update = 0;
        sp.cls_layers = hdr.cls_layers;
        sp.logits = logits_out;
        sp.trace_e_enc = trace ? trace->e_enc : nullptr;
        HIP_TRY(re ? (launch_step<true, false>(sp, st)) : (launch_step<false, false>(sp, st)));
        PROF_MARK(GNNCCA_K_STEP_LAST);
        return GNNCCA_OK;
    }
    // DEFERRED classification (step_pipe.cuh: CIN): a message step that produces a classified state does not classify it; the step that reads
    // the state back does (the last step: its input and its output).  The classifying message steps then run the variant without the
    // classifier -- lighter, and eligible for two nodes per wave.  Where it pays: forwards whose message steps run two nodes per wave
    // (sp.npw == 2); fp32 edge state only (the bf16 state is rounded after its step classified it); logits bit for bit the same.
    static const int force_defer = diag_env_int("GNNCCA_DEFER_CLS", -1, 0, 1);   // diagnostics: 0 / 1 = never / whenever possible
    const bool can_defer = hdr.fast_consts != 0 && sp.attr_vec && !trace && d->agg != GNNCCA_AGG_MAX && !sp.msg_f32 && !sp.e_bf16 && !sp.pd_lds &&
                           sp.rng == nullptr && first_cls < L && !dropping;
    const bool defer = can_defer && (force_defer == 1 || (force_defer != 0 && npw_later == 2));
    for (int step = 1; step <= L; ++step) {
        const bool want_h = trace && trace->h_steps;
        const bool msg = step < L || want_h;
        sp.first = step == 1;
        sp.npw = step == 1 ? npw_first : npw_later;
        sp.step_no = step;
        sp.cls_no = out_idx;
        sp.stamp_slot = 2 + (step - 1 < 6 ? step - 1 : 5);
        sp.update = 1;
        sp.store_e = step < L;
        if (defer) {   // slot of step s: s - first_cls
            sp.cls_layers = step == L ? hdr.cls_layers : 0;
            sp.logits = step == L ? logits_out + (size_t)(L - first_cls) * E : nullptr;
            sp.logits_in = step - 1 >= first_cls ? logits_out + (size_t)(step - 1 - first_cls) * E : nullptr;
        } else {
            sp.cls_layers = step >= first_cls ? hdr.cls_layers : 0;
            sp.logits = step >= first_cls ? logits_out + (size_t)(out_idx++) * E : nullptr;
            sp.logits_in = nullptr;
        }
        sp.pd_in = pd[(step - 1) & 1];
        sp.so_pd = (unsigned)ws.pd[(step - 1) & 1];
        sp.psq_in = psq[(step - 1) & 1];
        sp.pd_out = step < L ? pd[step & 1] : nullptr;
        sp.psq_out = step < L ? psq[step & 1] : nullptr;
        sp.trace_e_enc = (trace && step == 1) ? trace->e_enc : nullptr;
        sp.trace_e = (trace && trace->e_steps) ? trace->e_steps + (size_t)(step - 1) * E * kEF : nullptr;
        sp.trace_h = want_h ? trace->h_steps + (size_t)(step - 1) * N * kH : nullptr;
        hipError_t err;
        const bool fast = hdr.fast_consts != 0 && sp.attr_vec && !trace && d->agg != GNNCCA_AGG_MAX;
        static const bool step_nomem = diag_env("GNNCCA_STEP_NOMEM") != nullptr;   // diagnostics: arithmetic-only timing of the step kernel
        static const bool step_noepi = diag_env("GNNCCA_STEP_NOEPI") != nullptr;   // diagnostics: what the projection epilogue costs
        static const bool step_nohook = diag_env("GNNCCA_STEP_NOHOOK") != nullptr;   // diagnostics: second round's state requested after the first round
        static const bool step_earlybar = diag_env("GNNCCA_STEP_EARLYBAR") != nullptr;   // diagnostics: the staging barrier in front of the loop even with several waves per node
        sp.diag = (step_nomem ? 1 : 0) | (step_noepi ? 2 : 0) | (step_nohook ? 4 : 0) | (step_earlybar ? 8 : 0);
        const bool pipe_ok = fast && !sp.msg_f32;
        if (pipe_ok)
            err = launch_pipe_dispatch(sp, msg, st);
        else if (fast)
            err = launch_fast_dispatch(sp, msg, st);
        else if (re)
            err = msg ? launch_step<true, true>(sp, st) : launch_step<true, false>(sp, st);
        else
            err = msg ? launch_step<false, true>(sp, st) : launch_step<false, false>(sp, st);
        HIP_TRY(err);
        PROF_MARK(msg ? GNNCCA_K_STEP : GNNCCA_K_STEP_LAST);
    }
    return GNNCCA_OK;
}

extern "C" {

int gnncca_mpn_forward(const gnncca_mpn_dims* d, const void* packed_dev, const float* x, const int64_t* edge_index,
                       const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* workspace, size_t workspace_bytes,
                       float* logits_out, const gnncca_trace* trace, gnncca_stream_t stream) {
    return forward_impl(d, packed_dev, x, edge_index, edge_attr, n_nodes, n_edges, workspace, workspace_bytes, logits_out,
                        trace, stream, nullptr, 0u);
}

int gnncca_mpn_forward_ex(const gnncca_mpn_dims* d, const void* packed_dev, const float* x, const int64_t* edge_index,
                          const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* workspace, size_t workspace_bytes,
                          float* logits_out, const gnncca_trace* trace, uint32_t options, gnncca_stream_t stream) {
    return forward_impl(d, packed_dev, x, edge_index, edge_attr, n_nodes, n_edges, workspace, workspace_bytes, logits_out,
                        trace, stream, nullptr, options);
}

int gnncca_mpn_forward_train(const gnncca_mpn_dims* d, const void* packed_dev, const float* x, const int64_t* edge_index,
                             const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* workspace, size_t workspace_bytes,
                             float* logits_out, const gnncca_trace* trace, const gnncca_dropout* dropout, gnncca_stream_t stream) {
    return forward_impl(d, packed_dev, x, edge_index, edge_attr, n_nodes, n_edges, workspace, workspace_bytes, logits_out,
                        trace, stream, nullptr, 0u, dropout);
}

size_t gnncca_mlp_eval_workspace_bytes(const gnncca_mlp* mlp, int64_t rows) {
    if (!mlp_shape_ok(mlp) || rows < 0) return 0;
    return mlp_eval_ws(mlp, rows, nullptr, nullptr, nullptr, nullptr);
}

int gnncca_mlp_eval(const gnncca_mlp* mlp, const float* const* params_dev, int n_params, const float* in, int64_t rows, float* out,
                    void* workspace, size_t workspace_bytes, gnncca_stream_t stream) {
    return mlp_eval_impl(mlp, params_dev, n_params, in, rows, out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int gnncca_gather_cat(const float* a, const int64_t* ia, int wa, int64_t rows_a, const float* b, const int64_t* ib, int wb, int64_t rows_b,
                      const float* c, const int64_t* ic, int wc, int64_t rows_c, int64_t rows, float* out, gnncca_stream_t stream) {
    if (rows < 0 || wa < 0 || wb < 0 || wc < 0 || wa + wb + wc <= 0 || !out) return GNNCCA_ERR_INVALID_ARG;
    if ((wa > 0 && (!a || rows_a <= 0)) || (wb > 0 && (!b || rows_b <= 0)) || (wc > 0 && (!c || rows_c <= 0))) return rows == 0 ? GNNCCA_OK : GNNCCA_ERR_INVALID_ARG;
    if (rows == 0) return GNNCCA_OK;
    hipLaunchKernelGGL(tr_cat64_kernel, grid1((size_t)rows * (wa + wb + wc), 256), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       reinterpret_cast<const long long*>(ia), wa, (long long)rows_a, b, reinterpret_cast<const long long*>(ib), wb,
                       (long long)rows_b, c, reinterpret_cast<const long long*>(ic), wc, (long long)rows_c, out, (long long)rows);
    HIP_TRY(hipGetLastError());
    return GNNCCA_OK;
}

int gnncca_pad_frame(const float* x, int64_t n_nodes, const int64_t* edge_index, const float* edge_attr, int64_t n_edges, float* x_pad,
                     int64_t n_real_max, int n_dummy, int64_t* edge_index_pad, float* edge_attr_pad, int64_t e_pad, int node_in, int edge_in,
                     gnncca_stream_t stream) {
    if (n_nodes < 0 || n_edges < 0 || n_real_max < n_nodes || e_pad < n_edges || n_dummy < 1 || node_in < 1 || edge_in < 1) return GNNCCA_ERR_INVALID_ARG;
    if (!x_pad || !edge_index_pad || !edge_attr_pad || (n_nodes > 0 && !x) || (n_edges > 0 && (!edge_index || !edge_attr))) return GNNCCA_ERR_INVALID_ARG;
    const long long total = (n_real_max + n_dummy) * (long long)node_in + e_pad * (2ll + edge_in);
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pad_frame_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long long)n_nodes,
                       reinterpret_cast<const long long*>(edge_index), edge_attr, (long long)n_edges, x_pad, (long long)n_real_max, n_dummy,
                       reinterpret_cast<long long*>(edge_index_pad), edge_attr_pad, (long long)e_pad, node_in, edge_in);
    HIP_TRY(hipGetLastError());
    return GNNCCA_OK;
}

size_t gnncca_aggregate_workspace_bytes(int64_t n_nodes, int64_t n_edges) { return agg_ws(n_nodes, n_edges).total; }

int gnncca_aggregate(const float* messages, const int64_t* edge_index, int64_t n_nodes, int64_t n_edges, int width, int agg, float* out,
                     void* workspace, size_t workspace_bytes, gnncca_stream_t stream) {
    return aggregate_impl(messages, edge_index, n_nodes, n_edges, width, agg, out, workspace, workspace_bytes,
                          static_cast<hipStream_t>(stream));
}

size_t gnncca_train_tape_bytes(const gnncca_mpn_dims* d, int64_t n_nodes, int64_t n_edges) {
    TrPlan P;
    if (!d || n_nodes < 0 || n_edges < 0 || !tr_plan(d, n_nodes, n_edges, &P)) return 0;
    return P.total;
}

int gnncca_train_tape_latents(const gnncca_mpn_dims* d, int64_t n_nodes, int64_t n_edges, int64_t* offsets_out, int n_offsets) {
    TrPlan P;
    if (!d || !offsets_out || n_nodes < 0 || n_edges < 0 || !tr_plan(d, n_nodes, n_edges, &P)) return GNNCCA_ERR_INVALID_ARG;
    const int L = d->num_enc_steps;
    if (n_offsets != 2 + 2 * L) return GNNCCA_ERR_INVALID_ARG;
    auto last = [](const TrCall& c, const gnncca_mlp& m) -> int64_t { return m.n_layers > 0 ? (int64_t)c.lay[m.n_layers - 1].a : -1; };
    offsets_out[0] = last(P.enc_node, d->enc_node);
    offsets_out[1] = last(P.enc_edge, d->enc_edge);
    for (int s = 0; s < L; ++s) {
        offsets_out[2 + 2 * s] = (int64_t)P.h[s];
        offsets_out[3 + 2 * s] = last(P.edge[s], d->edge_mlp);
    }
    return GNNCCA_OK;
}

int gnncca_train_forward(const gnncca_mpn_dims* d, float* const* params_dev, int n_params, const float* x, const int64_t* edge_index,
                         const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* tape, size_t tape_bytes, float* logits_out,
                         const gnncca_dropout* dropout, gnncca_stream_t stream) {
    if (!x || (n_edges > 0 && (!edge_index || !edge_attr || !logits_out))) return GNNCCA_ERR_INVALID_ARG;
    return train_forward_impl(d, params_dev, n_params, x, edge_index, edge_attr, n_nodes, n_edges, tape, tape_bytes, logits_out, dropout,
                              static_cast<hipStream_t>(stream));
}

int gnncca_train_backward(const gnncca_mpn_dims* d, float* const* params_dev, int n_params, const float* x, const int64_t* edge_index,
                          const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* tape, size_t tape_bytes,
                          const float* grad_logits, float* const* grads_dev, const gnncca_dropout* dropout, gnncca_stream_t stream) {
    if (!x || (n_edges > 0 && (!edge_index || !edge_attr))) return GNNCCA_ERR_INVALID_ARG;
    return train_backward_impl(d, params_dev, n_params, x, edge_index, edge_attr, n_nodes, n_edges, tape, tape_bytes, grad_logits,
                               grads_dev, dropout, static_cast<hipStream_t>(stream));
}

int gnncca_mpn_forward_profiled(const gnncca_mpn_dims* d, const void* packed_dev, const float* x,
                                const int64_t* edge_index, const float* edge_attr, int64_t n_nodes, int64_t n_edges,
                                void* workspace, size_t workspace_bytes, float* logits_out, gnncca_stream_t stream,
                                gnncca_profile* profile) {
    if (!profile) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Profiler p;
    p.out = profile;
    profile->count = 0;
    int s = prof_begin(&p);
    if (s != GNNCCA_OK) return s;
    s = forward_impl(d, packed_dev, x, edge_index, edge_attr, n_nodes, n_edges, workspace, workspace_bytes, logits_out,
                     nullptr, stream, &p, profile->options);
    const int s2 = prof_end(&p, st);
    return s != GNNCCA_OK ? s : s2;
}


// ---- SURVEY.md 8f row N2 ------------------------------------------------------------------------------------
size_t gnncca_post_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
    if (n_nodes < 0 || n_edges < 0) return 0;
    const size_t N = (size_t)n_nodes, E = (size_t)n_edges;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    return up(256) + up((E / 256 + 2) * 4) + up((N + 1) * 4) + up(E * 4) + up(E * 4) + up((N + 1) * 4);
}

int gnncca_post_threshold(const float* logits, int64_t n_edges, float* probs_out, int64_t* predictions_out,
                          gnncca_stream_t stream) {
    if (n_edges < 0) return GNNCCA_ERR_INVALID_ARG;
    if (n_edges == 0) return GNNCCA_OK;
    if (!logits || !probs_out || !predictions_out) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(post_threshold_kernel, grid1((size_t)n_edges, 256), dim3(256), 0, st, logits, (long long)n_edges, probs_out,
                       reinterpret_cast<long long*>(predictions_out));
    HIP_TRY(hipGetLastError());
    return GNNCCA_OK;
}

int gnncca_post_prune_cluster(const int64_t* edge_index, const int64_t* predictions, int64_t n_nodes, int64_t n_edges,
                              void* workspace, size_t workspace_bytes, int64_t* pruned_out, int32_t* flow_out,
                              int32_t* flow_in, int32_t* labels_out, int32_t* n_clusters_out, gnncca_stream_t stream) {
    return gnncca_post_prune_cluster_frames(edge_index, predictions, n_nodes, n_edges, nullptr, nullptr, 0, workspace,
                                            workspace_bytes, pruned_out, flow_out, flow_in, labels_out, n_clusters_out, stream);
}

int gnncca_post_prune_cluster_frames(const int64_t* edge_index, const int64_t* predictions, int64_t n_nodes, int64_t n_edges,
                                     const int32_t* node_ptr_dev, const int32_t* edge_ptr_dev, int32_t n_frames,
                                     void* workspace, size_t workspace_bytes, int64_t* pruned_out, int32_t* flow_out,
                                     int32_t* flow_in, int32_t* labels_out, int32_t* n_clusters_out,
                                     gnncca_stream_t stream) {
    return gnncca_post_prune_cluster_frames_ex(edge_index, predictions, n_nodes, n_edges, node_ptr_dev, edge_ptr_dev, n_frames, workspace,
                                               workspace_bytes, pruned_out, flow_out, flow_in, labels_out, n_clusters_out, nullptr, nullptr, stream);
}

// `plan` (internal; null from the public entry points): the graph plan the MPN forward of the same batch left in ITS workspace (seg_ptr,
// col32, perm, flags) -- gnncca_frames_forward hands it over, so the pruning needs no plan launches of its own
struct PostPlan {
    const int* seg_ptr;
    const int* col32;
    const int* perm;
    const unsigned* flags;
    // gnncca_frames_forward's two other savings: the counters were zeroed by an earlier kernel of the batch (no memset node here), and the
    // threshold rides in the prune kernel (logits in, probabilities and predictions out)
    bool counters_zeroed;
    const float* logits;
    float* probs_out;
    int64_t* preds_out;
};
static int post_prune_cluster_impl(const int64_t* edge_index, const int64_t* predictions, int64_t n_nodes, int64_t n_edges,
                                   const int32_t* node_ptr_dev, const int32_t* edge_ptr_dev, int32_t n_frames, void* workspace,
                                   size_t workspace_bytes, int64_t* pruned_out, int32_t* flow_out, int32_t* flow_in, int32_t* labels_out,
                                   int32_t* n_clusters_out, int32_t* sizes_scratch, int32_t* triggers_out, const PostPlan* plan,
                                   gnncca_stream_t stream);

int gnncca_post_prune_cluster_frames_ex(const int64_t* edge_index, const int64_t* predictions, int64_t n_nodes, int64_t n_edges,
                                        const int32_t* node_ptr_dev, const int32_t* edge_ptr_dev, int32_t n_frames, void* workspace,
                                        size_t workspace_bytes, int64_t* pruned_out, int32_t* flow_out, int32_t* flow_in,
                                        int32_t* labels_out, int32_t* n_clusters_out, int32_t* sizes_scratch, int32_t* triggers_out,
                                        gnncca_stream_t stream) {
    return post_prune_cluster_impl(edge_index, predictions, n_nodes, n_edges, node_ptr_dev, edge_ptr_dev, n_frames, workspace, workspace_bytes,
                                   pruned_out, flow_out, flow_in, labels_out, n_clusters_out, sizes_scratch, triggers_out, nullptr, stream);
}

static int post_prune_cluster_impl(const int64_t* edge_index, const int64_t* predictions, int64_t n_nodes, int64_t n_edges,
                                   const int32_t* node_ptr_dev, const int32_t* edge_ptr_dev, int32_t n_frames, void* workspace,
                                   size_t workspace_bytes, int64_t* pruned_out, int32_t* flow_out, int32_t* flow_in, int32_t* labels_out,
                                   int32_t* n_clusters_out, int32_t* sizes_scratch, int32_t* triggers_out, const PostPlan* plan,
                                   gnncca_stream_t stream) {
    if (n_nodes < 0 || n_edges < 0 || n_frames < 0) return GNNCCA_ERR_INVALID_ARG;
    if ((sizes_scratch == nullptr) != (triggers_out == nullptr)) return GNNCCA_ERR_INVALID_ARG;
    if ((node_ptr_dev == nullptr) != (edge_ptr_dev == nullptr) || (node_ptr_dev != nullptr && n_frames == 0))
        return GNNCCA_ERR_INVALID_ARG;
    if (n_nodes >= (1ll << 31) - 64 || n_edges >= (1ll << 31) - 64) return GNNCCA_ERR_UNSUPPORTED;
    const size_t n_trig = (size_t)(node_ptr_dev ? n_frames : 1);
    if (n_nodes == 0) {
        if (n_clusters_out) HIP_TRY(hipMemsetAsync(n_clusters_out, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)));
        if (triggers_out) HIP_TRY(hipMemsetAsync(triggers_out, 0, n_trig * sizeof(int32_t), static_cast<hipStream_t>(stream)));
        return GNNCCA_OK;
    }
    if (!workspace || !flow_out || !flow_in || !labels_out || !n_clusters_out) return GNNCCA_ERR_INVALID_ARG;
    if (n_edges > 0 && (!edge_index || !predictions || !pruned_out)) return GNNCCA_ERR_INVALID_ARG;
    if (!plan && workspace_bytes < gnncca_post_workspace_bytes(n_nodes, n_edges)) return GNNCCA_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = (int)n_nodes, E = (int)n_edges;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    char* base = static_cast<char*>(workspace);
    unsigned* flags = reinterpret_cast<unsigned*>(base);
    size_t off = up(256);
    unsigned* blockflags = reinterpret_cast<unsigned*>(base + off);
    off += up(((size_t)E / 256 + 2) * 4);
    int* seg_ptr = reinterpret_cast<int*>(base + off);
    off += up(((size_t)N + 1) * 4);
    int* col32 = reinterpret_cast<int*>(base + off);
    off += up((size_t)E * 4);
    int* perm = reinterpret_cast<int*>(base + off);
    off += up((size_t)E * 4);
    int* cursor = reinterpret_cast<int*>(base + off);
    if (plan) {   // the MPN forward's plan of the same edge_index: nothing to build here
        seg_ptr = const_cast<int*>(plan->seg_ptr), col32 = const_cast<int*>(plan->col32), perm = const_cast<int*>(plan->perm);
        flags = const_cast<unsigned*>(plan->flags);
    }
    const long long* ei = reinterpret_cast<const long long*>(edge_index);
    const long long* pred = reinterpret_cast<const long long*>(predictions);
    long long* pruned = reinterpret_cast<long long*>(pruned_out);
    // zero the counters: one memset when the caller laid flow_out | flow_in | n_clusters out back to back (gnn_cca_amd.postprocess does)
    const bool one_block = flow_in == flow_out + N && n_clusters_out == flow_in + N;
    const bool trig_block = one_block && triggers_out && sizes_scratch == n_clusters_out + 1 && triggers_out == sizes_scratch + N;
    if (plan && plan->counters_zeroed) {
        // nothing: done launches ago
    } else if (trig_block) {
        HIP_TRY(hipMemsetAsync(flow_out, 0, ((size_t)3 * N + 1 + n_trig) * 4, st));
    } else if (one_block) {
        HIP_TRY(hipMemsetAsync(flow_out, 0, ((size_t)2 * N + 1) * 4, st));
    } else {
        HIP_TRY(hipMemsetAsync(flow_out, 0, (size_t)N * 4, st));
        HIP_TRY(hipMemsetAsync(flow_in, 0, (size_t)N * 4, st));
        HIP_TRY(hipMemsetAsync(n_clusters_out, 0, sizeof(int32_t), st));
    }
    if (triggers_out && !trig_block && !(plan && plan->counters_zeroed)) {
        HIP_TRY(hipMemsetAsync(sizes_scratch, 0, (size_t)N * 4, st));
        HIP_TRY(hipMemsetAsync(triggers_out, 0, n_trig * 4, st));
    }
    if (E > 0 && !plan) {
        EncPlanParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.ei = ei;
        ep.seg_ptr = seg_ptr;
        ep.col32 = col32;
        ep.blockflags = blockflags;
        ep.E = E;
        ep.N = N;
        hipLaunchKernelGGL(plan_only_kernel, dim3(plan_num_blocks(E)), dim3(256), 0, st, ep);
        HIP_TRY(hipGetLastError());
    }
    if (!plan) {
        hipLaunchKernelGGL(gen_plan_finish_kernel, dim3(1), dim3(256), 0, st, ei, E, N, seg_ptr, col32, perm, cursor, flags,
                           (const unsigned*)blockflags);
        HIP_TRY(hipGetLastError());
    }
    if (E > 0) {
        if (plan && plan->logits)
            hipLaunchKernelGGL(post_prune_kernel<true>, grid1((size_t)E, 256), dim3(256), 0, st, ei, pred, (long long)E, (const int*)seg_ptr,
                               (const int*)col32, (const int*)perm, (const unsigned*)flags, pruned, flow_out, flow_in, plan->logits,
                               plan->probs_out, reinterpret_cast<long long*>(plan->preds_out));
        else
            hipLaunchKernelGGL(post_prune_kernel<false>, grid1((size_t)E, 256), dim3(256), 0, st, ei, pred, (long long)E, (const int*)seg_ptr,
                               (const int*)col32, (const int*)perm, (const unsigned*)flags, pruned, flow_out, flow_in, (const float*)nullptr,
                               (float*)nullptr, (long long*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(post_cc_kernel, dim3(node_ptr_dev ? (unsigned)n_frames : 1u), dim3(1024), 0, st, ei,
                       (const long long*)pruned, (long long)E, N, node_ptr_dev, edge_ptr_dev, labels_out, n_clusters_out,
                       (const int*)flow_out, (const int*)flow_in, sizes_scratch, triggers_out);
    HIP_TRY(hipGetLastError());
    return GNNCCA_OK;
}


int gnncca_frames_forward(const gnncca_mpn_dims* d, const void* packed_dev, const gnncca_frames_io* io, void* mpn_workspace,
                          size_t mpn_workspace_bytes, void* post_workspace, size_t post_workspace_bytes, uint32_t options,
                          gnncca_stream_t stream) {
    if (!d || !io || !io->staged_dev) return GNNCCA_ERR_INVALID_ARG;
    const int64_t n = io->n_nodes, g = io->n_frames, e = io->n_edges;
    if (n < 1 || g < 1 || e < 0) return GNNCCA_ERR_INVALID_ARG;
    if (n > 4096) return GNNCCA_ERR_UNSUPPORTED;   // (the one-launch normalisation's limit; bigger batches take the separate entry points)
    if (!io->node_embeds || !io->reid_embeds || !io->edge_index || !io->edge_attr || !io->edge_labels || !io->logits || !io->probs ||
        !io->predictions || !io->pruned || !io->counters || !io->labels || (io->normalize && (!io->node_norm || !io->reid_norm)))
        return GNNCCA_ERR_INVALID_ARG;
    if (io->counters_len < 3 * n + 1 + g) return GNNCCA_ERR_INVALID_ARG;   // flow_out | flow_in | n_clusters | sizes | triggers (ABI 2: stated, not assumed)
    // the staging image (gnncca_plan_frames): f64 xw[n], yw[n], max_dist[g]; i64 ids[n]; i32 person, cam, graph_of, graph_ptr, src_order, edge_ptr, edge_ptr_g
    const char* base = static_cast<const char*>(io->staged_dev);
    gnncca_frames fr;
    fr.xw = reinterpret_cast<const double*>(base);
    fr.yw = fr.xw + n;
    fr.max_dist = fr.yw + n;
    const int32_t* i32 = reinterpret_cast<const int32_t*>(base + 8 * (3 * n + g));
    fr.person_id = i32, fr.cam = i32 + n, fr.graph_of = i32 + 2 * n, fr.graph_ptr = i32 + 3 * n;
    fr.src_order = i32 + 3 * n + g + 1, fr.edge_ptr = i32 + 4 * n + g + 1;
    const int32_t* edge_ptr_g = i32 + 5 * n + g + 2;
    const float* x = io->node_embeds;
    const float* reid = io->reid_embeds;
    int st = GNNCCA_OK;
    if (io->normalize) {
        st = gnncca_normalize_columns2(io->reid_embeds, io->reid_dim, io->reid_norm, io->node_embeds, d->node_in, io->node_norm, n, stream);
        if (st != GNNCCA_OK) return st;
        x = io->node_norm, reid = io->reid_norm;
    }
    // (the post stage's counters -- flow_out | flow_in | n_clusters | sizes | triggers -- are zeroed by this launch: no memset node later)
    const bool zero_here = e > 0 && mpn_workspace != nullptr;
    st = build_edges_zeroing(&fr, reid, io->reid_dim, n, e, io->mode, io->edge_index, io->edge_attr, io->edge_labels,
                             zero_here ? io->counters : nullptr, zero_here ? 3 * n + 1 + g : 0, stream);
    if (st != GNNCCA_OK) return st;
    const int n_out = gnncca_num_outputs(d);
    if (n_out < 1) return GNNCCA_ERR_UNSUPPORTED;
    if (e > 0) {
        st = gnncca_mpn_forward_ex(d, packed_dev, x, io->edge_index, io->edge_attr, n, e, mpn_workspace, mpn_workspace_bytes, io->logits, nullptr,
                                   options, stream);
        if (st != GNNCCA_OK) return st;
        if (!zero_here) {   // (no shared plan: the threshold keeps its own launch)
            st = gnncca_post_threshold(io->logits + (size_t)(n_out - 1) * e, e, io->probs, io->predictions, stream);
            if (st != GNNCCA_OK) return st;
        }
    }
    // the pruning searches reverse edges in the CSR plan of edge_index -- the one the forward above left in ITS workspace (seg_ptr / col32 /
    // perm / flag word: same plan_block + plan_finish, same stream): handed over instead of being built a second time (two launches less)
    PostPlan plan;
    const PostPlan* have_plan = nullptr;
    if (e > 0 && mpn_workspace) {
        char* wb = static_cast<char*>(mpn_workspace);
        if (classify(d) == kFamilyMfma32x6) {
            const Workspace ws = carve(d, n, e);
            plan = PostPlan{reinterpret_cast<const int*>(wb + ws.seg_ptr), reinterpret_cast<const int*>(wb + ws.col32),
                            reinterpret_cast<const int*>(wb + ws.perm), reinterpret_cast<const unsigned*>(wb + ws.flags), false, nullptr, nullptr, nullptr};
        } else {
            const GenWorkspace ws = carve_generic(d, n, e);
            plan = PostPlan{reinterpret_cast<const int*>(wb + ws.seg_ptr), reinterpret_cast<const int*>(wb + ws.col32),
                            reinterpret_cast<const int*>(wb + ws.perm), reinterpret_cast<const unsigned*>(wb + ws.flags), false, nullptr, nullptr, nullptr};
        }
        plan.counters_zeroed = zero_here;
        plan.logits = io->logits + (size_t)(n_out - 1) * e, plan.probs_out = io->probs, plan.preds_out = io->predictions;
        have_plan = &plan;
    }
    int32_t* c = io->counters;   // flow_out | flow_in | n_clusters | sizes (scratch) | triggers [G]
    return post_prune_cluster_impl(io->edge_index, io->predictions, n, e, fr.graph_ptr, edge_ptr_g, (int32_t)g, post_workspace, post_workspace_bytes,
                                   io->pruned, c, c + n, io->labels, c + 2 * n, c + 2 * n + 1, c + 3 * n + 1, have_plan, stream);
}

// ---- backward ---------------------------------------------------------------------------
// gnncca_dropout -> DropCfg; GNNCCA_OK with all p == 0 for a null / inactive one
static int drop_cfg(const gnncca_dropout* dropout, DropCfg* out) {
    std::memset(out, 0, sizeof(*out));
    if (!dropout) return GNNCCA_OK;
    const float ps[4] = {dropout->p_enc, dropout->p_edge, dropout->p_node, dropout->p_cls};
    bool any = false;
    for (float q : ps) {
        if (!(q >= 0.f && q < 1.f)) return GNNCCA_ERR_INVALID_ARG;
        any = any || q > 0.f;
    }
    if (!any) return GNNCCA_OK;
    if (!dropout->seed_dev) return GNNCCA_ERR_INVALID_ARG;
    out->p_enc = dropout->p_enc, out->p_edge = dropout->p_edge, out->p_node = dropout->p_node, out->p_cls = dropout->p_cls;
    out->seed = reinterpret_cast<const unsigned long long*>(dropout->seed_dev);
    return GNNCCA_OK;
}

static bool backward_ok(const gnncca_mpn_dims* d) {
    if (classify(d) != kFamilyMfma32x6) return false;
    if (d->num_enc_steps < 1) return false;
    if (d->enc_node.n_layers != 2) return false;
    const gnncca_mlp* all[5] = {&d->enc_node, &d->enc_edge, &d->edge_mlp, &d->node_mlp, &d->cls_edge};
    for (int mi = 0; mi < 5; ++mi)
        for (int l = 0; l < all[mi]->n_layers; ++l)
            if (all[mi]->layers[l].has_bn && !(mi == 4 && l == 0 && d->cls_edge.n_layers == 2)) return false;
    return true;  // BatchNorm is allowed only between the classifier's two layers (the shipped inference config)
}

// index of the first tensor of layer `l` of MLP `mi` in the canonical parameter order
static int param_index(const gnncca_mpn_dims* d, int mi, int l) {
    int idx = 0;
    for (int m = 0; m < 5; ++m) {
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        for (int k = 0; k < mlp.n_layers; ++k) {
            if (m == mi && k == l) return idx;
            idx += 2 + (mlp.layers[k].has_bn ? 4 : 0);
        }
    }
    return idx;
}

// Train-mode classifier with BatchNorm1d between its two layers: batch statistics over the E edges for every
// classified step (models/mpn.py:290-293 with models/mlp.py:15 in train mode); running buffers updated in place.
int gnncca_classifier_train(const gnncca_mpn_dims* d, const float* const* params_dev, int n_params, const float* e_steps,
                            int64_t n_edges, void* scratch /* 2*C1 doubles */, float* bn_stat_out /* [n_out][C1][2] */,
                            float* logits_out, gnncca_stream_t stream) {
    return gnncca_classifier_train_dropout(d, params_dev, n_params, e_steps, n_edges, scratch, bn_stat_out, logits_out, nullptr, stream);
}

int gnncca_classifier_train_dropout(const gnncca_mpn_dims* d, const float* const* params_dev, int n_params, const float* e_steps,
                                    int64_t n_edges, void* scratch /* 2*C1 doubles */, float* bn_stat_out /* [n_out][C1][2] */,
                                    float* logits_out, const gnncca_dropout* dropout, gnncca_stream_t stream) {
    DropCfg drop;
    {
        const int ds = drop_cfg(dropout, &drop);
        if (ds != GNNCCA_OK) return ds;
    }
    if (!dims_valid(d) || !params_dev || n_params != gnncca_param_count(d) || n_edges < 0) return GNNCCA_ERR_INVALID_ARG;
    if (!backward_ok(d) || d->cls_edge.n_layers != 2 || !d->cls_edge.layers[0].has_bn) return GNNCCA_ERR_UNSUPPORTED;
    if (n_edges == 0) return GNNCCA_OK;
    if (!e_steps || !scratch || !bn_stat_out || !logits_out) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int C1 = d->cls_edge.layers[0].out_dim, L = d->num_enc_steps, first_cls = L - d->num_class_steps + 1;
    const int pc = param_index(d, 4, 0);
    const float *W1 = params_dev[pc], *b1 = params_dev[pc + 1], *gamma = params_dev[pc + 2], *beta = params_dev[pc + 3];
    float *rm = const_cast<float*>(params_dev[pc + 4]), *rv = const_cast<float*>(params_dev[pc + 5]);
    const float *W2 = params_dev[pc + 6], *b2 = params_dev[pc + 7];
    const long long E = n_edges;
    double* sums = static_cast<double*>(scratch);
    int li = 0;
    for (int s = 1; s <= L; ++s) {
        if (s < first_cls) continue;
        const float* e = e_steps + (size_t)(s - 1) * E * kEF;
        float* stat = bn_stat_out + (size_t)li * C1 * 2;
        HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C1, st));
        hipLaunchKernelGGL(cls_bn_stats_kernel, grid1((size_t)E, 256), dim3(256), 0, st, e, E, W1, b1, C1, sums);
        hipLaunchKernelGGL(cls_bn_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)sums, E, C1, stat, rm, rv);
        hipLaunchKernelGGL(cls_bn_apply_kernel, grid1((size_t)E, 256), dim3(256), 0, st, e, E, W1, b1, gamma, beta, (const float*)stat,
                           W2, b2, C1, logits_out + (size_t)li * E, drop, li);
        HIP_TRY(hipGetLastError());
        ++li;
    }
    return GNNCCA_OK;
}

int gnncca_backward_supported(const gnncca_mpn_dims* d) {
    if (!dims_valid(d)) return GNNCCA_ERR_INVALID_ARG;
    return backward_ok(d) ? GNNCCA_OK : GNNCCA_ERR_UNSUPPORTED;
}

size_t gnncca_backward_workspace_bytes(const gnncca_mpn_dims* d, int64_t n_nodes, int64_t n_edges) {
    if (!dims_valid(d) || !backward_ok(d) || n_nodes < 0 || n_edges < 0) return 0;
    const size_t N = (size_t)n_nodes, E = (size_t)n_edges, F1 = (size_t)d->enc_node.layers[0].out_dim;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t L = (size_t)std::max(d->num_enc_steps, 1);
    return up(N * kH * 4) + up(E * kEF * 4) + up(N * 4) + up(N * kH * 4) + up(2 * N * kH * 4) + up(L * N * 44 * 4) + 2 * up(N * kH * 4) + 2 * up(E * kEF * 4) + 2 * up(N * F1 * 4) +
           up(32 * N * F1 * 4) + up(sizeof(double) * 128) + up(sizeof(float) * 128);
}

size_t gnncca_pack_program_bytes(void) { return sizeof(PackProgram); }

int gnncca_pack_program(const gnncca_mpn_dims* d, void* program_host, size_t program_bytes) {
    if (!dims_valid(d) || !program_host) return GNNCCA_ERR_INVALID_ARG;
    if (classify(d) != kFamilyMfma32x6) return GNNCCA_ERR_UNSUPPORTED;
    if (program_bytes < sizeof(PackProgram)) return GNNCCA_ERR_INVALID_ARG;
    return pack_program(d, static_cast<PackProgram*>(program_host)) ? GNNCCA_OK : GNNCCA_ERR_UNSUPPORTED;
}

int gnncca_pack_weights_device(const gnncca_mpn_dims* d, const float* const* params_dev, int n_params, const void* program_dev,
                               void* packed_dev, size_t packed_bytes, gnncca_stream_t stream) {
    if (!dims_valid(d) || !params_dev || !program_dev || !packed_dev) return GNNCCA_ERR_INVALID_ARG;
    if (classify(d) != kFamilyMfma32x6) return GNNCCA_ERR_UNSUPPORTED;
    if (n_params != gnncca_param_count(d) || n_params > kMaxPackParams) return GNNCCA_ERR_INVALID_ARG;
    if (packed_bytes < gnncca_packed_weights_bytes(d)) return GNNCCA_ERR_INVALID_ARG;
    PackProgram host;  // segment count only: the program itself is read on the device
    if (!pack_program(d, &host)) return GNNCCA_ERR_UNSUPPORTED;
    PackPtrs ptrs;
    std::memset(&ptrs, 0, sizeof(ptrs));
    for (int i = 0; i < n_params; ++i) {
        if (!params_dev[i]) return GNNCCA_ERR_INVALID_ARG;
        ptrs.p[i] = params_dev[i];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pack_device_kernel, dim3(64, (unsigned)host.n_segs + 1), dim3(256), 0, st,
                       static_cast<const PackProgram*>(program_dev), ptrs, static_cast<float*>(packed_dev));
    HIP_TRY(hipGetLastError());
    return GNNCCA_OK;
}

int gnncca_mpn_backward(const gnncca_mpn_dims* d, const float* const* params_dev, int n_params, const float* x,
                        const int64_t* edge_index, const float* edge_attr, int64_t n_nodes, int64_t n_edges,
                        const gnncca_trace* saved, const float* cls_bn_stat, const float* grad_logits,
                        float* const* grads_dev, void* workspace, size_t workspace_bytes, gnncca_stream_t stream) {
    return gnncca_mpn_backward_ex(d, params_dev, n_params, x, edge_index, edge_attr, n_nodes, n_edges, saved, cls_bn_stat,
                                  grad_logits, grads_dev, workspace, workspace_bytes, 0u, stream);
}

int gnncca_mpn_backward_ex(const gnncca_mpn_dims* d, const float* const* params_dev, int n_params, const float* x,
                           const int64_t* edge_index, const float* edge_attr, int64_t n_nodes, int64_t n_edges,
                           const gnncca_trace* saved, const float* cls_bn_stat, const float* grad_logits,
                           float* const* grads_dev, void* workspace, size_t workspace_bytes, uint32_t options,
                           gnncca_stream_t stream) {
    return gnncca_mpn_backward_train(d, params_dev, n_params, x, edge_index, edge_attr, n_nodes, n_edges, saved, cls_bn_stat,
                                     grad_logits, grads_dev, workspace, workspace_bytes, options, nullptr, stream);
}

int gnncca_mpn_backward_train(const gnncca_mpn_dims* d, const float* const* params_dev, int n_params, const float* x,
                              const int64_t* edge_index, const float* edge_attr, int64_t n_nodes, int64_t n_edges,
                              const gnncca_trace* saved, const float* cls_bn_stat, const float* grad_logits,
                              float* const* grads_dev, void* workspace, size_t workspace_bytes, uint32_t options,
                              const gnncca_dropout* dropout, gnncca_stream_t stream) {
    DropCfg drop;
    {
        const int ds = drop_cfg(dropout, &drop);
        if (ds != GNNCCA_OK) return ds;
    }
    if (!dims_valid(d) || n_nodes < 0 || n_edges < 0) return GNNCCA_ERR_INVALID_ARG;
    if (!backward_ok(d)) return GNNCCA_ERR_UNSUPPORTED;
    if (n_params != gnncca_param_count(d) || !params_dev || !grads_dev) return GNNCCA_ERR_INVALID_ARG;
    for (int i = 0; i < n_params; ++i)
        if (!params_dev[i] || !grads_dev[i]) return GNNCCA_ERR_INVALID_ARG;
    if (n_nodes >= (1ll << 31) - 64 || n_edges >= (1ll << 31) - 64) return GNNCCA_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = (int)n_nodes, E = (int)n_edges;
    const int D = d->node_in, F1 = d->enc_node.layers[0].out_dim, A = d->edge_in;
    const int L = d->num_enc_steps, first_cls = L - d->num_class_steps + 1;
    const int c1 = d->cls_edge.n_layers == 2 ? d->cls_edge.layers[0].out_dim : 0;
    // zero every gradient
    if (!(options & GNNCCA_BWD_GRADS_ZEROED)) {
        const gnncca_mlp* all[5] = {&d->enc_node, &d->enc_edge, &d->edge_mlp, &d->node_mlp, &d->cls_edge};
        int pi = 0;
        for (const gnncca_mlp* m : all)
            for (int l = 0; l < m->n_layers; ++l) {
                HIP_TRY(hipMemsetAsync(grads_dev[pi++], 0, (size_t)m->layers[l].in_dim * m->layers[l].out_dim * 4, st));
                HIP_TRY(hipMemsetAsync(grads_dev[pi++], 0, (size_t)m->layers[l].out_dim * 4, st));
                if (m->layers[l].has_bn)  // gamma, beta, and the two buffers (no gradient: left at zero)
                    for (int k = 0; k < 4; ++k) HIP_TRY(hipMemsetAsync(grads_dev[pi++], 0, (size_t)m->layers[l].out_dim * 4, st));
            }
    }
    if (N == 0 || E == 0) return GNNCCA_OK;
    if (!x || !edge_index || !edge_attr || !saved || !saved->h_enc || !saved->e_enc || !saved->h_steps || !saved->e_steps ||
        !grad_logits || !workspace)
        return GNNCCA_ERR_INVALID_ARG;
    if (workspace_bytes < gnncca_backward_workspace_bytes(d, n_nodes, n_edges)) return GNNCCA_ERR_WORKSPACE;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    char* base = static_cast<char*>(workspace);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = base + off; off += up(bytes); return q; };
    float* gh0_acc = reinterpret_cast<float*>(take((size_t)N * kH * 4));   // reattach_initial_nodes: d loss / d h0 via the copies
    float* ge0_acc = reinterpret_cast<float*>(take((size_t)E * kEF * 4));  // reattach_initial_edges: d loss / d e0 via the copies
    int* deg = reinterpret_cast<int*>(take((size_t)N * 4));
    float* Q = reinterpret_cast<float*>(take((size_t)N * kH * 4));
    int* hmax = reinterpret_cast<int*>(take((size_t)2 * N * kH * 4));  // 'max' aggregation only: maxima, then tie counts
    int* hcnt = hmax + (size_t)N * kH;
    float* dP_all = reinterpret_cast<float*>(take((size_t)std::max(L, 1) * N * 44 * 4));  // one table per step, cleared once
    float* Hb[2] = {reinterpret_cast<float*>(take((size_t)N * kH * 4)), reinterpret_cast<float*>(take((size_t)N * kH * 4))};
    float* Gb[2] = {reinterpret_cast<float*>(take((size_t)E * kEF * 4)), reinterpret_cast<float*>(take((size_t)E * kEF * 4))};
    float* a1 = reinterpret_cast<float*>(take((size_t)N * F1 * 4));
    float* gz1 = reinterpret_cast<float*>(take((size_t)N * F1 * 4));
    float* part = reinterpret_cast<float*>(take((size_t)32 * N * F1 * 4));  // split-K partials of the a1 recompute
    const float *W1 = params_dev[0], *b1 = params_dev[1], *W2 = params_dev[2];
    const float *We = params_dev[6], *Wn = params_dev[8], *bn = params_dev[9];
    const bool cls_bn = d->cls_edge.layers[0].has_bn != 0;
    if (cls_bn && !cls_bn_stat) return GNNCCA_ERR_INVALID_ARG;
    const int pw2 = cls_bn ? 16 : 12;  // second classifier layer follows the four BatchNorm tensors
    const float *Wc1 = params_dev[10], *bc1 = params_dev[11], *Wc2 = c1 ? params_dev[pw2] : nullptr;
    float *gW1 = grads_dev[0], *gb1 = grads_dev[1], *gW2 = grads_dev[2], *gb2 = grads_dev[3];
    float *gWe0 = grads_dev[4], *gbe0 = grads_dev[5], *gWe = grads_dev[6], *gbe = grads_dev[7];
    float *gWn = grads_dev[8], *gbn = grads_dev[9], *gWc1 = grads_dev[10], *gbc1 = grads_dev[11];
    float *gWc2 = c1 ? grads_dev[pw2] : nullptr, *gbc2 = c1 ? grads_dev[pw2 + 1] : nullptr;
    double* bn_sums = reinterpret_cast<double*>(take(sizeof(double) * 2 * 64));
    float* bn_red = reinterpret_cast<float*>(take(sizeof(float) * 2 * 64));
    const long long* ei = reinterpret_cast<const long long*>(edge_index);
    if (d->agg == GNNCCA_AGG_MEAN) {
        HIP_TRY(hipMemsetAsync(deg, 0, (size_t)N * 4, st));
        hipLaunchKernelGGL(bwd_degree_kernel, grid1((size_t)E, 256), dim3(256), 0, st, ei, (long long)E, N, deg);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync(dP_all, 0, (size_t)std::max(L, 1) * N * 44 * 4, st));
    const bool re_n = d->reattach_nodes != 0, re_e = d->reattach_edges != 0;
    const int HI = re_n ? 2 * kH : kH, WeLd = 2 * HI + (re_e ? 2 * kEF : kEF), WnLd = HI + kEF;
    if (re_n) HIP_TRY(hipMemsetAsync(gh0_acc, 0, (size_t)N * kH * 4, st));
    if (re_e) HIP_TRY(hipMemsetAsync(ge0_acc, 0, (size_t)E * kEF * 4, st));
    const float* g_h = nullptr;   // d loss / d h_s of the step being processed (null for s = L: its node update is dead)
    const float* ge_in = nullptr; // d loss / d e_s arriving from step s+1
    int out_idx = gnncca_num_outputs(d) - 1;
    for (int s = L; s >= 1; --s) {
        const float* h_prev = s == 1 ? saved->h_enc : saved->h_steps + (size_t)(s - 2) * N * kH;
        const float* e_cur = saved->e_steps + (size_t)(s - 1) * E * kEF;
        const float* e_prev = s == 1 ? saved->e_enc : saved->e_steps + (size_t)(s - 2) * E * kEF;
        if (g_h) {
            hipLaunchKernelGGL(bwd_q_kernel, grid1((size_t)N * kH, 256), dim3(256), 0, st, saved->h_enc, h_prev, Wn, bn, Q, N, HI);
            HIP_TRY(hipGetLastError());
        }
        float* dP = dP_all + (size_t)(s - 1) * N * 44;
        BwdEdgeParams bp;
        std::memset(&bp, 0, sizeof(bp));
        bp.ei = ei;
        bp.e_cur = e_cur;
        bp.e_prev = e_prev;
        bp.e0 = re_e ? saved->e_enc : nullptr;
        bp.ge0_acc = re_e ? ge0_acc : nullptr;
        bp.HI = HI;
        bp.Q = Q;
        bp.g_h = g_h;
        bp.deg = d->agg == GNNCCA_AGG_MEAN ? deg : nullptr;
        if (g_h && d->agg == GNNCCA_AGG_MAX) {  // which edge attained each node's maximum
            HIP_TRY(hipMemsetAsync(hmax, 0, (size_t)2 * N * kH * 4, st));
            hipLaunchKernelGGL(bwd_max_kernel<false>, grid1((size_t)E, 256), dim3(256), 0, st, ei, e_cur, (const float*)Q, Wn,
                               (long long)E, N, HI, hmax, hcnt, drop, s);
            hipLaunchKernelGGL(bwd_max_kernel<true>, grid1((size_t)E, 256), dim3(256), 0, st, ei, e_cur, (const float*)Q, Wn,
                               (long long)E, N, HI, hmax, hcnt, drop, s);
            HIP_TRY(hipGetLastError());
            bp.hmax = hmax;
            bp.hcnt = hcnt;
        }
        bp.g_logit = s >= first_cls ? grad_logits + (size_t)out_idx * E : nullptr;
        if (bp.g_logit && cls_bn) {  // reductions the BatchNorm backward needs before any per-edge gradient
            const float* stat = cls_bn_stat + (size_t)out_idx * c1 * 2;
            HIP_TRY(hipMemsetAsync(bn_sums, 0, sizeof(double) * 2 * c1, st));
            hipLaunchKernelGGL(bwd_cls_bn_reduce_kernel, grid1((size_t)E, 256), dim3(256), 0, st, e_cur, bp.g_logit, (long long)E, Wc1,
                               bc1, params_dev[12], params_dev[13], stat, Wc2, c1, bn_sums, gWc2, gbc2, drop, out_idx);
            hipLaunchKernelGGL(bwd_cls_bn_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)bn_sums, (long long)E, c1, bn_red,
                               grads_dev[12], grads_dev[13]);
            HIP_TRY(hipGetLastError());
            bp.bn_gamma = params_dev[12];
            bp.bn_beta = params_dev[13];
            bp.bn_stat = stat;
            bp.bn_red = bn_red;
        }
        if (bp.g_logit) --out_idx;
        bp.drop = drop;
        bp.step_no = s;
        bp.cls_no = bp.g_logit ? out_idx + 1 : 0;   // out_idx was already stepped down past this classified step
        bp.ge_in = ge_in;
        bp.ge_out = Gb[s & 1];
        bp.dP = dP;
        bp.We = We;
        bp.Wn = Wn;
        bp.Wc1 = Wc1;
        bp.bc1 = bc1;
        bp.Wc2 = Wc2;
        bp.gWe = gWe;
        bp.gbe = gbe;
        bp.gWn = gWn;
        bp.gbn = gbn;
        bp.gWc1 = gWc1;
        bp.gbc1 = gbc1;
        bp.gWc2 = gWc2;
        bp.gbc2 = gbc2;
        bp.E = E;
        bp.N = N;
        bp.cls_hidden = c1;
        {   // persistent grid: enough workgroups to fill the chip, few enough that the final flush of the LDS-resident
            // parameter-gradient sums stays a few thousand atomics
            const unsigned chunks = (unsigned)(((size_t)E + 255) / 256);
            hipLaunchKernelGGL(bwd_edge_kernel, dim3(std::min(chunks, 512u)), dim3(256), 0, st, bp);
        }
        HIP_TRY(hipGetLastError());
        float* g_h_prev = Hb[s & 1];
        hipLaunchKernelGGL(bwd_node_kernel, grid1((size_t)N * HI, 256), dim3(256), 0, st, (const float*)dP, We, Wn, g_h_prev, gh0_acc, N,
                           HI, WeLd);
        HIP_TRY(hipGetLastError());
        // d W_src, d W_dst (columns 0..31, 32..63 of the edge-MLP weight), d W_nx (columns 0..31 of the node-MLP weight)
        // d W_src, d W_dst (columns [0, HI) and [HI, 2 HI) of the edge-MLP weight), d W_nx (columns [0, HI) of the node-MLP
        // weight): one product dP^T hin [44][HI], rows routed to the three weight blocks; hin = cat(h0, h_prev) with
        // reattach_initial_nodes, i.e. two 32-column products
        for (int part = 0; part < (re_n ? 2 : 1); ++part) {
            const float* hsrc = (re_n && part == 0) ? saved->h_enc : h_prev;
            const int coff = part * kH;
            OuterOut oo;
            oo.ptr[0] = gWe + coff, oo.ptr[1] = gWe + HI + coff, oo.ptr[2] = g_h ? gWn + coff : nullptr;
            oo.ld[0] = oo.ld[1] = WeLd, oo.ld[2] = WnLd;
            oo.row_begin[0] = 0, oo.row_begin[1] = 6, oo.row_begin[2] = 12, oo.row_begin[3] = 44;
            HIP_TRY(launch_outer_multi(dP, 44, hsrc, kH, oo, nullptr, N, g_h ? 44 : 12, kH, st));
        }
        g_h = g_h_prev;
        ge_in = Gb[s & 1];
    }
    // gradients that reached the encoder outputs through the reattached copies
    if (re_n) hipLaunchKernelGGL(bwd_add_kernel, grid1((size_t)N * kH, 256), dim3(256), 0, st, const_cast<float*>(g_h),
                                 (const float*)gh0_acc, (long long)N * kH);
    if (re_e) hipLaunchKernelGGL(bwd_add_kernel, grid1((size_t)E * kEF, 256), dim3(256), 0, st, const_cast<float*>(ge_in),
                                 (const float*)ge0_acc, (long long)E * kEF);
    // ---- encoders ---------------------------------------------------------------------------------------------------
    hipLaunchKernelGGL(bwd_edge_enc_kernel, dim3(std::min((unsigned)(((size_t)E + 255) / 256), 512u)), dim3(256), 0, st, ge_in,
                       saved->e_enc, edge_attr, A, (long long)E,
                       gWe0, gbe0, 1.f / (1.f - drop.p_enc));
    HIP_TRY(hipGetLastError());
    {   // a1 = ReLU(x W1^T + b1) is recomputed instead of stored: the forward's split-K MFMA GEMM + its reduce kernel
        int ks = 1;
        while (ks < 32 && (size_t)((N + 31) / 32) * ks < 512 && D / (ks * 2) >= 64) ks *= 2;
        int kslice = (D + ks - 1) / ks;
        kslice = (kslice + 63) / 64 * 64;
        EncPlanParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.in = x;
        ep.W = W1;
        ep.part = part;
        ep.M = N;
        ep.K = D;
        ep.O = F1;
        ep.kslice = kslice;
        ep.vec_ok = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        ep.nrt = (N + 31) / 32;
        ep.nks = ks;
        ep.gemm_blocks = ep.nrt * ks * ((F1 + 127) / 128);
        hipLaunchKernelGGL(enc_gemm_plan_kernel, dim3(ep.gemm_blocks), dim3(256), 0, st, ep);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(reduce_bias_act_kernel, grid1((size_t)N * F1, 256), dim3(256), 0, st, (const float*)part, b1, a1, N, F1, ks,
                           1);
        HIP_TRY(hipGetLastError());
        if (drop.p_enc > 0.f) {   // the forward's layer-2 input was a1 AFTER Dropout: re-derive the same mask
            hipLaunchKernelGGL(apply_dropout_kernel, grid1((size_t)N * F1, 256), dim3(256), 0, st, a1, (long long)N * F1, drop,
                               (unsigned)kDropEncNode1, drop.p_enc);
            HIP_TRY(hipGetLastError());
        }
    }
    float* gz2 = const_cast<float*>(g_h);  // [N][32] d loss / d h_enc, masked in place
    const float enc_scale = 1.f / (1.f - drop.p_enc);
    hipLaunchKernelGGL(bwd_relu_mask_kernel, grid1((size_t)N * kH, 256), dim3(256), 0, st, gz2, saved->h_enc, (long long)N * kH, enc_scale);
    HIP_TRY(launch_outer(gz2, kH, a1, F1, gW2, F1, gb2, N, kH, F1, st));
    hipLaunchKernelGGL(bwd_matmul_mask_kernel, grid1((size_t)N * F1, 256), dim3(256), 0, st, (const float*)gz2, W2, (const float*)a1,
                       gz1, N, kH, F1, enc_scale);
    HIP_TRY(launch_outer(gz1, F1, x, D, gW1, D, gb1, N, F1, D, st));
    HIP_TRY(hipGetLastError());
    return GNNCCA_OK;
}

}  // extern "C"
