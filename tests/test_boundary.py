"""Boundary tests that need no GPU: the C-ABI library loads and exports every symbol the header declares, the
module mirrors the reference's surface (constructor, state_dict keys, error behaviour), and the packed weight blob
(csrc/pack.cpp: BN folding, weight splitting, transposes, MFMA operand layout) reproduces the oracle when the
split-form algebra of DESIGN.md section 2 is evaluated from the blob alone (numpy, no compute calls into the library).
"""
import copy
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, ROOT, golden_cases
from oracle.mpn_oracle import load_case


def _model(name):
    from gnn_cca_amd import MOTMPNet
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.eval(), params, arch, sd, a


def test_library_exports_every_declared_symbol():
    from gnn_cca_amd import _native as nat
    header = open(os.path.join(ROOT, "include", "gnncca_mpn.h")).read()
    declared = sorted(set(re.findall(r"GNNCCA_API[^;(]*?\b(gnncca_\w+)\s*\(", header)))
    assert declared, "no declarations found in the header"
    lib = nat.lib()
    for name in declared:
        assert hasattr(lib, name), f"libgnncca_mpn.so does not export {name}"
    assert sorted(nat.exported_symbols()) == declared, "ctypes binding and header disagree"
    assert lib.gnncca_abi_version() == nat.ABI_VERSION
    assert lib.gnncca_status_string(2).decode().startswith("GRAPH_NET_PARAMS not supported")


def test_frames_forward_refuses_a_short_counters_buffer():
    """ABI 2: gnncca_frames_forward writes [3 N + 1 + G] int32 through io.counters; the caller states the length it allocated and a
    shorter one is refused BEFORE anything is launched (no device needed: the argument checks come first; the pointers are never followed)."""
    import ctypes as C
    from gnn_cca_amd import _native as nat
    m, *_ = _model("terrace32")
    d = m.native_dims()
    assert d.abi_version == 2 == nat.ABI_VERSION
    n, g, e = 10, 2, 40
    io = nat.FramesIO()
    fake = 0x10000
    for name, _t in nat.FramesIO._fields_:
        if _t is C.c_void_p:
            setattr(io, name, fake)
    io.n_nodes, io.n_frames, io.n_edges, io.reid_dim, io.mode, io.normalize = n, g, e, 8, 0, 0
    lib = nat.lib()
    for short in (0, 2 * n + 1, 3 * n + g):
        io.counters_len = short
        assert lib.gnncca_frames_forward(C.byref(d), fake, C.byref(io), None, 0, None, 0, 0, None) == nat.ERR_INVALID_ARG
    header = open(os.path.join(ROOT, "include", "gnncca_mpn.h")).read()
    assert "int64_t counters_len;" in header and "#define GNNCCA_ABI_VERSION 2" in header


def test_forward_options_match_the_header():
    """The option bits of gnncca_mpn_forward_ex: the Python constants equal the header's #defines, and the module attributes
    (edge_state_dtype, encoder_products, encoder_unsplit) map onto them; illegal values raise before anything is launched."""
    from gnn_cca_amd import _native as nat
    header = open(os.path.join(ROOT, "include", "gnncca_mpn.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"#define (GNNCCA_OPT_\w+) (\d+)u", header)}
    assert defs == {"GNNCCA_OPT_EDGE_STATE_BF16": nat.OPT_EDGE_STATE_BF16, "GNNCCA_OPT_ENC_SPLIT3": nat.OPT_ENC_SPLIT3,
                    "GNNCCA_OPT_ENC_UNSPLIT": nat.OPT_ENC_UNSPLIT, "GNNCCA_OPT_COLUMN_RANGES": nat.OPT_COLUMN_RANGES}
    assert len(set(defs.values())) == 4 and all(v & (v - 1) == 0 for v in defs.values())   # distinct single bits
    m, *_ = _model("dense64")
    assert m._options() == 0
    m.edge_state_dtype, m.encoder_products, m.encoder_unsplit = "bf16", 3, True
    assert m._options() == nat.OPT_EDGE_STATE_BF16 | nat.OPT_ENC_SPLIT3 | nat.OPT_ENC_UNSPLIT
    m.column_ranges = True
    assert m._options() == nat.OPT_EDGE_STATE_BF16 | nat.OPT_ENC_SPLIT3 | nat.OPT_ENC_UNSPLIT | nat.OPT_COLUMN_RANGES
    m.encoder_products = 4
    with pytest.raises(ValueError):
        m._options()
    m.encoder_products, m.edge_state_dtype = 6, "fp16"
    with pytest.raises(ValueError):
        m._options()


@pytest.mark.parametrize("case", ["default_bn", "bn_off_mean", "reattach", "generic"])
def test_same_seed_gives_the_reference_parameters_and_generator_state(case):
    """Construction under a seed reproduces the reference bit for bit: every parameter / buffer of the state_dict (sha256 over
    names and bytes, in order) AND the state of torch's global generator afterwards -- the reference's unused `node_mlp_old`
    (models/mpn.py:241-242) draws from it after the real parameters exist, so a seeded training script that shuffles or samples
    after building the model sees the same stream.  Golden: tests/golden/make_golden_rng.py (the reference's own constructor)."""
    import hashlib
    import json
    from gnn_cca_amd import MOTMPNet
    g = np.load(os.path.join(GOLDEN_DIR, "rng_after_init.npz"))
    params, arch, seed = json.loads(str(g[f"{case}::params_json"])), str(g[f"{case}::arch"]), int(g[f"{case}::seed"])
    torch.manual_seed(seed)
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    after = torch.rand(8).numpy()
    h = hashlib.sha256()
    for k, v in m.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), g[f"{case}::state_sha256"])
    assert np.array_equal(after, g[f"{case}::after"])


@pytest.mark.parametrize("name", golden_cases())
def test_state_dict_keys_and_shapes_match_reference(name):
    m, params, arch, sd, _ = _model(name)
    mine = m.state_dict()
    assert list(mine.keys()) == list(sd.keys())  # same keys, same order as the reference module
    for k, v in sd.items():
        assert tuple(mine[k].shape) == tuple(np.asarray(v).shape), k


def test_constructor_contract():
    from gnn_cca_amd import MOTMPNet
    params, arch, _, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    p = copy.deepcopy(params)
    MOTMPNet(p, None, arch)
    assert p["encoder_feats_dict"]["edges"]["node_in_dim"] == 2048  # in-place merge, like mpn.py:167-169
    bad = copy.deepcopy(params)
    bad["node_agg_fn"] = "median"
    with pytest.raises(AssertionError):
        MOTMPNet(bad, None, arch)
    bad = copy.deepcopy(params)
    bad["edge_model_feats_dict"]["fc_dims"] = 6
    with pytest.raises(AssertionError):
        MOTMPNet(bad, None, arch)
    bad = copy.deepcopy(params)
    del bad["num_class_steps"]
    with pytest.raises(KeyError):
        MOTMPNet(bad, None, arch)
    with pytest.raises(KeyError):
        MOTMPNet(copy.deepcopy(params), None, "no_such_arch")


def test_no_cpu_fallback_and_unsupported_train_mode_refused():
    m, _, _, _, a = _model("n8_sum")

    class D:
        x, edge_index, edge_attr = (torch.from_numpy(a[k]) for k in ("x", "edge_index", "edge_attr"))

    with pytest.raises(RuntimeError):          # CPU tensors: no fallback
        m(D())
    m.train()
    with pytest.raises(RuntimeError):          # train mode of a shipped shape: supported, but GPU only
        m(D())
    m2, _, _, _, _ = _model("generic_dims")    # the generic family trains on the layer-by-layer engine: GPU only as well
    m2.train()
    with pytest.raises(RuntimeError):
        m2(D())
    m2.train_engine = "fused"                  # ... and cannot be forced onto the fused kernels of the shipped shapes
    m2.train()
    with pytest.raises(NotImplementedError):
        m2(D())


def test_supported_family_and_counts():
    from gnn_cca_amd import _native as nat
    lib = nat.lib()
    for name in golden_cases():
        m, params, _, sd, _ = _model(name)
        d = m.native_dims()
        assert lib.gnncca_param_count(C.byref(d)) == len(m.native_param_tensors())
        n_out = 1 if params["num_enc_steps"] == 0 else min(params["num_enc_steps"], params["num_class_steps"])
        assert lib.gnncca_num_outputs(C.byref(d)) == n_out
        assert lib.gnncca_supported(C.byref(d)) == nat.OK, name  # MFMA family or the generic family
        assert lib.gnncca_workspace_bytes(C.byref(d), 256, 65280) > 0
        nbytes = lib.gnncca_packed_weights_bytes(C.byref(d))
        assert nbytes > 0 and m.pack_weights_host().numel() == nbytes
    bad = nat.MpnDims()
    assert lib.gnncca_supported(C.byref(bad)) == nat.ERR_INVALID_ARG


def test_too_deep_mlp_is_refused_loudly():
    from gnn_cca_amd import MOTMPNet
    params, arch, _, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    p = copy.deepcopy(params)
    p["encoder_feats_dict"]["nodes"][arch]["node_fc_dims"] = [64] * 9  # 10 layers > GNNCCA_MAX_LAYERS
    m = MOTMPNet(p, None, arch)
    with pytest.raises(NotImplementedError):
        m.native_dims()


# ---- the packed blob, evaluated with the split-form algebra in numpy ------------------------------------------
class BlobHeader(C.Structure):  # mirror of csrc/internal.h: BlobHeader
    _fields_ = [("magic", C.c_uint32), ("abi_version", C.c_uint32), ("family", C.c_uint32),
                ("total_floats", C.c_uint32), ("enc_node_layers", C.c_int32), ("enc_node_w", C.c_int32 * 8),
                ("enc_node_b", C.c_int32 * 8), ("enc_last_wT", C.c_int32), ("enc_edge_w", C.c_int32),
                ("enc_edge_b", C.c_int32), ("wee", C.c_int32), ("wne_b", C.c_int32), ("proj_wT", C.c_int32),
                ("proj_b", C.c_int32), ("cls_layers", C.c_int32), ("cls_hidden", C.c_int32), ("cls_w1", C.c_int32),
                ("cls_b1", C.c_int32), ("cls_w2", C.c_int32), ("cls_b2", C.c_int32), ("fast_consts", C.c_int32),
                ("enc_w3", C.c_int32), ("wne_bf16", C.c_int32), ("enc_w2h", C.c_int32), ("enc_w2h_bad", C.c_int32),
                ("pad", C.c_int32 * 3)]


def blob_forward(blob_u8, params, arch, x, edge_index, edge_attr):
    raw = blob_u8.numpy().tobytes()
    h = BlobHeader.from_buffer_copy(raw[:C.sizeof(BlobHeader)])
    f = np.frombuffer(raw, dtype=np.float32)
    assert h.magic == 0x4D504E36 and h.total_floats * 4 == len(raw)   # "MPN6"
    enc = params["encoder_feats_dict"]["nodes"][arch]
    dims = [enc["node_in_dim"]] + list(enc["node_fc_dims"]) + [enc["node_out_dim"]]
    if h.enc_w2h:   # the fp16 piece image of the first encoder weight (csrc/enc_f16.cuh): w0 = fp16(w), w1 = fp16((w - w0) * 2048), bit for bit
        # numpy's round-to-nearest-even conversions, fragment-major inside every 32-deep chunk (csrc/pack.cpp: w2h_index)
        K, O = dims[0], dims[1]
        assert O == 128 and K % 32 == 0
        W = f[h.enc_node_w[0]:h.enc_node_w[0] + K * O].reshape(O, K)
        # [chunk][piece][column tile][k-step][lane half h][column in tile][8]: lane = column % 32 + 32 h holds k = 16 s + 8 h + (0 ... 7)
        img = np.frombuffer(raw, dtype=np.float16)[2 * h.enc_w2h:2 * h.enc_w2h + 2 * K * O].reshape(K // 32, 2, 4, 2, 2, 32, 8)
        nat = img.transpose(1, 2, 5, 0, 3, 4, 6)                          # [piece][tile][column][chunk][s][h][8] -> natural (column, k)
        w0 = nat[0].reshape(O, K)
        w1 = nat[1].reshape(O, K)
        want0 = W.astype(np.float16)
        want1 = ((W - want0.astype(np.float32)) * np.float32(2048)).astype(np.float16)
        assert np.array_equal(w0.view(np.uint16), want0.view(np.uint16)) and np.array_equal(w1.view(np.uint16), want1.view(np.uint16))
        bad = np.frombuffer(raw, dtype=np.uint32)[h.enc_w2h_bad:h.enc_w2h_bad + 64]
        assert bad.any() == bool((np.abs(W) >= 65520).any())
    nf = 2 if params["reattach_initial_nodes"] else 1
    ef = 2 if params["reattach_initial_edges"] else 1
    edge_in = params["encoder_feats_dict"]["edges"]["edge_in_dim"]
    relu = lambda v: np.maximum(v, 0)
    hcur = x
    for i in range(h.enc_node_layers):
        W = f[h.enc_node_w[i]:h.enc_node_w[i] + dims[i] * dims[i + 1]].reshape(dims[i + 1], dims[i])
        if i == h.enc_node_layers - 1:  # the transposed copy must agree with the row-major one
            WT = f[h.enc_last_wT:h.enc_last_wT + dims[i] * 32].reshape(dims[i], 32)
            assert np.array_equal(WT.T, W)
        hcur = relu(hcur @ W.T + f[h.enc_node_b[i]:h.enc_node_b[i] + dims[i + 1]])
    h0 = hcur
    e = relu(edge_attr @ f[h.enc_edge_w:h.enc_edge_w + 6 * edge_in].reshape(6, edge_in).T + f[h.enc_edge_b:h.enc_edge_b + 6])
    e0 = e
    projT = f[h.proj_wT:h.proj_wT + nf * 32 * 48].reshape(nf * 32, 48)
    projb = f[h.proj_b:h.proj_b + 48]
    wee = f[h.wee:h.wee + 6 * ef * 6].reshape(6, ef * 6)
    wneb = f[h.wne_b:h.wne_b + 192].reshape(3, 64)
    wne = np.zeros((32, 6), np.float32)
    for s in range(3):
        for l in range(64):
            wne[l & 31, 2 * s + (l >> 5)] = wneb[s, l]
    row, col = edge_index
    n = x.shape[0]
    L, first = params["num_enc_steps"], params["num_enc_steps"] - params["num_class_steps"] + 1

    def classify(ee):
        if h.cls_layers == 2:
            c1 = h.cls_hidden
            z = relu(ee @ f[h.cls_w1:h.cls_w1 + c1 * 6].reshape(c1, 6).T + f[h.cls_b1:h.cls_b1 + c1])
            return z @ f[h.cls_w2:h.cls_w2 + c1].reshape(c1, 1) + f[h.cls_b2]
        return ee @ f[h.cls_w1:h.cls_w1 + 6].reshape(6, 1) + f[h.cls_b1]

    out, hl = [], h0
    for step in range(1, L + 1):
        hin = np.concatenate([h0, hl], 1) if nf == 2 else hl
        proj = hin @ projT + projb
        pdst, psrc, q = proj[:, 0:6], proj[:, 8:14], proj[:, 16:48]
        ein = np.concatenate([e0, e], 1) if ef == 2 else e
        e = relu(psrc[row] + pdst[col] + ein @ wee.T)
        m = relu(q[row] + e @ wne.T)
        agg = params["node_agg_fn"]
        if agg == "max":
            full = np.full((n, 32), -np.inf, np.float32)
            np.maximum.at(full, row, m)
            has = np.zeros(n, bool)
            has[row] = True
            hl = np.where(has[:, None], full, 0).astype(np.float32)
        else:
            hl = np.zeros((n, 32), np.float32)
            np.add.at(hl, row, m)
            if agg == "mean":
                hl = hl / np.maximum(np.bincount(row, minlength=n), 1)[:, None].astype(np.float32)
        if step >= first:
            out.append(classify(e))
    if L == 0:
        out.append(classify(e))
    if h.fast_consts:
        fc = f[h.fast_consts:h.fast_consts + 152]
        # the three matrices are stored transposed ([in][out]): SGPR-pair operands of the packed-fp32 FMAs
        assert np.array_equal(fc[32:68].reshape(6, 6).T.ravel(), f[h.wee:h.wee + 36]) and np.array_equal(fc[104:152], projb)
        assert np.array_equal(fc[0:24].reshape(4, 6).T.ravel(), f[h.enc_edge_w:h.enc_edge_w + 24]) and fc[100] == f[h.cls_b2]
        if h.cls_w1:
            assert np.array_equal(fc[68:92].reshape(6, 4).T.ravel(), f[h.cls_w1:h.cls_w1 + 24])
    return out


@pytest.mark.parametrize("name", [n for n in golden_cases() if not n.startswith("generic_")])
def test_packed_blob_reproduces_reference(name):
    m, params, arch, sd, a = _model(name)
    out = blob_forward(m.pack_weights_host(), params, arch, a["x"], a["edge_index"], a["edge_attr"])
    assert len(out) == int(a["n_logits"])
    for i, o in enumerate(out):
        assert np.abs(o - a[f"logits_{i}"]).max() <= 5e-6, (name, i)


# ---- the device-side pack program (gnncca_pack_program), interpreted in numpy ------------------------------------
class PackSeg(C.Structure):  # mirror of csrc/internal.h: PackSeg
    _fields_ = [(n, C.c_int32) for n in ("kind", "dst", "param", "bn", "src_off", "rows", "cols", "unit0", "drs", "dcs", "srs",
                                         "scs", "plane")] + [("pad", C.c_int32 * 3)]


class PackProgram(C.Structure):
    _fields_ = [("header", BlobHeader), ("n_segs", C.c_int32), ("pad", C.c_int32 * 3), ("segs", PackSeg * 96)]


def _bf16_rne(v):
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return u.astype(np.uint16)


def run_pack_program(prog, params, nbytes):
    """What pack_device_kernel does, restated with numpy: strided copies through the BatchNorm fold (double, one rounding
    per operation), plus the three bf16 planes of the first encoder weight."""
    blob = np.zeros(nbytes // 4, dtype=np.float32)
    hdr = np.frombuffer(bytes(prog.header), dtype=np.float32)
    blob[:len(hdr)] = hdr
    u16 = blob.view(np.uint16)
    for g in list(prog.segs)[:prog.n_segs]:
        src = params[g.param].ravel()
        r, c = np.meshgrid(np.arange(g.rows), np.arange(g.cols), indexing="ij")
        v = src[g.src_off + r * g.srs + c * g.scs].astype(np.float32)
        if g.bn >= 0:
            gamma, beta, mean, var = (params[g.bn + k].astype(np.float64) for k in range(4))
            s = (gamma / np.sqrt(var + 1e-5))[g.unit0 + r]
            if g.kind == 1:
                v = ((v.astype(np.float64) - mean[g.unit0 + r]) * s + beta[g.unit0 + r]).astype(np.float32)
            else:
                v = (v.astype(np.float64) * s).astype(np.float32)
        k = r * g.drs + c * g.dcs
        if g.kind == 3:   # W_ne[ch = r][k = c] as bf16 pieces in the lane layout of the message MFMAs' B operands
            h0 = _bf16_rne(v)
            r1 = v - (h0.astype(np.uint32) << 16).view(np.float32)
            h1 = _bf16_rne(r1)
            r2 = r1 - (h1.astype(np.uint32) << 16).view(np.float32)
            h2 = _bf16_rne(r2)
            j2 = 2 * g.dst + (c // 2) * 128 + (c & 1)
            u16[j2 + 2 * r], u16[j2 + 2 * (r + 32)] = h0, h0
            u16[384 + j2 + 2 * r], u16[384 + j2 + 2 * (r + 32)] = h1, h1
            u16[768 + j2 + 2 * r], u16[768 + j2 + 2 * (r + 32)] = h2, h0
        elif g.kind == 4:   # two fp16 pieces in the swizzled chunk image of csrc/enc_f16.cuh + this element's pack-block flag word
            h0 = v.astype(np.float16)
            h1 = ((v - h0.astype(np.float32)) * np.float32(2048)).astype(np.float16)
            kq = c % 32
            kk = 2 * g.dst + (c // 32) * (2 * 128 * 32) + ((((r // 32) * 2 + kq // 16) * 64 + (r % 32) + 32 * ((kq % 16) // 8)) * 8 + (kq % 8))
            u16[kk], u16[kk + 128 * 32] = h0.view(np.uint16), h1.view(np.uint16)
            t = r * g.cols + c
            np.bitwise_or.at(blob.view(np.uint32), g.plane + (t // 256) % 64, (~(np.abs(v) < 65520.0)).astype(np.uint32))
        elif g.kind == 2:
            k = (c // 32) * 3 * g.plane + r * 32 + (c % 32)   # [in/32][3 pieces][out][32]
            h0 = _bf16_rne(v)
            r1 = v - (h0.astype(np.uint32) << 16).view(np.float32)
            h1 = _bf16_rne(r1)
            r2 = r1 - (h1.astype(np.uint32) << 16).view(np.float32)
            base = 2 * g.dst
            u16[base + k], u16[base + g.plane + k], u16[base + 2 * g.plane + k] = h0, h1, _bf16_rne(r2)
        else:
            blob[g.dst + k] = v
    return blob


@pytest.mark.parametrize("name", [n for n in golden_cases() if not n.startswith("generic_")])
def test_pack_program_reproduces_host_blob(name):
    from gnn_cca_amd import _native as nat
    m, params, arch, sd, a = _model(name)
    lib, d = nat.lib(), m.native_dims()
    assert lib.gnncca_pack_program_bytes() == C.sizeof(PackProgram)
    prog = PackProgram()
    nat.check(lib.gnncca_pack_program(C.byref(d), C.byref(prog), C.sizeof(prog)), "gnncca_pack_program")
    host = m.pack_weights_host().numpy()
    tensors = [t.detach().numpy() for t in m.native_param_tensors()]
    emulated = run_pack_program(prog, tensors, host.size)
    assert np.array_equal(emulated.view(np.uint8), host), name


def test_pack_program_refuses_generic_family():
    gen = [n for n in golden_cases() if n.startswith("generic_")]
    if not gen:
        pytest.skip("no generic golden case")
    from gnn_cca_amd import _native as nat
    m, *_ = _model(gen[0])
    d = m.native_dims()
    prog = PackProgram()
    assert nat.lib().gnncca_pack_program(C.byref(d), C.byref(prog), C.sizeof(prog)) == nat.ERR_UNSUPPORTED
