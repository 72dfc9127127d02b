"""Row N4 on the GPU: a checkpoint file (libs/utils.py:406-424) converted to the packed HBM blob
(`gnn_cca_amd.checkpoint.checkpoint_to_blob`), written to disk, installed with `MOTMPNet.load_packed_blob` on a CUDA module
and run through the HIP forward -- against (a) a module that loaded the same checkpoint through `load_pretrained_weights`
(bitwise) and (b) the logits of the REFERENCE's own model after the reference's own `utils.load_pretrained_weights`
(libs/utils.py:458-507; tests/golden/make_golden_checkpoint.py stores them).  GPU only."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu

TOL_TIGHT = 5e-6


class Data:
    def __init__(self, x, edge_index, edge_attr):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr


def _params(cls_bn):
    z = np.load(os.path.join(GOLDEN_DIR, "n8_sum.npz"), allow_pickle=False)
    meta = json.loads(str(z["params_json"]))
    params = copy.deepcopy(meta["model_params"])
    params["classifier_feats_dict"]["use_batchnorm"] = cls_bn
    return params, meta["arch"]


def _case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    ckpt = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("ckpt::")}
    init = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init::")}
    want = {k[8:]: z[k] for k in z.files if k.startswith("loaded::")}
    data = Data(*(torch.from_numpy(z[k]).cuda() for k in ("x", "edge_index", "edge_attr")))
    logits = [z[f"logits_{i}"] for i in range(3)]
    return ckpt, init, want, data, logits


@pytest.mark.parametrize("name,cls_bn,n_discarded", [("ckpt_module_prefix", True, 2), ("ckpt_bn_variant", False, 7)])
def test_checkpoint_to_blob_to_hip_forward(tmp_path, name, cls_bn, n_discarded):
    from gnn_cca_amd import MOTMPNet
    from gnn_cca_amd.checkpoint import checkpoint_to_blob, load_pretrained_weights
    ckpt, init, want, data, ref_logits = _case(name)
    params, arch = _params(cls_bn)
    ckpt_path = os.path.join(tmp_path, "run_best.pth.tar")
    torch.save({"epoch": 12, "model_state_dict": ckpt, "prec": 91.0}, ckpt_path)   # what utils.save_checkpoint writes

    # (a) the usual route: module, loader, .cuda().eval(), forward (packs on first use)
    torch.manual_seed(1234)
    a = MOTMPNet(copy.deepcopy(params), None, arch)
    a.load_state_dict(init, strict=True)                  # the starting point the reference's target model had
    a, rep = load_pretrained_weights(a, ckpt_path, verbose=False)
    assert len(rep.discarded) == n_discarded
    for k in want:                                        # same state as the reference's loader left ITS model in
        assert np.array_equal(a.state_dict()[k].numpy(), want[k]), k
    a = a.cuda().eval()
    with torch.no_grad():
        out_a = [t.clone() for t in a(data)["classified_edges"]]

    # (b) the converter's route: checkpoint -> blob -> bytes on disk -> load_packed_blob on a CUDA module whose own
    # parameters are DIFFERENT (so only the installed blob can explain the result)
    blob, rep_b = checkpoint_to_blob({"model_state_dict": a.cpu().state_dict()}, copy.deepcopy(params), arch)
    a = a.cuda()
    assert not rep_b.discarded and not rep_b.missing
    _, rep_c = checkpoint_to_blob(ckpt_path, copy.deepcopy(params), arch)    # the raw file: same verdict as the loader's
    assert sorted(rep_c.discarded) == sorted(rep.discarded) and rep_c.missing == rep.missing
    blob_path = os.path.join(tmp_path, "run_best.blob")
    with open(blob_path, "wb") as f:
        f.write(blob.numpy().tobytes())
    torch.manual_seed(999)
    b = MOTMPNet(copy.deepcopy(params), None, arch).cuda().eval()
    assert any(not torch.equal(p, q) for p, q in zip(a.state_dict().values(), b.state_dict().values()))
    b.load_packed_blob(blob_path)
    assert b._packed[1].is_cuda
    with torch.no_grad():
        out_b = b(data)["classified_edges"]
    torch.cuda.synchronize()
    assert len(out_a) == len(out_b) == len(ref_logits)
    for i, (x, y, r) in enumerate(zip(out_a, out_b, ref_logits)):
        assert torch.equal(x, y), (name, i)                                   # blob route == loader route, bit for bit
        assert tuple(y.shape) == r.shape
        err = np.abs(y.cpu().numpy() - r).max()
        assert err <= TOL_TIGHT, (name, i, err)                               # == the reference after ITS loader
    # the device blob equals what the module would pack itself (host and device packers are byte-identical)
    assert torch.equal(b._packed[1].cpu(), blob)
    assert torch.equal(a._packed_weights(torch.device("cuda", torch.cuda.current_device())).cpu(), blob)


def test_cli_convert_then_forward(tmp_path):
    """`python -m gnn_cca_amd.checkpoint convert CKPT CONFIG OUT.blob` (run in-process) writes the blob a CUDA module accepts."""
    import yaml

    from gnn_cca_amd import MOTMPNet
    from gnn_cca_amd import checkpoint as ck
    ckpt, init, want, data, ref_logits = _case("ckpt_module_prefix")
    params, arch = _params(True)
    full = {**init, **{k: torch.from_numpy(v) for k, v in want.items()}}     # the reference's loaded state as a checkpoint
    ckpt_path, cfg_path, blob_path = (os.path.join(tmp_path, n) for n in ("c.pth.tar", "c.yaml", "c.blob"))
    torch.save({"model_state_dict": {"module." + k: v for k, v in full.items()}}, ckpt_path)
    with open(cfg_path, "w") as f:
        yaml.safe_dump({"GRAPH_NET_PARAMS": copy.deepcopy(params), "CNN_MODEL": {"arch": arch}}, f)
    assert ck._main(["checkpoint", "convert", ckpt_path, cfg_path, blob_path]) == 0
    m = MOTMPNet(copy.deepcopy(params), None, arch).cuda().eval().load_packed_blob(blob_path)
    with torch.no_grad():
        out = m(data)["classified_edges"]
    for i, (y, r) in enumerate(zip(out, ref_logits)):
        assert np.abs(y.cpu().numpy() - r).max() <= TOL_TIGHT, i


def test_packed_blob_refusals_on_a_cuda_module():
    """Stale header, wrong size and another configuration are refused BEFORE anything reaches the GPU; the module keeps working."""
    from gnn_cca_amd import MOTMPNet
    _, init, _, data, _ = _case("ckpt_module_prefix")
    params, arch = _params(True)
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict(init, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        before = [t.clone() for t in m(data)["classified_edges"]]
    good = m.pack_weights_host()
    stale = good.clone()
    stale[0] ^= 0x01
    with pytest.raises(RuntimeError, match="different build"):
        m.load_packed_blob(stale)
    with pytest.raises(RuntimeError, match="bytes"):
        m.load_packed_blob(good[:-4])
    with pytest.raises(RuntimeError, match="too short"):
        m.load_packed_blob(b"\x00" * 8)
    from oracle.mpn_oracle import load_case
    p_big, arch_big, _, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    other = MOTMPNet(copy.deepcopy(p_big), None, arch_big).cuda().eval()
    with pytest.raises(RuntimeError):
        other.load_packed_blob(good)                       # another GRAPH_NET_PARAMS (node_in 2048): another size
    with torch.no_grad():
        after = m(data)["classified_edges"]
    assert all(torch.equal(x, y) for x, y in zip(before, after))
    # a later load_state_dict supersedes an installed blob (the cache is keyed on the parameters' versions)
    m.load_packed_blob(good.numpy().tobytes())
    sd = {k: (v + 0.25 if v.dtype.is_floating_point else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    with torch.no_grad():
        moved = m(data)["classified_edges"]
    assert not torch.equal(moved[-1], before[-1])
