"""Row N4 of SURVEY.md 8f: checkpoint compatibility for the MI355X module.

The reference stores training checkpoints with ``utils.save_checkpoint`` (libs/utils.py:406-424: a dict with
``model_state_dict`` next to optimizer state and metrics) and restores them with ``utils.load_pretrained_weights``
(libs/utils.py:458-507): keys lose a leading ``module.``, a tensor is taken only if its NAME AND SIZE match the model,
everything else is reported and skipped, and the merged dict is loaded strictly.  ``gnn_cca_amd.MOTMPNet`` keeps the
reference's state_dict keys, so the reference function works on it unchanged; this module offers the same behaviour
without importing the reference (plus a report object), and a converter from a checkpoint file to the packed HBM blob.

    python -m gnn_cca_amd.checkpoint verify  CKPT.pth.tar CONFIG.yaml [arch]
    python -m gnn_cca_amd.checkpoint convert CKPT.pth.tar CONFIG.yaml OUT.blob [arch]
"""
import sys
import warnings
from collections import OrderedDict
from dataclasses import dataclass, field

import torch


@dataclass
class LoadReport:
    matched: list = field(default_factory=list)
    discarded: list = field(default_factory=list)   # in the checkpoint, but unknown name or different size
    missing: list = field(default_factory=list)     # in the model, not supplied by the checkpoint (kept as initialised)


def extract_state_dict(checkpoint):
    """libs/utils.py:474-477: a training checkpoint holds the weights under 'model_state_dict'."""
    if isinstance(checkpoint, dict) and 'model_state_dict' in checkpoint:
        return checkpoint['model_state_dict']
    return checkpoint


def load_pretrained_weights(model, weights, verbose=True):
    """Same contract as the reference's ``utils.load_pretrained_weights(model, weight_path)``; ``weights`` may be a
    path or an already loaded checkpoint / state_dict.  Returns ``(model, LoadReport)``."""
    if isinstance(weights, (str, bytes)):
        weights = torch.load(weights, map_location='cpu', weights_only=False)
    state_dict = extract_state_dict(weights)
    model_dict = model.state_dict()
    taken, rep = OrderedDict(), LoadReport()
    for k, v in state_dict.items():
        if k.startswith('module.'):
            k = k[7:]  # written by nn.DataParallel
        if k in model_dict and model_dict[k].size() == v.size():
            taken[k] = v
            rep.matched.append(k)
        else:
            rep.discarded.append(k)
    rep.missing = [k for k in model_dict if k not in taken]
    model_dict.update(taken)
    model.load_state_dict(model_dict, strict=True)
    if not rep.matched:
        warnings.warn('The pretrained weights cannot be loaded, please check the key names manually '
                      '(** ignored and continue **)')
    elif verbose and rep.discarded:
        print('** The following layers are discarded due to unmatched keys or layer size: {}'.format(rep.discarded))
    return model, rep


def checkpoint_to_blob(weights, model_params, arch):
    """Checkpoint -> the packed weight blob of gnncca_pack_weights (CPU uint8 tensor) + the load report."""
    from .mpn import MOTMPNet
    model = MOTMPNet(model_params, None, arch).eval()
    model, rep = load_pretrained_weights(model, weights, verbose=False)
    return model.pack_weights_host(), rep


def _main(argv):
    import yaml
    if len(argv) < 4 or argv[1] not in ('verify', 'convert'):
        print(__doc__)
        return 2
    with open(argv[3]) as f:
        cfg = yaml.safe_load(f)
    params = cfg['GRAPH_NET_PARAMS']
    if argv[1] == 'convert':
        arch = argv[5] if len(argv) > 5 else cfg['CNN_MODEL']['arch']
    else:
        arch = argv[4] if len(argv) > 4 else cfg['CNN_MODEL']['arch']
    blob, rep = checkpoint_to_blob(argv[2], params, arch)
    print(f"matched {len(rep.matched)} tensors; discarded {rep.discarded}; not in checkpoint {rep.missing}")
    if argv[1] == 'convert':
        with open(argv[4], 'wb') as f:
            f.write(blob.numpy().tobytes())
        print(f"wrote {blob.numel()} bytes to {argv[4]}")
    return 0 if rep.matched and not rep.missing else 1


if __name__ == '__main__':
    sys.exit(_main(sys.argv))
