"""Row N4 of SURVEY.md 8f: checkpoint compatibility for the MI355X module.

The reference stores training checkpoints with ``utils.save_checkpoint`` (libs/utils.py:406-424: a dict with
``model_state_dict`` next to optimizer state and metrics) and restores them with ``utils.load_pretrained_weights``
(libs/utils.py:458-507): keys lose a leading ``module.``, a tensor is taken only if its NAME AND SIZE match the model,
everything else is reported and skipped.  ``gnn_cca_amd.MOTMPNet`` keeps the reference's state_dict keys, so the
reference function works on it unchanged; this module offers the same acceptance rule without importing the reference
(as a pure planning step, ``plan_load``, plus a report object with the reason for every skipped tensor), and a converter
from a checkpoint file to the packed HBM blob.

    python -m gnn_cca_amd.checkpoint verify  CKPT.pth.tar CONFIG.yaml [arch]
    python -m gnn_cca_amd.checkpoint convert CKPT.pth.tar CONFIG.yaml OUT.blob [arch]
"""
import sys
import warnings
from collections import OrderedDict
from dataclasses import dataclass, field

import torch


DATA_PARALLEL_PREFIX = 'module.'


@dataclass
class LoadReport:
    matched: list = field(default_factory=list)     # tensor names copied into the model
    discarded: list = field(default_factory=list)   # in the checkpoint, but unknown name or different size
    missing: list = field(default_factory=list)     # in the model, not supplied by the checkpoint (kept as initialised)
    reasons: dict = field(default_factory=dict)     # discarded name -> why ('not a tensor of this model' / shapes)

    def summary(self):
        lines = [f"{len(self.matched)} tensor(s) taken from the checkpoint"]
        lines += [f"  skipped {k}: {why}" for k, why in self.reasons.items()]
        lines += [f"  left as initialised: {k}" for k in self.missing]
        return "\n".join(lines)


def extract_state_dict(checkpoint):
    """A training checkpoint (libs/utils.py:406-424) holds the weights under 'model_state_dict' next to the optimizer
    state and metrics; a bare state_dict is passed through."""
    if isinstance(checkpoint, dict) and 'model_state_dict' in checkpoint:
        return checkpoint['model_state_dict']
    return checkpoint


def plan_load(expected_shapes, state_dict):
    """Pure planning step (no tensors are touched): which checkpoint entries go where.

    `expected_shapes`: {model tensor name: shape}.  A checkpoint entry is accepted iff, after dropping one leading
    DataParallel prefix, its name is a tensor of the model AND the shapes are equal -- the acceptance rule of the
    reference's loader (libs/utils.py:458-507), which is what makes partially compatible checkpoints loadable.
    Returns (accepted: OrderedDict model name -> checkpoint tensor, LoadReport)."""
    rep, accepted = LoadReport(), OrderedDict()
    for raw_name, tensor in state_dict.items():
        name = raw_name[len(DATA_PARALLEL_PREFIX):] if raw_name.startswith(DATA_PARALLEL_PREFIX) else raw_name
        want = expected_shapes.get(name)
        have = tuple(tensor.shape) if hasattr(tensor, 'shape') else None
        if want is None:
            rep.reasons[name] = 'not a tensor of this model'
        elif have != tuple(want):
            rep.reasons[name] = f'shape {have} in the checkpoint, {tuple(want)} in the model'
        else:
            accepted[name] = tensor
    rep.matched = list(accepted)
    rep.discarded = list(rep.reasons)
    rep.missing = [name for name in expected_shapes if name not in accepted]
    return accepted, rep


def load_pretrained_weights(model, weights, verbose=True):
    """Counterpart of the reference's ``utils.load_pretrained_weights(model, weight_path)`` for this package's module:
    ``weights`` is a path, a loaded checkpoint or a state_dict.  Accepted tensors (see plan_load) are copied into the
    model's own parameters and buffers; everything else keeps its current value.  Returns ``(model, LoadReport)``.
    (``gnn_cca_amd.MOTMPNet`` keeps the reference's state_dict keys, so the reference's own function works on it too.)"""
    if isinstance(weights, (str, bytes)):
        weights = torch.load(weights, map_location='cpu', weights_only=False)
    shapes = OrderedDict((name, tuple(t.shape)) for name, t in model.state_dict().items())
    accepted, rep = plan_load(shapes, extract_state_dict(weights))
    if accepted:
        outcome = model.load_state_dict(accepted, strict=False)  # MOTMPNet.load_state_dict also drops the packed blob
        assert not outcome.unexpected_keys, outcome.unexpected_keys
    else:
        warnings.warn('no tensor of the checkpoint fits this model (names and shapes were compared after removing a '
                      f'"{DATA_PARALLEL_PREFIX}" prefix); the model is unchanged')
    if verbose and (rep.discarded or rep.missing):
        print(rep.summary())
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
