"""Parameter container with the module/state_dict layout of the reference's ``models/mlp.py:4-28``.

The reference MLP is an ``nn.Sequential`` named ``fc_layers`` whose indices depend on which of
Linear / BatchNorm1d / ReLU / Dropout each width contributes; checkpoints written by the reference
(``utils.save_checkpoint``, libs/utils.py:406-424) address parameters by those indices
(``fc_layers.0.weight``, ``fc_layers.3.weight`` ...), so the same Sequential is rebuilt here.  The arithmetic is
NOT done by these torch modules: ``MOTMPNet.forward`` hands the parameters to the HIP kernels.
"""
from torch import nn


def layer_plan(input_dim, fc_dims, dropout_p, use_batchnorm):
    """One entry per Linear: (in, out, has_bn, relu, has_dropout) following models/mlp.py:11-24."""
    assert isinstance(fc_dims, (list, tuple)), \
        'fc_dims must be either a list or a tuple, but got {}'.format(type(fc_dims))  # mlp.py:8 (same message)
    plan = []
    for width in fc_dims:
        wide = width != 1  # a width-1 output layer is a bare Linear (mlp.py:14,17,20)
        plan.append((input_dim, width, bool(use_batchnorm) and wide, wide, dropout_p is not None and wide))
        input_dim = width
    return plan


class MLP(nn.Module):
    def __init__(self, input_dim, fc_dims, dropout_p=0.4, use_batchnorm=False):
        super().__init__()
        self.plan = layer_plan(input_dim, fc_dims, dropout_p, use_batchnorm)
        mods, self.linear_index, self.bn_index = [], [], []
        for fan_in, width, has_bn, relu, has_do in self.plan:
            self.linear_index.append(len(mods))
            mods.append(nn.Linear(fan_in, width))
            self.bn_index.append(len(mods) if has_bn else None)
            if has_bn:
                mods.append(nn.BatchNorm1d(width))
            if relu:
                mods.append(nn.ReLU(inplace=True))
            if has_do:
                mods.append(nn.Dropout(p=dropout_p))
        self.fc_layers = nn.Sequential(*mods)

    def native_params(self):
        """Tensors in the order gnncca_pack_weights expects (include/gnncca_mpn.h): per layer weight, bias and,
        with BatchNorm, gamma, beta, running_mean, running_var."""
        out = []
        for li, bi in zip(self.linear_index, self.bn_index):
            lin = self.fc_layers[li]
            out += [lin.weight, lin.bias]
            if bi is not None:
                bn = self.fc_layers[bi]
                out += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        return out

    def forward(self, input):
        """models/mlp.py:26-28 called on its own (MOTMPNet.forward never comes through here: its arithmetic is fused):
        gnncca_mlp_eval, eval semantics, GPU tensors only."""
        from . import mpn
        return mpn.standalone_mlp(self, input)
