"""ctypes wrapper of oracle/mpn_oracle_c.c (the fused C / OpenMP CPU restatement).  TEST INFRASTRUCTURE -- NOT PRODUCT CODE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it (see the header of the C file).

    build()                      compile oracle/_build/libmpn_oracle_c.so with gcc -O3 -fopenmp (called by __graft_entry__.build)
    COracle(params, arch, sd).forward(x, edge_index, edge_attr) -> list of [E, 1] float32 arrays (like NumpyOracle)
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .mpn_oracle import model_layout

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mpn_oracle_c.c")
LIB = os.path.join(HERE, "_build", "libmpn_oracle_c.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # x86-64-v3 (AVX2 + FMA), not -march=native: the library is built in the build container and RUN on the GPU box's host
    subprocess.run(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-Wall", SRC, "-o", LIB + ".tmp", "-lm"],
                   check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


class _Layer(C.Structure):
    _fields_ = [("in_dim", C.c_int), ("out_dim", C.c_int), ("has_bn", C.c_int), ("relu", C.c_int),
                ("W", C.c_void_p), ("b", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p),
                ("var", C.c_void_p)]


class _Mlp(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("layers", _Layer * 8)]


class _Model(C.Structure):
    _fields_ = [("enc_node", _Mlp), ("enc_edge", _Mlp), ("edge_mlp", _Mlp), ("node_mlp", _Mlp), ("cls_edge", _Mlp),
                ("agg", C.c_int), ("L", C.c_int), ("n_cls", C.c_int), ("reattach_nodes", C.c_int), ("reattach_edges", C.c_int)]


class COracle:
    def __init__(self, model_params, arch, sd, threads=None):
        self.lib = C.CDLL(build())
        self.lib.oc_forward.restype = C.c_int
        self.lay = model_layout(model_params, arch)
        self.sd = {k: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in sd.items() if "num_batches_tracked" not in k}
        self.model = _Model()
        for field, prefix in (("enc_node", "encoder.node_mlp"), ("enc_edge", "encoder.edge_mlp"),
                              ("edge_mlp", "MPNet.edge_model.edge_mlp"), ("node_mlp", "MPNet.node_model.node_mlp"),
                              ("cls_edge", "classifier.edge_mlp")):
            self._fill(getattr(self.model, field), prefix)
        self.model.agg = {"sum": 0, "mean": 1, "max": 2}[self.lay["agg"].lower()]
        self.model.L, self.model.n_cls = int(self.lay["L"]), int(self.lay["n_cls"])
        self.model.reattach_nodes, self.model.reattach_edges = int(bool(self.lay["reattach_nodes"])), int(bool(self.lay["reattach_edges"]))
        if threads:
            os.environ["OMP_NUM_THREADS"] = str(threads)

    def _fill(self, mlp, prefix):
        layers = self.lay[prefix] or []
        if len(layers) > 8:
            raise NotImplementedError("more than 8 layers")
        mlp.n_layers = len(layers)
        for i, (lin, fan_in, width, bn, relu) in enumerate(layers):
            L = mlp.layers[i]
            p = f"{prefix}.fc_layers.{lin}."
            L.in_dim, L.out_dim, L.has_bn, L.relu = fan_in, width, int(bn is not None), int(relu)
            L.W, L.b = self.sd[p + "weight"].ctypes.data, self.sd[p + "bias"].ctypes.data
            if bn is not None:
                q = f"{prefix}.fc_layers.{bn}."
                L.gamma, L.beta = self.sd[q + "weight"].ctypes.data, self.sd[q + "bias"].ctypes.data
                L.mean, L.var = self.sd[q + "running_mean"].ctypes.data, self.sd[q + "running_var"].ctypes.data

    def supported(self):
        em, nm = self.lay["MPNet.edge_model.edge_mlp"], self.lay["MPNet.node_model.node_mlp"]
        return len(em) == 1 and len(nm) == 1 and em[0][3] is None and nm[0][3] is None and \
            self.lay["MPNet.node_model.node_mlp"][0][2] <= 128

    def forward(self, x, edge_index, edge_attr):
        x = np.ascontiguousarray(x, dtype=np.float32)
        ei = np.ascontiguousarray(edge_index, dtype=np.int64)
        ea = np.ascontiguousarray(edge_attr, dtype=np.float32)
        n, e = x.shape[0], ei.shape[1]
        n_out = max(1, min(int(self.lay["L"]), int(self.lay["n_cls"]))) if int(self.lay["L"]) > 0 else 1
        out = np.zeros((n_out, max(e, 1)), dtype=np.float32)
        rc = self.lib.oc_forward(C.byref(self.model), C.c_void_p(x.ctypes.data), C.c_void_p(ei.ctypes.data), C.c_void_p(ea.ctypes.data),
                                 C.c_int64(n), C.c_int64(e), C.c_int(x.shape[1]), C.c_int(ea.shape[1] if ea.ndim == 2 else 0),
                                 C.c_void_p(out.ctypes.data))
        if rc < 0:
            raise RuntimeError(f"oc_forward failed ({rc})")
        return [out[i, :e].reshape(e, 1).copy() for i in range(rc)]
