#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8f row N2 (post-processing), produced by the reference's own functions
`libs.utils.remove_edges_single_direction` (libs/utils.py:387-404) and `libs.utils.compute_SCC_and_Clusters`
(libs/utils.py:295-317), plus the sigmoid/threshold lines of inference.py:286-291, on synthetic logits over
cross-camera graphs.  Build container only:  python tests/golden/make_golden_post.py
(`cv2` / `torch_scatter` stand-ins as in make_golden_checkpoint.py; networkx is installed.)"""
import os
import sys
import types

import networkx as nx
import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _install_torch_scatter_standin, cross_camera_edges, union  # noqa: E402


def main():
    _install_torch_scatter_standin()
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, "/root/reference")
    from libs import utils  # the reference, unmodified
    from torch_scatter import scatter_add

    rng = np.random.default_rng(3)
    cases = {
        "terrace32": cross_camera_edges([8, 8, 8, 8]),
        "batch": union([("cams", [3, 3]), ("cams", [4, 4, 4]), ("cams", [2, 5, 3])]),
        "shuffled": cross_camera_edges([5, 4, 6]),
    }
    for name, (n, ei) in cases.items():
        if name == "shuffled":
            ei = ei[:, rng.permutation(ei.shape[1])]
        # identities: nodes of different cameras that share an id attract each other (positive logits), with noise
        ident = rng.integers(0, max(n // 3, 2), size=n)
        same = ident[ei[0]] == ident[ei[1]]
        logits = np.where(same, 2.0, -2.5) + rng.normal(0, 1.6, size=ei.shape[1])
        logits = logits.astype(np.float32)
        preds = torch.from_numpy(logits)
        preds_prob = torch.nn.Sigmoid()(preds)              # inference.py:289-290
        predictions = (preds_prob >= 0.5) * 1               # inference.py:291
        edge_list = ei
        active = [(edge_list[0][pos], edge_list[1][pos]) for pos, p in enumerate(predictions) if p == 1]
        id_raw, n_raw = utils.compute_SCC_and_Clusters(nx.DiGraph(active), n)
        pruned, active_p = utils.remove_edges_single_direction(active, predictions, edge_list)
        id_pruned, n_pruned = utils.compute_SCC_and_Clusters(nx.DiGraph(active_p), n)
        flow_out = scatter_add(pruned, torch.from_numpy(ei[0]), dim_size=n)   # libs/utils.py:54-55
        flow_in = scatter_add(pruned, torch.from_numpy(ei[1]), dim_size=n)
        np.savez(os.path.join(HERE, f"post_{name}.npz"), n_nodes=np.int64(n), edge_index=ei, logits=logits,
                 probs=preds_prob.numpy(), predictions=predictions.numpy(), pruned=pruned.numpy(),
                 id_pruned=id_pruned, n_clusters_pruned=np.int64(n_pruned), id_raw=id_raw, n_clusters_raw=np.int64(n_raw),
                 flow_out=flow_out.numpy(), flow_in=flow_in.numpy())
        print(f"post_{name:10s} N={n} E={ei.shape[1]} active={int(predictions.sum())} pruned={int(pruned.sum())} "
              f"clusters {n_raw} -> {n_pruned} max_flow_out={int(flow_out.max())}")


if __name__ == "__main__":
    main()
