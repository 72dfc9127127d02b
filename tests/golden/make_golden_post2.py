#!/usr/bin/env python3
"""Golden vectors for the REST of SURVEY.md 8f row N2: the degree-constraint rounding (`libs.utils.compute_rounding`,
libs/utils.py:25-173) and the big-cluster splitting (`libs.utils.disjoint_big_clusters`, libs/utils.py:319-386) in the order
the shipped inference configuration runs them (config_inference.yaml:6-8 ROUNDING / PRUNING / SPLITTING = True):
threshold -> PRUNING -> ROUNDING -> PRUNING -> SPLITTING (inference.py:286-345).

That call sequence is inline code of `inference.py: validate_GNN_cross_camera_association`, in a module that cannot be
imported here (torch_geometric / cv2 / torchreid at import time).  As tests/golden/make_golden_graph.py does for row N1, this
script READS those lines from /root/reference/inference.py at run time, dedents them and executes them unmodified -- with
`mpn_model` a stand-in that returns the case's logits -- against the reference's own `libs.utils` (imported, unmodified;
`cv2` / `torch_scatter` stand-ins as in make_golden_post.py; networkx is installed).  Nothing of the reference's text is
stored in this repository: only the numeric inputs and outputs (.npz).  One frame per case (the reference validates with
batch size 1: main.py:368 / config_inference.yaml:53-54).  Build container only:

    python tests/golden/make_golden_post2.py
"""
import os
import sys
import textwrap
import types

import networkx as nx
import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _install_torch_scatter_standin, cross_camera_edges  # noqa: E402

REF_FILE = "/root/reference/inference.py"
FIRST_LINE, LAST_LINE = 283, 345   # outputs = mpn_model(data_batch) ... ID_pred after SPLITTING


def reference_statements():
    with open(REF_FILE) as f:
        lines = f.readlines()[FIRST_LINE - 1:LAST_LINE]
    return textwrap.dedent("".join(lines))


class Batch:
    pass


def run_reference(utils, n, ei, logits, rounding, pruning, splitting):
    b = Batch()
    b.edge_index = torch.from_numpy(ei)
    b.num_nodes = n
    b.edge_labels = torch.zeros(ei.shape[1])
    ns = {"np": np, "torch": torch, "nx": nx, "utils": utils, "data_batch": b,
          "CONFIG": {"ROUNDING": rounding, "PRUNING": pruning, "SPLITTING": splitting},
          "mpn_model": lambda batch: {"classified_edges": [torch.from_numpy(logits).view(-1, 1)]}}
    exec(compile(reference_statements(), "<reference inference.py:283-345>", "exec"), ns)
    pred = ns["predictions"]
    pred = pred.numpy() if torch.is_tensor(pred) else np.asarray(pred)
    return pred.astype(np.int64).reshape(-1), np.asarray(ns["ID_pred"]).astype(np.int64), ns["preds_prob"].numpy()


def logits_from_pairs(ei, pairs, rng, off=-4.0, noise=0.3):
    """Negative logits everywhere except on the listed undirected pairs {(i, j): (logit i->j, logit j->i)}."""
    logit = (off + noise * rng.standard_normal(ei.shape[1])).astype(np.float32)
    pos = {(int(a), int(b)): k for k, (a, b) in enumerate(ei.T)}
    for (i, j), (lij, lji) in pairs.items():
        logit[pos[(i, j)]] = lij
        logit[pos[(j, i)]] = lji
    return logit


def main():
    _install_torch_scatter_standin()
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, "/root/reference")
    from libs import utils  # the reference, unmodified

    rng = np.random.default_rng(11)
    cases = {}

    def pairs_with(rng_, plist, lo=1.0, hi=4.0):
        return {p: (float(rng_.uniform(lo, hi)), float(rng_.uniform(lo, hi))) for p in plist}

    # six cameras x two detections: node v sits in camera v // 2, so any two nodes of different cameras share both edges
    n, ei = cross_camera_edges([2] * 6)
    # a star: hub 0 with four leaves -> flow 4 > 3, every edge a bridge (rounding, "if there are bridges" arm)
    cases["star_bridges"] = (n, ei, logits_from_pairs(ei, pairs_with(rng, [(0, 2), (0, 4), (0, 6), (0, 8)]), rng))
    # K5 on nodes 0, 2, 4, 6, 8: every node has flow 4, no bridges (rounding, argmin arm), and a cluster of five
    k5 = [(a, b) for i, a in enumerate([0, 2, 4, 6, 8]) for b in [0, 2, 4, 6, 8][i + 1:]]
    cases["k5_no_bridges"] = (n, ei, logits_from_pairs(ei, pairs_with(rng, k5), rng))
    # two triangles joined by one edge: flows <= 3 (no rounding), a cluster of six with a bridge (splitting, bridge arm)
    tri = [(0, 2), (2, 4), (0, 4), (6, 8), (8, 10), (6, 10), (4, 6)]
    cases["two_triangles_bridge"] = (n, ei, logits_from_pairs(ei, pairs_with(rng, tri), rng))
    # a ring of five: no bridges, a cluster of five (splitting, no-bridge arm; the opened ring is then all bridges)
    ring = [(0, 2), (2, 4), (4, 6), (6, 8), (0, 8)]
    cases["ring5"] = (n, ei, logits_from_pairs(ei, pairs_with(rng, ring), rng))
    # a ring of six next to a pair whose only edge is a bridge of LOWER probability than anything in the ring
    ring6 = [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (0, 10)]
    pr = pairs_with(rng, ring6, 2.0, 4.0)
    pr[(1, 3)] = (0.4, 0.3)
    cases["ring6_and_weak_pair"] = (n, ei, logits_from_pairs(ei, pr, rng))
    # two big clusters (the recursion of disjoint_big_clusters): rings of five on the even and on the odd nodes
    two = [(0, 2), (2, 4), (4, 6), (6, 8), (0, 8), (1, 3), (3, 5), (5, 7), (7, 9), (1, 9)]
    cases["two_rings5"] = (n, ei, logits_from_pairs(ei, pairs_with(rng, two), rng))
    # hub with a bridge leaf inside a dense blob: flow > 3 at a node that has bridge AND non-bridge edges
    blob = [(0, 2), (0, 4), (0, 6), (2, 4), (2, 6), (4, 6), (0, 8)]
    cases["blob_with_leaf"] = (n, ei, logits_from_pairs(ei, pairs_with(rng, blob), rng))
    # in-flow and out-flow violations that survive the first pruning only partly: asymmetric probabilities around 0.5
    asym = {p: (float(rng.uniform(-0.3, 3.0)), float(rng.uniform(-0.3, 3.0))) for p in k5 + [(1, 2), (1, 4), (3, 6)]}
    cases["asymmetric_k5"] = (n, ei, logits_from_pairs(ei, asym, rng, noise=1.0))
    # nothing to do: two clean pairs and a triangle
    cases["clean"] = (n, ei, logits_from_pairs(ei, pairs_with(rng, [(0, 2), (4, 6), (1, 3), (3, 5), (1, 5)]), rng))
    # noisy identity structure on Terrace-shaped frames (four cameras) and on bigger ones, several seeds
    for name, cams, seeds in (("terrace", [8, 8, 8, 8], (0, 1, 2)), ("cams6", [3, 4, 2, 5, 3, 4], (3, 4, 5)), ("cams5x3", [3] * 5, (6, 7, 8, 9))):
        for s in seeds:
            r = np.random.default_rng(100 + s)
            nn, e2 = cross_camera_edges(cams)
            ident = r.integers(0, max(nn // 4, 2), size=nn)
            same = ident[e2[0]] == ident[e2[1]]
            lg = (np.where(same, 2.0, -2.8) + r.normal(0, 1.5, size=e2.shape[1])).astype(np.float32)
            cases[f"{name}_s{s}"] = (nn, e2, lg)

    out, summary = {}, []
    for name, (nn, e2, lg) in cases.items():
        p_all, id_all, probs = run_reference(utils, nn, e2, lg, True, True, True)          # the shipped configuration
        p_pr, id_pr, _ = run_reference(utils, nn, e2, lg, True, True, False)                # ... stopped before SPLITTING
        p_p, id_p, _ = run_reference(utils, nn, e2, lg, False, True, False)                 # PRUNING only (the device chain of round 4)
        p_s, id_s, _ = run_reference(utils, nn, e2, lg, False, True, True)                  # no ROUNDING
        for k, v in (("n_nodes", np.int64(nn)), ("edge_index", e2), ("logits", lg), ("probs", probs),
                     ("pred_final", p_all), ("id_final", id_all), ("pred_rounded", p_pr), ("id_rounded", id_pr),
                     ("pred_pruned", p_p), ("id_pruned", id_p), ("pred_split_only", p_s), ("id_split_only", id_s)):
            out[f"{name}::{k}"] = v
        summary.append(f"{name:22s} N={nn:2d} E={e2.shape[1]:4d} active: pruned {int(p_p.sum()):3d} rounded {int(p_pr.sum()):3d} final {int(p_all.sum()):3d}  "
                       f"clusters: pruned {len(set(id_p.tolist())):2d} (max {np.bincount(id_p).max()}) rounded {len(set(id_pr.tolist())):2d} "
                       f"final {len(set(id_all.tolist())):2d} (max {np.bincount(id_all).max()})")
    np.savez_compressed(os.path.join(HERE, "post2_heuristics.npz"), names=np.array(sorted(cases)), **out)
    print("\n".join(summary))


if __name__ == "__main__":
    main()
