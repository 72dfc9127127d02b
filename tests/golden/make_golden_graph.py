#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8f row N1 (graph construction + edge attributes), produced by the REFERENCE's own
statements.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_graph.py

The reference has no function for this step: it is inline code inside
`inference.py: validate_GNN_cross_camera_association` (lines 189-279; duplicated at train.py:257-361, 616-692), in a
module that cannot be imported here (it needs torch_geometric / cv2 / torchreid at import time).  So this script
reads those lines from /root/reference/inference.py AT RUN TIME, dedents them and executes them unmodified against
synthetic per-frame detection tables; nothing of the reference's text is stored in this repository -- only the
numeric inputs and outputs (.npz).  Stand-ins supplied to the executed statements:

  * `Tensor.cuda()` is the identity (no GPU in the build container);
  * `Data` / `Batch.from_data_list` restate torch_geometric 2.0.1's published behaviour for the attributes used:
    per-graph containers, concatenation along dim 0 with `edge_index` shifted by the running node count;
  * `paired_distances` is scikit-learn's own function (installed here), `F` is torch.nn.functional.
"""
import os
import sys
import textwrap

import numpy as np
import pandas as pd
import torch
import torch.nn.functional as F
from sklearn.metrics.pairwise import paired_distances

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILE = "/root/reference/inference.py"
FIRST_LINE, LAST_LINE = 189, 279  # F.normalize(...) ... data_batch = Batch.from_data_list(batch)


class Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Batch(Data):
    @staticmethod
    def from_data_list(items):
        out, off, eis = Batch(), 0, []
        for d in items:
            eis.append(d.edge_index + off)
            off += d.x.shape[0]
        out.x = torch.cat([d.x for d in items], dim=0)
        out.edge_index = torch.cat(eis, dim=1)
        out.edge_attr = torch.cat([d.edge_attr for d in items], dim=0)
        out.edge_labels = torch.cat([d.edge_labels for d in items], dim=0)
        out.y = torch.cat([d.y for d in items], dim=0)
        out.num_nodes = off
        return out


def reference_statements():
    with open(REF_FILE) as f:
        lines = f.readlines()[FIRST_LINE - 1:LAST_LINE]
    return textwrap.dedent("".join(lines))


def run_reference(frames, node_embeds, reid_embeds, max_dist, only_appearance=False, only_dist=False):
    ns = {
        "np": np, "torch": torch, "F": F, "paired_distances": paired_distances, "Data": Data, "Batch": Batch,
        "CONFIG": {"TRAINING": {"ONLY_APPEARANCE": only_appearance, "ONLY_DIST": only_dist}},
        "data_df": [f.copy() for f in frames], "len_graphs": [len(f) for f in frames],
        "node_embeds": node_embeds.clone(), "reid_embeds": reid_embeds.clone(), "max_dist": list(max_dist),
        "print": lambda *a, **k: None,
    }
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        exec(compile(reference_statements(), "<reference inference.py:189-279>", "exec"), ns)
    finally:
        torch.Tensor.cuda = saved
    return ns["data_batch"], ns["node_embeds"], ns["reid_embeds"]


def make_frame(rng, cam_ids, n_people, frame_no, shuffle=False):
    """One frame's detections: columns as libs/datasets.py builds them (only the ones this step reads matter)."""
    cam_ids = np.asarray(cam_ids)
    n = len(cam_ids)
    ids = rng.integers(0, n_people, size=n)
    pos = rng.uniform(-8, 8, size=(n_people, 2))
    xy = pos[ids] + rng.normal(0, 0.4, size=(n, 2))
    df = pd.DataFrame({"frame": frame_no, "id": ids.astype(np.int64), "id_cam": cam_ids.astype(np.int64),
                       "xw": xy[:, 0], "yw": xy[:, 1]})
    if shuffle:
        df = df.iloc[rng.permutation(n)].reset_index(drop=True)
    return df


def save_case(name, frames, d_node, d_reid, max_dist, seed, **modes):
    rng = np.random.default_rng(seed)
    n_tot = sum(len(f) for f in frames)
    g = torch.Generator().manual_seed(seed)
    node_raw = torch.randn(n_tot, d_node, generator=g)
    reid_raw = torch.randn(n_tot, d_reid, generator=g) + 0.5
    batch, node_n, reid_n = run_reference(frames, node_raw, reid_raw, max_dist, **modes)
    rec = {
        "n_graphs": np.int64(len(frames)), "graph_sizes": np.array([len(f) for f in frames], dtype=np.int64),
        "xw": np.concatenate([f["xw"].values for f in frames]), "yw": np.concatenate([f["yw"].values for f in frames]),
        "id": np.concatenate([f["id"].values for f in frames]), "id_cam": np.concatenate([f["id_cam"].values for f in frames]),
        "max_dist": np.asarray(max_dist, dtype=np.float64),
        "node_embeds_raw": node_raw.numpy(), "reid_embeds_raw": reid_raw.numpy(),
        "node_embeds": node_n.numpy(), "reid_embeds": reid_n.numpy(),  # after the reference's F.normalize(dim=0)
        "only_appearance": np.bool_(modes.get("only_appearance", False)), "only_dist": np.bool_(modes.get("only_dist", False)),
        "x": batch.x.numpy(), "edge_index": batch.edge_index.numpy(), "edge_attr": batch.edge_attr.numpy(),
        "edge_labels": batch.edge_labels.numpy(), "y": batch.y.numpy(),
    }
    np.savez(os.path.join(HERE, "graph_" + name + ".npz"), **rec)
    ei = rec["edge_index"]
    print(f"graph_{name:22s} graphs={len(frames)} N={n_tot:4d} E={ei.shape[1]:6d} attr={rec['edge_attr'].shape[1]} "
          f"row_sorted={bool(np.all(np.diff(ei[0]) >= 0))} positives={int(rec['edge_labels'].sum())}")
    del rng


def main():
    rng = np.random.default_rng(1)
    f1 = make_frame(rng, [0] * 4 + [1] * 3 + [2] * 5, 6, 10)
    save_case("one_frame", [f1], 8, 256, [37.5], 201)
    frames = [make_frame(rng, [0] * 3 + [2] * 2, 4, 1), make_frame(rng, [0] * 5 + [1] * 4 + [2] * 6 + [3] * 3, 8, 2),
              make_frame(rng, [1] * 2 + [3] * 2, 3, 3)]
    save_case("batch3", frames, 8, 32, [20.0, 33.0, 7.0], 202)
    fs = make_frame(rng, [0, 1, 0, 2, 1, 0, 2, 2, 1], 5, 4, shuffle=True)  # cameras interleaved -> `row` not sorted
    save_case("interleaved", [fs, make_frame(rng, [0] * 2 + [1] * 2, 2, 5)], 8, 16, [15.0, 15.0], 203)
    save_case("only_appearance", [f1], 8, 16, [37.5], 204, only_appearance=True)
    save_case("only_dist", [f1], 8, 16, [37.5], 205, only_dist=True)
    big = make_frame(rng, sum(([c] * 8 for c in range(4)), []), 12, 6)  # the Terrace-shaped 4 x 8 frame
    save_case("terrace32", [big], 8, 256, [80.0], 206)


if __name__ == "__main__":
    main()
