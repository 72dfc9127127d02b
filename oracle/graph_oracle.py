"""CPU ORACLE for SURVEY.md 8f row N1 (graph construction + edge attributes).  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Parity status: PINNED by tests/golden/graph_*.npz, which hold the outputs of the reference's own statements
(inference.py:189-279, executed by tests/golden/make_golden_graph.py on synthetic frames).

Restates, per frame graph g (reference lines relative to /root/reference):
  inference.py:189-190  reid / node embeddings L2-normalised over dim 0 (per feature column, across ALL nodes of the batch)
  inference.py:207-212  for each camera id in np.unique order: cartesian_prod(nodes in cam, nodes in the other cams)
  inference.py:222-226  F.pairwise_distance (p=2, eps=1e-6) and F.cosine_similarity (eps=1e-8) of the reid rows, fp32
  inference.py:229-242  ground-plane L2 / L1 distances in float64 (sklearn paired_distances), divided by max_dist[g],
                        then cast to float32
  inference.py:245-256  edge_attr = [l2/max, l1/max, emb_dist, emb_cos]  (ONLY_APPEARANCE: last two; ONLY_DIST: first two)
  inference.py:262-266  edge_labels = 1.0 where both detections carry the same person id
  inference.py:269-279  node ids local to the graph, then Batch.from_data_list re-offsets them: global ids again
"""
import numpy as np


def normalize_columns(x):
    """F.normalize(x, p=2, dim=0): x / max(||column||_2, 1e-12)."""
    x = np.asarray(x, dtype=np.float32)
    nrm = np.sqrt((x.astype(np.float32) ** 2).sum(axis=0, dtype=np.float32))
    return (x / np.maximum(nrm, np.float32(1e-12))).astype(np.float32)


def edge_list(id_cam, graph_sizes):
    """edge_index [2, E] (global node ids) in the reference's order."""
    rows, cols, off = [], [], 0
    for n in graph_sizes:
        cams = np.asarray(id_cam[off:off + n])
        nodes = np.arange(off, off + n)
        for c in np.unique(cams):
            inside, outside = nodes[cams == c], nodes[cams != c]
            rows.append(np.repeat(inside, len(outside)))
            cols.append(np.tile(outside, len(inside)))
        off += n
    return np.stack([np.concatenate(rows), np.concatenate(cols)]).astype(np.int64)


def build(xw, yw, ids, id_cam, graph_sizes, max_dist, reid_embeds, only_appearance=False, only_dist=False):
    """reid_embeds: already normalised [N, R] float32.  Returns edge_index, edge_attr (float32), edge_labels (float32)."""
    ei = edge_list(id_cam, graph_sizes)
    r, c = ei
    graph_of = np.repeat(np.arange(len(graph_sizes)), graph_sizes)
    md = np.asarray(max_dist, dtype=np.float64)[graph_of[r]]
    dx = np.asarray(xw, np.float64)[r] - np.asarray(xw, np.float64)[c]
    dy = np.asarray(yw, np.float64)[r] - np.asarray(yw, np.float64)[c]
    l2 = (np.sqrt(dx * dx + dy * dy) / md).astype(np.float32)
    l1 = ((np.abs(dx) + np.abs(dy)) / md).astype(np.float32)
    a, b = np.asarray(reid_embeds, np.float32)[r], np.asarray(reid_embeds, np.float32)[c]
    diff = (a - b) + np.float32(1e-6)
    emb = np.sqrt((diff * diff).sum(axis=1, dtype=np.float32)).astype(np.float32)
    na = np.maximum(np.sqrt((a * a).sum(axis=1, dtype=np.float32)), np.float32(1e-8))
    nb = np.maximum(np.sqrt((b * b).sum(axis=1, dtype=np.float32)), np.float32(1e-8))
    cos = ((a / na[:, None]) * (b / nb[:, None])).sum(axis=1, dtype=np.float32).astype(np.float32)
    if only_appearance:
        attr = np.stack([emb, cos], axis=1)
    elif only_dist:
        attr = np.stack([l2, l1], axis=1)
    else:
        attr = np.stack([l2, l1, emb, cos], axis=1)
    labels = (np.asarray(ids)[r] == np.asarray(ids)[c]).astype(np.float32)
    return ei, attr.astype(np.float32), labels
