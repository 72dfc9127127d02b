"""TEST INFRASTRUCTURE (the checker, never the product path): CPU restatement of the step right after the MPN.

`threshold`            inference.py:286-291            sigmoid, >= 0.5
`prune`                libs/utils.py:387-404           remove_edges_single_direction: an active edge (i, j) survives iff (j, i) is active too
`flows`                libs/utils.py:54-55, 140-141    scatter_add of the predictions over the source / the target node
`clusters`             libs/utils.py:295-317           compute_SCC_and_Clusters on the active edges: strongly connected components, every
                                                       untouched node a cluster of its own (cluster NUMBERING is an artefact of set
                                                       iteration order there; the partition and the count are what is compared)
Pinned by tests/golden/post_*.npz, produced by the reference's own functions (tests/golden/make_golden_post.py)."""
import numpy as np


def threshold(logits):
    x = np.asarray(logits, dtype=np.float32).reshape(-1)
    probs = (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)
    return probs, (probs >= 0.5).astype(np.int64)


def prune(edge_index, predictions):
    ei = np.asarray(edge_index)
    pred = np.asarray(predictions).reshape(-1).astype(np.int64)
    n = int(ei.max()) + 1 if ei.size else 0
    key = ei[0].astype(np.int64) * n + ei[1]
    rev = ei[1].astype(np.int64) * n + ei[0]
    active = set(key[pred == 1].tolist())
    out = pred.copy()
    for k in np.nonzero(pred == 1)[0]:
        if int(rev[k]) not in active:
            out[k] = 0
    return out


def flows(edge_index, predictions, n_nodes):
    ei = np.asarray(edge_index)
    pred = np.asarray(predictions).reshape(-1).astype(np.int64)
    return np.bincount(ei[0], weights=pred, minlength=n_nodes).astype(np.int64), np.bincount(ei[1], weights=pred, minlength=n_nodes).astype(np.int64)


def clusters(edge_index, predictions, n_nodes):
    """-> (labels [n_nodes], number of clusters); labels are the smallest node id of the component."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    ei = np.asarray(edge_index)
    pred = np.asarray(predictions).reshape(-1).astype(bool)
    a = sp.coo_matrix((np.ones(int(pred.sum())), (ei[0][pred], ei[1][pred])), shape=(n_nodes, n_nodes)).tocsr()
    n_comp, lab = connected_components(a, directed=True, connection="strong")
    first = np.full(n_comp, n_nodes, dtype=np.int64)
    np.minimum.at(first, lab, np.arange(n_nodes))
    return first[lab], int(n_comp)


def same_partition(a, b):
    a, b = np.asarray(a), np.asarray(b)
    # label-independent: the first occurrence of every label, in order
    def canon(v):
        _, idx, inv = np.unique(v, return_index=True, return_inverse=True)
        return idx[inv]
    return np.array_equal(canon(a), canon(b))
