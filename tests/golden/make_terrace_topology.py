#!/usr/bin/env python3
"""Per-frame graph topology of the reference's own evaluation sequence, as NUMBERS only -> tests/golden/terrace_topology.npz.

Reads datasets/EPFL-Terrace/terrace1-c{0..3}/gt/gt.txt (the annotation files shipped with the reference) the
way libs/datasets.py does (`lost == 0`, libs/datasets.py:80; a frame is valid when detections come from more than one camera and at
least one identity is seen by two cameras, libs/datasets.py:226-232) and stores, per valid frame, the detections in the order
`EPFL_dataset` concatenates them (camera by camera, file order inside a camera): camera id and person id of every detection and the
frame boundaries.  No image, no annotation text, no code of the reference is stored; positions and embeddings of the bench workload
are synthetic (images and ReID weights are not in the repository).

Run in the build container only:   python tests/golden/make_terrace_topology.py
"""
import os

import numpy as np

REF = "/root/reference/datasets/EPFL-Terrace"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    cams = sorted(d for d in os.listdir(REF) if os.path.isdir(os.path.join(REF, d)))
    rows = []   # (frame, cam, id) in the dataset's concatenation order: camera by camera
    for c in cams:
        a = np.loadtxt(os.path.join(REF, c, "gt", "gt.txt"), usecols=(0, 5, 6), dtype=np.int64)   # id, frame, lost (COL_NAMES_EPFL)
        a = a[a[:, 2] == 0]
        cam = int(c[-1:])
        rows.append(np.stack([a[:, 1], np.full(len(a), cam), a[:, 0]], axis=1))
    det = np.concatenate(rows)
    frames = np.arange(det[:, 0].min(), det[:, 0].max() + 1)
    cam_l, id_l, ptr = [], [], [0]
    valid = []
    order = np.argsort(det[:, 0], kind="stable")   # stable: keeps camera-by-camera order inside a frame (data_det[frame == f])
    det = det[order]
    starts = np.searchsorted(det[:, 0], frames, side="left")
    ends = np.searchsorted(det[:, 0], frames, side="right")
    for f, s, e in zip(frames, starts, ends):
        if e - s == 0:
            continue
        d = det[s:e]
        if len(np.unique(d[:, 1])) > 1 and np.max(np.bincount(d[:, 2])) > 1:
            valid.append(f)
            cam_l.append(d[:, 1])
            id_l.append(d[:, 2])
            ptr.append(ptr[-1] + (e - s))
    cam = np.concatenate(cam_l).astype(np.uint8)
    pid = np.concatenate(id_l).astype(np.uint8)
    ptr = np.asarray(ptr, dtype=np.int32)
    sizes = np.diff(ptr)
    edges = np.array([int(((cam[a:b][:, None] != cam[a:b][None, :]).sum())) for a, b in zip(ptr[:-1], ptr[1:])])
    np.savez_compressed(os.path.join(HERE, "terrace_topology.npz"), frame=np.asarray(valid, dtype=np.int32), node_ptr=ptr, cam=cam, person=pid,
                        edges=edges.astype(np.int32))
    print(f"{len(valid)} valid frames; nodes/frame mean {sizes.mean():.1f} max {sizes.max()}; edges/frame mean {edges.mean():.0f} max {edges.max()}")


if __name__ == "__main__":
    main()
