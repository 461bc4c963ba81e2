"""First-frame identity assignment (host side, once per video) — `MarkerTracker._process_first_frame`
(`marker_detection.py:275-347`; the same block is inlined at `tracking.py:106-178`).

The reference picks the marker nearest the mean as (0, 0), clusters the others' radii into `num_layers`
rings with an unseeded `KMeans`, and orders each ring by angle.  Two properties of the published code
matter for a drop-in and are selectable here:

 * id_mode="as_written": the placeholder key `(layer, -1)` (`:318-321`) is reused for every marker of a
   layer, so only the last one (in detection order) survives and becomes `(layer, 0)`.  This mode
   reproduces that exactly (1 + num_layers IDs).
 * id_mode="full": every marker keeps its own `(layer, angle_idx)` — the behaviour the docstring at
   `tracking.py:13-16` describes.  Dict order: (0,0), then layer-major, by ascending angle.

Clustering: kmeans="optimal" is a deterministic 1-D k-means (exact optimum over contiguous partitions
of the sorted radii); kmeans="sklearn" calls `sklearn.cluster.KMeans(n_clusters, n_init=10)` like the
reference (result may vary between runs on regular grids).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def kmeans_1d(values, k: int) -> np.ndarray:
    """Labels (0..k-1 by ascending centre) of the SSE-optimal partition of 1-D `values` into k clusters."""
    v = np.asarray(values, dtype=np.float64).ravel()
    n = v.size
    k = max(1, min(int(k), n))
    order = np.argsort(v, kind="stable")
    s = v[order]
    c1 = np.concatenate(([0.0], np.cumsum(s)))
    c2 = np.concatenate(([0.0], np.cumsum(s * s)))
    lo = np.arange(n + 1)[:, None]
    hi = np.arange(n + 1)[None, :]
    cnt = np.maximum(hi - lo, 1)
    sse = (c2[None, :] - c2[:, None]) - (c1[None, :] - c1[:, None]) ** 2 / cnt     # sse[i, j] of s[i:j]
    sse = np.where(hi > lo, sse, np.inf)
    cost = np.full((k + 1, n + 1), np.inf)
    back = np.zeros((k + 1, n + 1), dtype=np.int64)
    cost[0, 0] = 0.0
    for c in range(1, k + 1):
        cand = cost[c - 1][:, None] + sse            # [i, j]
        back[c] = np.argmin(cand, axis=0)            # first minimum = smallest split point
        cost[c] = cand[back[c], np.arange(n + 1)]
    lab_sorted = np.empty(n, dtype=np.int64)
    j = n
    for c in range(k, 0, -1):
        i = int(back[c, j])
        lab_sorted[i:j] = c - 1
        j = i
    labels = np.empty(n, dtype=np.int64)
    labels[order] = lab_sorted
    return labels


def assign_ids(markers: List[dict], num_layers: int = 5, id_mode: str = "as_written",
               kmeans: str = "optimal") -> Dict[Tuple[int, int], dict]:
    if not markers:
        raise ValueError("No markers detected in first frame!")
    if id_mode not in ("as_written", "full"):
        raise ValueError(f"id_mode must be 'as_written' or 'full', got {id_mode!r}")
    pts = np.array([m["center"] for m in markers], dtype=np.float64)
    ci = int(np.argmin(np.linalg.norm(pts - pts.mean(axis=0), axis=1)))
    centre = markers[ci]
    table: Dict[Tuple[int, int], dict] = {
        (0, 0): {**centre, "Ox": centre["center"][0], "Oy": centre["center"][1]}}
    rest = [m for i, m in enumerate(markers) if i != ci]
    if not rest:
        return table
    vec = np.array([m["center"] for m in rest], dtype=np.float64) - np.asarray(centre["center"])
    radius = np.linalg.norm(vec, axis=1)
    theta = np.arctan2(vec[:, 1], vec[:, 0])
    if kmeans == "sklearn":
        from sklearn.cluster import KMeans
        km = KMeans(n_clusters=num_layers, n_init=10).fit(radius.reshape(-1, 1))
        rank = np.empty(num_layers, dtype=np.int64)
        rank[np.argsort(km.cluster_centers_.ravel())] = np.arange(num_layers)
        layer = rank[km.labels_] + 1
    elif kmeans == "optimal":
        layer = kmeans_1d(radius, num_layers) + 1
    else:
        raise ValueError(f"kmeans must be 'optimal' or 'sklearn', got {kmeans!r}")

    def entry(i):
        m = rest[i]
        return {**m, "angle_rad": theta[i], "Ox": m["center"][0], "Oy": m["center"][1]}

    for lay in range(1, num_layers + 1):
        members = [i for i in range(len(rest)) if layer[i] == lay]
        if not members:
            continue
        if id_mode == "as_written":
            continue
        members.sort(key=lambda i: theta[i])
        start = int(np.argmin([abs(theta[i]) for i in members]))
        for pos, i in enumerate(members):
            table[(lay, (pos - start) % len(members))] = entry(i)
    if id_mode == "as_written":
        # the (layer, -1) slot is created at the layer's first appearance in detection order and
        # overwritten by every later member; the slots are then renamed (layer, 0) in layer order
        slot: Dict[int, int] = {}
        for i in range(len(rest)):
            slot[int(layer[i])] = i
        for lay in range(1, num_layers + 1):
            if lay in slot:
                table[(lay, 0)] = entry(slot[lay])
    return table


def reference_arrays(table: Dict[Tuple[int, int], dict]):
    """(ids int64 [M,2], ref_xy float64 [M,2]) in dict order — the layout the kernels consume."""
    ids = np.array(list(table.keys()), dtype=np.int64).reshape(-1, 2)
    xy = np.array([[v["Ox"], v["Oy"]] for v in table.values()], dtype=np.float64).reshape(-1, 2)
    return ids, xy
