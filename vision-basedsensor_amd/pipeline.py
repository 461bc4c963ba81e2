"""Batch driver: frames resident on the device -> fixed-slot tables -> (all-gather) -> displacement
-> plane-fit pose.  This is what `bench.py` times and what the multi-GPU path runs per rank."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from . import dist as D
from . import ids as _ids
from .engine import Engine
from .marker_detection import _det_to_markers


@dataclass
class TrackResult:
    ids: np.ndarray            # [M,2] (row, col) per slot
    ref_xy: np.ndarray         # [M,2]
    table: torch.Tensor        # [N,M,10] float32 (gathered when distributed)
    disp: Optional[torch.Tensor]     # [n_local,M,5]  this rank's frames [frame_begin, frame_end)
    plane: Optional[torch.Tensor]    # [n_local,5]
    frame_begin: int = 0
    frame_end: int = 0
    counts: Optional[torch.Tensor] = None   # [n_local] int32 detections per frame of this rank (never negative here)


ID_CHECK = {}                  # what the last reference_from_frame0 found (bench.py reports it)


def reference_from_frame0(eng: Engine, frame0: torch.Tensor, num_layers=5, id_mode="full", kmeans="optimal",
                          ids_on_device=True):
    """Detect frame 0 on the device and assign the identities (once per video, `marker_detection.py:275-347`).
    `ids_on_device` (default) runs the assignment on the GPU (`vbs_assign_ids`, deterministic clustering only) and the host
    restatement (`ids.assign_ids`, pinned to the reference body's golden) on the same detections as its checker:
      * the two tables agree bit for bit (every benchmark frame 0, the reference's real frame) -> the DEVICE table is what is
        returned and used (`ID_CHECK["used"] == "device"`);
      * they hold the same IDs but order markers at mathematically equal angles differently (collinear with the centre:
        np.arctan2's last bit decides on the host, the device's atan2 on the GPU) -> the host table, the reference's own
        order (`"host"`, `slots_in_another_order` says how many slots);
      * anything else raises.
    The device kernel covers `num_layers <= 16` (k_ids.hip: IDS_MAXK) and `kmeans == "optimal"`; other configurations run
    on the host alone (`"host"`, `on_device: False`) as they always did."""
    _, det, counts = eng.track_to_3d(frame0[:1], None, want_det=True)
    dev = None
    if ids_on_device and kmeans == "optimal" and int(num_layers) <= 16:
        ids_d, xy_d = eng.assign_ids(det, counts, num_layers, id_mode)      # raises the reference's ValueError on no markers
        dev = (ids_d.cpu().numpy().astype("int64"), xy_d.cpu().numpy())
    n0 = int(counts[0].item())
    if n0 < 0:
        raise L.VbsError(f"device status {n0} in frame 0: {L.status_text(n0, eng.max_markers)}")
    table = _ids.assign_ids(_det_to_markers(det[0].cpu().numpy(), n0), num_layers, id_mode, kmeans)
    ids, xy = _ids.reference_arrays(table)
    ID_CHECK.clear()
    ID_CHECK.update({"on_device": dev is not None, "used": "host"})
    if dev is not None:
        same_ids = dev[0].shape == np.asarray(ids).shape and bool(np.array_equal(dev[0], ids))
        if not same_ids:
            raise L.VbsError("vbs_assign_ids disagrees with the host assignment on the marker IDs (not only on the order of "
                             "markers at equal angles): the device path must not be trusted on this frame 0")
        differ = int((dev[1] != np.asarray(xy)).any(axis=1).sum())
        # a slot may only differ by holding ANOTHER marker of the same layer (a swap among equal angles)
        if differ and sorted(map(tuple, dev[1].tolist())) != sorted(map(tuple, np.asarray(xy).tolist())):
            raise L.VbsError("vbs_assign_ids returned reference coordinates the host assignment does not have")
        ID_CHECK.update({"equal_to_host": differ == 0, "slots_in_another_order": differ, "markers": int(len(ids))})
        if differ == 0:
            ID_CHECK["used"] = "device"
            return dev[0], dev[1]
    return ids, xy


def track_and_gather(eng: Engine, frames_local: torch.Tensor, n_total: int, xy, min_dist=20.0, cam=None,
                     min_marker_size_px=5.0, pipelined=True):
    """This rank's frames through the fused path, `eng.pass_streams` internal passes (`eng.max_batch` frames each) at a time,
    the all-gather of their rows issued as soon as they are enqueued (`dist.TableGather`): the exchange overlaps the next ones.
    `pipelined=False`: all passes first, then the single `dist.gather_tables` collective (SURVEY 8e as written).
    Returns (local table [n_local, M, 10], counts [n_local], gathered table [n_total, M, 10])."""
    rank, ws = D.world()
    m = int(np.asarray(xy).reshape(-1, 2).shape[0])
    n_local = int(frames_local.shape[0])
    if ws == 1 or not pipelined:
        local, _, counts = eng.track_to_3d(frames_local, xy, min_dist, cam, min_marker_size_px)
        return local, counts, (local if ws == 1 else D.gather_tables(local, n_total))
    # one call and one collective per `pass_streams` internal passes: with two, the library runs the second pass of a call
    # on its second workspace and stream (VBS_OPT_PASS_STREAMS), as it does for a single process
    chunk = eng.max_batch * max(1, int(getattr(eng, "pass_streams", 1)))
    g = D.TableGather(n_total, m, L.TABLE_COLS, eng.device, chunk)
    a, _ = D.shard_bounds(n_total, ws, rank)
    counts = torch.zeros((n_local,), dtype=torch.int32, device=eng.device)
    for off in range(0, g.n_max, chunk):
        part = frames_local[off:off + chunk]
        if part.shape[0]:
            t, _, c = eng.track_to_3d(part, xy, min_dist, cam, min_marker_size_px)
            counts[off:off + part.shape[0]] = c
        else:                                           # a shorter shard has run out of frames: still joins the collective
            t = torch.zeros((0, m, L.TABLE_COLS), dtype=torch.float32, device=eng.device)
        g.push(off, t)
    table = g.finish()
    return table[a:a + n_local], counts, table


def track_shard(eng: Engine, frames_local: torch.Tensor, n_total: int, ref=None, cam: L.Camera = None,
                min_dist=20.0, min_marker_size_px=5.0, warmup_frames=0, max_displacement=50.0,
                num_layers=5, id_mode="full", kmeans="optimal", with_plane=True, pipelined=True) -> TrackResult:
    """One rank's part of a sequence of `n_total` frames (`frames_local` = this rank's contiguous block).
    `ref` = (ids, ref_xy) if already known; otherwise the rank holding frame 0 computes and broadcasts it."""
    rank, ws = D.world()
    if ref is None:
        ids = xy = None
        if rank == 0:
            ids, xy = reference_from_frame0(eng, frames_local, num_layers, id_mode, kmeans)
        ids, xy = D.broadcast_reference(ids, xy, eng.device)
    else:
        ids, xy = ref
    local, counts, table = track_and_gather(eng, frames_local, n_total, xy, min_dist, cam, min_marker_size_px, pipelined)
    bad = torch.nonzero(counts < 0)            # (after the collective, so a failing rank cannot leave the others waiting in it)
    if bad.numel():
        f = int(bad[0].item())
        raise L.VbsError(f"device status {int(counts[f].item())} in frame {D.shard_bounds(n_total, ws, rank)[0] + f} "
                         f"({L.status_text(int(counts[f].item()), eng.max_markers)}): the reference would have emitted rows "
                         f"for it")
    a, b = D.shard_bounds(n_total, ws, rank)
    disp = plane = None
    if cam is not None:
        # every rank scans only its own frames of the gathered table (the look-back may cross into the previous
        # rank's block), so the per-rank work does not grow with the number of GPUs
        disp = eng.displacement(table, warmup_frames, min_marker_size_px, max_displacement, frame_range=(a, b))
        if with_plane:
            plane = eng.plane_fit(local)
    return TrackResult(ids, xy, table, disp, plane, a, b, counts)


def deviation_pose(eng: Engine, table_vert: torch.Tensor, table_tilt: torch.Tensor, ref_xyz, mode="plane", scale=1.0,
                   start=0, end=-1):
    """Pose misalignment from two tracked loadings (`ForceDistribution.py:168-208,218-243`): the displacement of every
    marker between frames `start` and `end` of the vertical session's table and of the tilted session's, their difference
    (the deviation field), the plane through reference position + scale * deviation and its tilt.  Both tables must use
    the same slot order (the same frame-0 identities).  Returns a dict with `deviation` [M,4] (device tensor: common,
    dX, dY, dZ), `n`, `a`, `b`, `c`, `tilt_deg`, `mean_vector` (scaled, as the reference plots it) and `mean_magnitude`."""
    dev, out = eng.deviation_plane(table_vert[start], table_vert[end], table_tilt[start], table_tilt[end], ref_xyz, mode, scale)
    o = out.cpu().numpy().astype(np.float64)
    return {"deviation": dev, "n": int(o[0]), "a": o[1], "b": o[2], "c": o[3], "tilt_deg": o[4],
            "mean_vector": o[5:8].copy(), "mean_magnitude": o[8]}
