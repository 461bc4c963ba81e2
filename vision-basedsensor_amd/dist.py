"""Multi-GPU sharding of the frame axis: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).

After frame 0 has fixed the reference table, frames are independent, so rank r takes the contiguous
block [r*N/G, (r+1)*N/G).  The only exchanges are a broadcast of the reference table (a few KB, once)
and ONE all-gather of the fixed-slot per-frame tables [N/G, M_ref, 10] float32; the last-seen
displacement scan (`3d_reconstruction.py:277-314`) then runs on the gathered table.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as td


def world():
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def shard_bounds(n_total: int, world_size: int, rank: int):
    """Contiguous block of rank `rank`; the first n_total % world_size ranks get one frame more."""
    base, extra = divmod(int(n_total), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_reference(ids, ref_xy, device, src: int = 0):
    """Broadcast (ids int64 [M,2], ref_xy float64 [M,2]) from `src`; other ranks pass None."""
    rank, ws = world()
    if ws == 1:
        return np.asarray(ids), np.asarray(ref_xy)
    if td.get_backend() == "gloo":
        device = torch.device("cpu")
    m = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        m[0] = len(ids)
    td.broadcast(m, src)
    M = int(m.item())
    buf = torch.zeros((M, 4), dtype=torch.float64, device=device)
    if rank == src:
        buf[:, :2] = torch.as_tensor(np.asarray(ids, dtype=np.float64), device=device)
        buf[:, 2:] = torch.as_tensor(np.asarray(ref_xy, dtype=np.float64), device=device)
    td.broadcast(buf, src)
    out = buf.cpu().numpy()
    return out[:, :2].astype(np.int64), out[:, 2:].copy()


def gather_tables(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather the per-rank tables [n_r, M, C] into [n_total, M, C] (frame order).  Shards may differ
    by one frame; they are padded to the largest shard so that a single collective is enough."""
    rank, ws = world()
    if ws == 1:
        return local
    per = [shard_bounds(n_total, ws, r) for r in range(ws)]
    nmax = max(b - a for a, b in per)
    pad = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    if td.get_backend() == "gloo" and local.is_cuda:
        # rehearsal backend (several ranks on one GPU / CPU tests): gloo gathers host tensors
        parts = [torch.empty(pad.shape, dtype=pad.dtype) for _ in range(ws)]
        td.all_gather(parts, pad.cpu().contiguous())
        out = torch.cat(parts, dim=0).to(local.device)
    else:
        out = torch.empty((ws * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        td.all_gather_into_tensor(out, pad.contiguous())
    if all(b - a == nmax for a, b in per):
        return out
    return torch.cat([out[r * nmax:r * nmax + (b - a)] for r, (a, b) in enumerate(per)], dim=0)


class TableGather:
    """All-gather of the per-frame tables issued PASS BY PASS, so that the exchange of one internal pass overlaps the
    kernels of the next one (RCCL runs on its own stream; `async_op=True` only orders it after the pass that produced
    the chunk).  Every pass is ONE `all_gather_into_tensor` into its own staging tensor [world x width, M, C] (each rank
    pads its rows of the pass to the widest rank's, so ragged shards take the same collective); `finish` waits and copies
    every rank's rows into its frames' rows of the final [n_total, M, C] tensor.  No tensor-list outputs: nothing relies on
    how a backend flattens and copies them.  NOTE: this branch has run under gloo (CPU tensors, 2 and 8 ranks, ragged
    shards) and, through host copies, with two ranks sharing one GPU; it has NOT yet run under RCCL on several GPUs (no
    multi-GPU box in this build): unverified there until the driver's scaling run, which checks itself (bench.py: the
    gathered table's checksum is compared across ranks; `--pipelined 0` falls back to the single collective).

        g = TableGather(n_total, M, C, device, chunk)
        for off in range(0, g.n_max, chunk):
            g.push(off, local_rows_of_this_pass)        # [<= chunk, M, C], may be empty on a short rank
        table = g.finish()
    """

    def __init__(self, n_total: int, m: int, cols: int, device, chunk: int, dtype=torch.float32):
        self.rank, self.ws = world()
        self.n_total, self.chunk = int(n_total), int(chunk)
        self.spans = [shard_bounds(n_total, self.ws, r) for r in range(self.ws)]
        self.n_max = max(b - a for a, b in self.spans)
        self.out = torch.empty((n_total, m, cols), dtype=dtype, device=device)
        self.pending = []                               # (work, staging tensor, source kept alive, off, rows per rank)
        self.host = self.ws > 1 and td.get_backend() == "gloo" and torch.device(device).type == "cuda"

    def push(self, off: int, local: torch.Tensor):
        a, b = self.spans[self.rank]
        if self.ws == 1:
            self.out[a + off:a + off + local.shape[0]] = local
            return
        # rows of this pass per rank (ranks with shorter shards have fewer / none in the last pass)
        cnt = [max(0, min(self.chunk, (rb - ra) - off)) for ra, rb in self.spans]
        width = max(cnt)
        if width == 0:
            return
        src = local
        if local.shape[0] != width:                     # ragged end: pad to the pass width
            src = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            src[:local.shape[0]] = local
        src = src.contiguous()
        if self.host:                                   # gloo rehearsal with device tensors: through the host, synchronous
            stage = torch.empty((self.ws * width,) + tuple(src.shape[1:]), dtype=src.dtype)
            td.all_gather_into_tensor(stage, src.cpu())
            self._scatter(stage.to(self.out.device), off, cnt)
            return
        self._reap()                                    # passes whose exchange has finished give their staging tensor back
        stage = torch.empty((self.ws * width,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)       # rank-major
        work = td.all_gather_into_tensor(stage, src, async_op=True)
        self.pending.append((work, stage, src, off, cnt))

    def _reap(self):
        """Copy the rows of every pass whose collective has completed into place and drop its staging tensor, so that the
        staging memory in flight stays at the few passes the exchange lags behind, not a second copy of the whole table."""
        keep = []
        for item in self.pending:
            work, stage, _src, off, cnt = item
            if work.is_completed():
                work.wait()                             # (orders the copy below behind the collective on this stream)
                self._scatter(stage, off, cnt)
            else:
                keep.append(item)
        self.pending = keep

    def _scatter(self, stage, off, cnt):
        width = stage.shape[0] // self.ws
        for r, (ra, _) in enumerate(self.spans):
            if cnt[r]:
                self.out[ra + off:ra + off + cnt[r]] = stage[r * width:r * width + cnt[r]]

    def finish(self) -> torch.Tensor:
        for work, stage, _src, off, cnt in self.pending:
            work.wait()
            self._scatter(stage, off, cnt)
        self.pending = []
        return self.out
