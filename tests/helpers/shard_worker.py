"""Child process of tests/test_gpu_parity.py::test_track_shard_two_ranks_on_one_gpu: one rank of a 2-rank gloo group,
both ranks on GPU 0.  argv: rank world port n_total outdir [workload c2|c5] [pipelined 1|0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def make_clip(n_total, workload="c2"):
    """`n_total` config-2 (or config-5: 1920x1200, 441 markers, the plane-fit workload) frames; the centre dot is painted
    out in frame n_total // 2, the first frame of rank 1, so the displacement of the following frame looks back ACROSS the
    shard edge."""
    import vbs_amd.synth as S
    spec = S.config5() if workload == "c5" else S.config2()
    frames = S.make_frames(spec, range(n_total), seed=11)
    cx, cy = spec.width // 2, spec.height // 2
    frames[n_total // 2, cy - 30:cy + 30, cx - 30:cx + 30] = 190
    return spec, frames


def main():
    rank, world, port, n_total, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    workload = sys.argv[6] if len(sys.argv) > 6 else "c2"
    pipelined = bool(int(sys.argv[7])) if len(sys.argv) > 7 else True
    import torch
    import torch.distributed as td
    import vbs_amd.synth as S
    from vbs_amd import _lib as L
    from vbs_amd import dist as D
    from vbs_amd.engine import Engine
    from vbs_amd.pipeline import track_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        spec, frames = make_clip(n_total, workload)
        a, b = D.shard_bounds(n_total, world, rank)
        K, dist, R, T = S.default_camera(spec)
        cam = L.make_camera(K, dist, R, T, 2.0)
        eng = Engine(spec.height, spec.width, max_markers=1024 if workload == "c5" else 512, max_batch=2 if workload == "c5" else 4, device=0)
        res = track_shard(eng, torch.from_numpy(frames[a:b]).cuda(), n_total, cam=cam, warmup_frames=0, pipelined=pipelined)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), table=res.table.cpu().numpy(), disp=res.disp.cpu().numpy(),
                 plane=res.plane.cpu().numpy(), ids=res.ids, xy=res.ref_xy, span=np.array([res.frame_begin, res.frame_end]),
                 counts=res.counts.cpu().numpy())
    finally:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
