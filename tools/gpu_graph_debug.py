"""Diagnostic: a multi-pass vbs_track_to_3d captured into a graph and replayed several times; prints where replays differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vbs_amd.synth as S
from vbs_amd import _lib as L
from vbs_amd.engine import Engine
from vbs_amd.pipeline import reference_from_frame0

spec = S.config1()
n = 22
ft = S.make_frames_torch(spec, range(n), seed=5, device="cuda")
cam = L.make_camera(*S.default_camera(spec), 2.0)
for mode in sys.argv[1:] or ["lazy", "eager", "one"]:
    eng = Engine(spec.height, spec.width, max_markers=256, max_batch=4)
    ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    xy_d = torch.as_tensor(xy, dtype=torch.float64, device="cuda")
    if mode == "eager":
        eng.set_option(L.OPT_PASS_STREAMS, 2)
    if mode == "one":
        eng.set_option(L.OPT_PASS_STREAMS, 1)
    e2 = Engine(spec.height, spec.width, max_markers=256, max_batch=4)
    e2.set_option(L.OPT_PASS_STREAMS, 1)
    want, _, wc = e2.track_to_3d(ft, xy_d, 20.0, cam, 5.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            table, _, counts = eng.track_to_3d(ft, xy_d, 20.0, cam, 5.0)
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(4):
        table.zero_(); counts.zero_()
        g.replay()
        torch.cuda.synchronize()
        bad = (counts != wc).nonzero().flatten().tolist()
        badt = (table != want).any(dim=2).any(dim=1).nonzero().flatten().tolist()
        print(mode, "rep", rep, "counts differ at frames", bad, "tables differ at frames", badt, "counts", counts.tolist()[:8], flush=True)
    # eager calls after the replays still right?
    t2, _, c2 = eng.track_to_3d(ft, xy_d, 20.0, cam, 5.0)
    torch.cuda.synchronize()
    print(mode, "eager after replays equal:", bool(torch.equal(t2, want)), bool(torch.equal(c2, wc)), flush=True)
    del g
    eng.close(); e2.close()
