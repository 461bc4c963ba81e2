"""Stage-by-stage GPU-vs-oracle diagnostic (prints, never asserts).  Run on the GPU box:
    python tests/gpu_diag.py [config]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine
from vbs_amd import _lib as L
from oracle import stages as O


def main():
    which = sys.argv[1:] or ["c1", "c2"]
    for tag in which:
        spec = {"c1": S.config1(), "c2": S.config2(), "c5": S.config5(), "ring": S.ring65_spec()}[tag]
        ch = 3 if tag in ("c1", "ring") else 1
        frames = S.make_frames(spec, [0, 1, 2], seed=1, channels=ch)
        print(f"=== {tag} {spec.name} frames {frames.shape}", flush=True)
        eng = Engine(spec.height, spec.width, max_markers=512, max_batch=2)
        ft = torch.from_numpy(frames).cuda()
        t0 = time.time()
        mask, area = eng.find_markers(ft)
        torch.cuda.synchronize()
        print("find_markers gpu s", time.time() - t0)
        ncc = eng.ncc_map(ft[:1]).cpu().numpy()[0]
        print("stats", eng.frame_stats(1))
        mask, area = mask.cpu().numpy(), area.cpu().numpy()
        for i in range(frames.shape[0]):
            t0 = time.time()
            om, oa = O.find_markers(frames[i])
            t1 = time.time() - t0
            print(f" frame {i}: oracle {t1:.2f}s area diff px {(oa != area[i]).sum()} (fg {int((oa>0).sum())})"
                  f" mask diff px {(om != mask[i]).sum()} (fg {int(om.sum())})", flush=True)
            if i == 0:
                p = O.branch_params(spec.height)
                gray = O.bgr2gray(frames[i])
                with np.errstate(all="ignore"):
                    ref = O.normxcorr2(O.gkern(p["tl"], p["tsig"]), oa)
                d = np.abs(ref - ncc)
                print("   ncc map max|diff|", d.max(), "median", np.median(d), "min|ref-0.1|", np.abs(ref - 0.1).min())
            if (oa != area[i]).any():
                yy, xx = np.nonzero(oa != area[i])
                print("   area mismatch rows", yy.min(), yy.max(), "cols", xx.min(), xx.max())
            # marker_center on the ORACLE masks (isolates the stage)
            mt, at = torch.from_numpy(om).cuda(), torch.from_numpy(oa).cuda()
            det, counts = eng.marker_center(mt, at)
            det, counts = det.cpu().numpy()[0], int(counts.cpu()[0])
            t0 = time.time()
            omk, dbg = O.marker_center(om, oa, True)
            print(f"   marker_center: gpu {counts} oracle {len(omk)} ({time.time()-t0:.2f}s)")
            k = min(counts, len(omk))
            if k:
                oc = np.array([[m["center"][0], m["center"][1], m["major_axis"], m["minor_axis"], m["angle"]]
                               for m in omk[:k]])
                dd = np.abs(det[:k, :5] - oc)
                print("   max|diff| x,y,major,minor,angle:", dd.max(0))
                bad = np.nonzero(dd[:, :2].max(1) > 1e-3)[0]
                if bad.size:
                    print("   first bad rows", bad[:5], det[bad[:3], :6], oc[bad[:3]])
        # fused path
        rows, ref = O.process_frames(list(frames), id_mode="full")
        ref_xy = np.array([[v["Ox"], v["Oy"]] for v in ref.values()])
        K, dist, R, T = S.default_camera(spec)
        cam = L.make_camera(K, dist, R, T, 2.0)
        table, det, counts = eng.track_to_3d(ft, ref_xy, 20.0, cam, 5.0, want_det=True)
        table = table.cpu().numpy()
        keys = list(ref.keys())
        nbad = 0
        worst = np.zeros(5)
        for r in rows:
            slot = keys.index((r["row"], r["col"]))
            trow = table[r["frameno"], slot]
            if int(trow[0]) & 1 == 0:
                nbad += 1
                continue
            o = np.array([r["Cx"], r["Cy"], r["major_axis"], r["minor_axis"], r["angle"]])
            worst = np.maximum(worst, np.abs(trow[1:6] - o))
        print(f" fused: oracle rows {len(rows)} gpu tracked {int((table[..., 0].astype(int) & 1).sum())} missing {nbad}"
              f" worst diffs {worst}")
        rows3 = O.track_markers_3d(rows, K, dist, R, T, warmup_frames=0)
        disp = eng.displacement(torch.from_numpy(table).cuda(), 0, 5.0, 50.0).cpu().numpy()
        w3 = np.zeros(7)
        miss = 0
        for r in rows3:
            slot = keys.index((r["row"], r["col"]))
            trow, drow = table[r["frameno"], slot], disp[r["frameno"], slot]
            if drow[0] != 1:
                miss += 1
                continue
            o = np.array([r["X"], r["Y"], r["Z"], r["dX"], r["dY"], r["dZ"], r["displacement"]])
            g = np.concatenate([trow[6:9], drow[1:5]])
            w3 = np.maximum(w3, np.abs(g - o))
        print(f" 3d: oracle rows {len(rows3)} gpu disp rows {int(disp[..., 0].sum())} missing {miss} worst {w3}")
        pl = eng.plane_fit(torch.from_numpy(table).cuda()).cpu().numpy()
        for f in range(frames.shape[0]):
            v = (table[f, :, 0].astype(int) & 2) > 0
            a, b, c, tilt = O.fit_plane(table[f, v, 6].astype(np.float64), table[f, v, 7].astype(np.float64),
                                        table[f, v, 8].astype(np.float64))
            print("  plane gpu", pl[f], "oracle", (a, b, c, tilt))
        eng.close()


if __name__ == "__main__":
    main()
