"""CPU tests of the host side of the product: ID assignment against the oracle and the reference
goldens, the drop-in classes' configuration / error behaviour, file formats, frame sharding with a
2-process gloo group, and the rule that compute never silently falls back to the CPU."""
import json
import os
import sys

import numpy as np
import pytest

import vbs_amd.synth as S
from vbs_amd import ids as I
from oracle import stages as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _markers(spec, seed=5, frame=0, shuffle=1):
    truth = S.dot_truth(spec, seed, [frame])[0]
    order = np.random.default_rng(shuffle).permutation(spec.n_markers)
    return [{"center": (float(truth[k, 0]) - 0.25, float(truth[k, 1]) - 0.25), "major_axis": float(truth[k, 2]),
             "minor_axis": float(truth[k, 2]) - 0.5, "angle": 90.0} for k in order]


@pytest.mark.parametrize("spec", [S.ring65_spec(), S.config1(), S.config2(), S.config5()], ids=lambda s: s.name)
@pytest.mark.parametrize("id_mode", ["as_written", "full"])
def test_assign_ids_equals_oracle(spec, id_mode):
    ms = _markers(spec)
    got = I.assign_ids(ms, 5, id_mode, "optimal")
    want = O.process_first_frame(ms, 5, id_mode, "optimal")
    assert list(got.keys()) == list(want.keys())
    for k in got:
        assert got[k]["Ox"] == want[k]["Ox"] and got[k]["Oy"] == want[k]["Oy"]
        if k != (0, 0):
            assert got[k]["angle_rad"] == want[k]["angle_rad"]
    assert len(got) == (6 if id_mode == "as_written" else spec.n_markers)
    ids, xy = I.reference_arrays(got)
    assert ids.shape == (len(got), 2) and xy.shape == (len(got), 2) and ids.dtype == np.int64


@pytest.mark.parametrize("name", ["ring65", "grid7"])
def test_assign_ids_reference_golden(golden_dir, name):
    """as_written IDs bit-exact against the output of the reference's own `_process_first_frame`."""
    g = json.load(open(os.path.join(golden_dir, "ids_as_written.json")))[name]
    fr0 = [{**m, "center": tuple(m["center"])} for m in g["frames"][0]]
    for km in ("optimal", "sklearn"):
        got = I.assign_ids(fr0, g["num_layers"], "as_written", km)
        assert [[k[0], k[1], v["Ox"], v["Oy"]] for k, v in got.items()] == g["ref"]


def test_kmeans_1d_is_optimal_and_deterministic():
    rng = np.random.default_rng(0)
    for _ in range(20):
        v = rng.normal(size=rng.integers(6, 40)) * rng.uniform(0.1, 5) + rng.choice([0, 3, 9], size=1)
        k = int(rng.integers(1, 6))
        a = I.kmeans_1d(v, k)
        b = O.kmeans_1d_optimal(v, k)
        assert np.array_equal(a, b)
        assert np.array_equal(a, I.kmeans_1d(v.copy(), k))
    with pytest.raises(ValueError):
        I.assign_ids([], 5)
    with pytest.raises(ValueError):
        I.assign_ids(_markers(S.config1()), 5, "nope")


def test_marker_tracker_config_errors(tmp_path):
    from vbs_amd.marker_detection import MarkerTracker, find_marker, marker_center, _crop_box
    with pytest.raises(ValueError, match="Missing required config key: crop_ratios"):
        MarkerTracker({"video_path": "x", "output_dir": str(tmp_path)})
    with pytest.raises(FileNotFoundError):
        MarkerTracker({"video_path": str(tmp_path / "nope.avi"), "output_dir": str(tmp_path), "crop_ratios": (0, 0, 0, 0)})
    p = tmp_path / "clip.avi"
    p.write_bytes(b"not a video")
    t = MarkerTracker({"video_path": str(p), "output_dir": str(tmp_path / "o"), "crop_ratios": (1 / 8, 1 / 8, 1 / 16, 0)})
    assert t.output_csv.endswith("clip_markers.csv") and os.path.isdir(tmp_path / "o")
    with pytest.raises(IOError):
        t._init_video()                       # no OpenCV here and not a .npy clip
    assert find_marker is MarkerTracker._find_markers and marker_center is MarkerTracker._marker_center
    assert _crop_box(1280, 1024, (1 / 8, 1 / 8, 1 / 16, 0)) == O.crop_box(1280, 1024, (1 / 8, 1 / 8, 1 / 16, 0)) == (160, 1120, 64, 1024)
    assert _crop_box(640, 480, (1 / 8, 1 / 8, 1 / 16, 0)) == (80, 560, 30, 480)
    np.testing.assert_array_equal(MarkerTracker._gkern(33, 7.4), O.gkern(33, 7.4))
    from vbs_amd.tracking import process_video
    with pytest.raises(FileNotFoundError):
        process_video(video_dir=str(tmp_path / "v"))


def test_avi_reader_and_writer(tmp_path):
    """f4 front end: the package's AVI reader (the `cv2.VideoCapture` subset `_init_video` uses) on files written by its
    own writer - uncompressed frames come back bit for bit, Motion-JPEG frames equal Pillow's decode of the same
    chunks (BGR order), headers carry size / rate / count; anything else is 'not opened'."""
    import io
    from PIL import Image
    from vbs_amd.video_io import AviReader, write_avi, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT, CAP_PROP_FRAME_WIDTH, \
        CAP_PROP_FRAME_HEIGHT
    spec = S.config1()
    bgr = S.make_frames(spec, [0, 1, 2], seed=1, channels=3)
    bgr[..., 0] //= 2                                             # B != R, so a channel swap would show
    gray = S.make_frames(spec, [0, 1], seed=1)
    for frames, codec in ((bgr, "RAW"), (gray, "RAW"), (bgr, "MJPG"), (gray, "MJPG")):
        p = str(tmp_path / f"{codec}_{frames.ndim}.avi")
        write_avi(p, frames, fps=25.0, codec=codec)
        cap = AviReader(p)
        assert cap.isOpened()
        assert (cap.get(CAP_PROP_FRAME_WIDTH), cap.get(CAP_PROP_FRAME_HEIGHT)) == (spec.width, spec.height)
        assert cap.get(CAP_PROP_FPS) == 25.0 and cap.get(CAP_PROP_FRAME_COUNT) == len(frames)
        got = []
        while True:
            ok, f = cap.read()
            if not ok:
                break
            assert f.dtype == np.uint8 and f.shape == (spec.height, spec.width, 3)
            got.append(f)
        assert len(got) == len(frames) and cap.read() == (False, None)
        want3 = frames if frames.ndim == 4 else np.repeat(frames[..., None], 3, axis=3)
        if codec == "RAW":
            assert np.array_equal(np.stack(got), want3)
        else:
            err = np.abs(np.stack(got).astype(int) - want3.astype(int))                   # JPEG quality 95, 4:2:0 chroma
            assert err.mean() < 2.0 and err.max() <= 64
            raw = open(p, "rb").read()
            k = raw.index(b"00dc")
            size = int.from_bytes(raw[k + 4:k + 8], "little")
            ref = np.asarray(Image.open(io.BytesIO(raw[k + 8:k + 8 + size])).convert("RGB"))[:, :, ::-1]
            assert np.array_equal(got[0], ref)
        cap.release()
        assert not cap.isOpened()
        # the threaded batch decode (`read_batch`: what `MarkerTracker.process` uses to decode ahead of the GPU) gives the
        # frames `read` gave, into a caller's buffer or its own, in batches that do not divide the clip
        cap = AviReader(p)
        buf = np.zeros((2, spec.height, spec.width, 3), np.uint8)
        seen = []
        while True:
            m = cap.read_batch(2, buf, threads=2)
            if not m:
                break
            seen.extend(buf[:m].copy())
        assert len(seen) == len(got) and all(np.array_equal(a, b) for a, b in zip(seen, got))
        assert cap.read_batch(2, buf) == 0
        cap.release()
        cap = AviReader(p)
        m, own = cap.read_batch(100, threads=1)
        assert m == len(got) and np.array_equal(own, np.stack(got))
        cap.release()
    junk = tmp_path / "junk.avi"
    junk.write_bytes(b"RIFF\x10\x00\x00\x00AVI junkjunk")
    assert not AviReader(str(junk)).isOpened() and not AviReader(str(tmp_path / "missing.avi")).isOpened()


def test_no_silent_cpu_fallback():
    """Without a GPU every compute entry point must raise — never answer from a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vbs_amd._lib import VbsError
    from vbs_amd.engine import Engine, undistort_points, calculate_3d
    from vbs_amd.marker_detection import MarkerTracker
    with pytest.raises(VbsError):
        Engine(480, 640)
    with pytest.raises(VbsError):
        MarkerTracker._find_markers(np.zeros((480, 640, 3), np.uint8))
    with pytest.raises(VbsError):
        MarkerTracker._marker_center(np.zeros((480, 640), np.uint8), np.zeros((480, 640), np.uint8))
    from vbs_amd import _lib as L
    cam = L.make_camera(np.eye(3), np.zeros(5), np.eye(3), np.zeros(3))
    with pytest.raises(VbsError):
        undistort_points(np.zeros((1, 2)), cam)
    with pytest.raises(VbsError):
        calculate_3d(np.zeros((1, 3)), cam)
    # and the product never imports the oracle
    for mod in ("marker_detection", "tracking", "reconstruction3d", "engine", "pipeline", "ids", "dist", "synth", "_lib"):
        src = open(os.path.join(ROOT, "vision-basedsensor_amd", mod + ".py")).read()
        assert "from oracle" not in src and "import oracle" not in src


def test_marker_analysis_files(tmp_path):
    import pandas as pd
    from vbs_amd.reconstruction3d import MarkerAnalysis, Config, CONFIG
    assert CONFIG.marker_diameter_mm == 2.0 and CONFIG.warmup_frames == 100 and CONFIG.min_marker_size_px == 5.0 \
        and CONFIG.max_displacement_px == 50.0 and CONFIG.column_mapping == {"Cx": "u", "Cy": "v", "major_axis": "major_axis"}
    cfg = Config(data_dir=tmp_path / "d", output_dir=tmp_path / "d" / "r", plots_dir=tmp_path / "d" / "r" / "p")
    ma = MarkerAnalysis(cfg)
    assert (tmp_path / "d" / "r" / "p").is_dir()
    intr = pd.DataFrame({"Param": ["fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3"],
                         "Value": [1400.5, 1399.25, 640.0, 512.0, -0.1, 0.01, 0.001, -0.002, 0.0], "Desc": [""] * 9})
    intr.to_csv(tmp_path / "intr.csv", index=False)
    R = np.eye(3)
    ext = {f"R_wc_{i + 1}{j + 1}": R[i, j] for i in range(3) for j in range(3)}
    ext.update({"T_wc_X": 1.0, "T_wc_Y": -2.0, "T_wc_Z": 30.0})
    (tmp_path / "ext.json").write_text(json.dumps(ext))
    ma.load_parameters(tmp_path / "intr.csv", tmp_path / "ext.json")
    assert ma.camera.matrix.dtype == np.float32 and ma.camera.matrix[0, 0] == np.float32(1400.5)
    assert ma.camera.dist_coeffs.dtype == np.float32 and ma.camera.dist_coeffs.shape == (5,)
    assert ma.camera.T_world_to_cam.shape == (3, 1) and ma.camera.T_world_to_cam[2, 0] == 30.0
    ext["R_wc_12"] = 0.1
    (tmp_path / "bad.json").write_text(json.dumps(ext))
    with pytest.raises(ValueError, match="not orthogonal"):
        ma.load_parameters(tmp_path / "intr.csv", tmp_path / "bad.json")
    intr.loc[0, "Value"] = -1.0
    intr.to_csv(tmp_path / "neg.csv", index=False)
    with pytest.raises(ValueError, match="Focal lengths"):
        ma.load_parameters(tmp_path / "neg.csv", tmp_path / "ext.json")
    # marker CSV: required columns, size filter, sort
    with pytest.raises(FileNotFoundError):
        ma.load_marker_data(tmp_path / "none.csv")
    pd.DataFrame({"frameno": [2, 1, 1], "row": [0, 0, 1], "col": [0, 0, 0], "Cx": [1.0, 2.0, 3.0],
                  "Cy": [4.0, 5.0, 6.0], "major_axis": [10.0, 4.0, 12.0]}).to_csv(tmp_path / "m.csv", index=False)
    df = ma.load_marker_data(tmp_path / "m.csv")
    assert list(df["frameno"]) == [1, 2] and {"u", "v"} <= set(df.columns)
    pd.DataFrame({"frameno": [1], "row": [0]}).to_csv(tmp_path / "bad.csv", index=False)
    with pytest.raises(ValueError, match="Missing required columns"):
        ma.load_marker_data(tmp_path / "bad.csv")


def _openpyxl_style_workbook(path, header, rows):
    """An .xlsx laid out the way pandas/openpyxl writes one (shared strings, t="s" cells, a styles part, a docProps
    part): built by hand so the reader is tested on that layout without an Excel engine."""
    import zipfile
    from xml.sax.saxutils import escape
    strings, sidx = [], {}

    def sref(v):
        if v not in sidx:
            sidx[v] = len(strings)
            strings.append(v)
        return sidx[v]

    def col(i):
        return chr(65 + i)

    xml_rows = []
    for r, row in enumerate([header] + rows, start=1):
        cells = []
        for c, v in enumerate(row):
            if v is None or v == "":
                continue
            if isinstance(v, str):
                cells.append(f'<c r="{col(c)}{r}" s="1" t="s"><v>{sref(v)}</v></c>')
            else:
                cells.append(f'<c r="{col(c)}{r}"><v>{v!r}</v></c>')
        xml_rows.append(f'<row r="{r}" spans="1:3">{"".join(cells)}</row>')
    ns = "http://schemas.openxmlformats.org/spreadsheetml/2006/main"
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("[Content_Types].xml", '<?xml version="1.0"?><Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types"/>')
        z.writestr("docProps/app.xml", "<Properties/>")
        z.writestr("xl/workbook.xml", f'<?xml version="1.0"?><workbook xmlns="{ns}" xmlns:r="http://schemas.openxmlformats.org/'
                   'officeDocument/2006/relationships"><sheets><sheet name="Sheet1" sheetId="1" r:id="rId7"/></sheets></workbook>')
        z.writestr("xl/_rels/workbook.xml.rels", '<?xml version="1.0"?><Relationships xmlns="http://schemas.openxmlformats.org/package/'
                   '2006/relationships"><Relationship Id="rId3" Type="x/styles" Target="styles.xml"/>'
                   '<Relationship Id="rId7" Type="x/worksheet" Target="/xl/worksheets/sheet1.xml"/></Relationships>')
        z.writestr("xl/styles.xml", f'<styleSheet xmlns="{ns}"/>')
        z.writestr("xl/sharedStrings.xml", f'<?xml version="1.0"?><sst xmlns="{ns}" count="{len(strings)}">' + "".join(
            f"<si><t>{escape(t)}</t></si>" if i % 2 == 0 else f"<si><r><t>{escape(t[:1])}</t></r><r><t>{escape(t[1:])}</t></r></si>"
            for i, t in enumerate(strings)) + "</sst>")
        z.writestr("xl/worksheets/sheet1.xml", f'<?xml version="1.0"?><worksheet xmlns="{ns}"><sheetData>' + "".join(xml_rows) +
                   "</sheetData></worksheet>")


def test_xlsx_parameter_sheets_and_result_sheet(tmp_path):
    """f2: the two parameter workbooks as the reference's calibration scripts lay them out (`intrinsic_calibration.py:33-51`
    key column `Param`; `extrinsic_calibration.py:135-151` five title rows, keys `T_wc_X/Y/Z`) and as `load_parameters`
    itself expects them (`Parameter`, `Tx_wc` ... `3d_reconstruction.py:84,121`), read without an Excel engine; the
    result sheet (`:296-307,430-433`) written and read back bit-exactly."""
    import pandas as pd
    from vbs_amd import xlsx_io as X
    from vbs_amd.reconstruction3d import MarkerAnalysis, Config
    cfg = Config(data_dir=tmp_path / "d", output_dir=tmp_path / "d" / "r", plots_dir=tmp_path / "d" / "r" / "p")
    ma = MarkerAnalysis(cfg)
    c, s_ = np.cos(0.3), np.sin(0.3)
    R = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    intr_rows = [["fx", 1400.123456789, "Focal length x"], ["fy", 1399.5, "Focal length y"], ["cx", 640.25, ""],
                 ["cy", 512.75, ""], ["skew", 0.0, ""], ["k1", -0.1234567, ""], ["k2", 0.0123, ""], ["p1", 1e-4, ""],
                 ["p2", -2e-4, ""], ["k3", 0.0, ""], ["Reproj Error", 0.21, "Mean error (px)"]]
    ext_rows = [["--- Camera Extrinsic Parameters ---", "", ""], ["Calibration Date", "2025-01-01 10:00:00", ""],
                ["Reprojection Error (px)", 0.4, ""], ["", "", ""], ["--- World to Camera Transformation ---", "", ""]]
    ext_rows += [[f"R_wc_{i + 1}{j + 1}", float(R[i, j]), f"Rotation matrix element ({i + 1},{j + 1})"]
                 for i in range(3) for j in range(3)]
    # (a) as the calibration scripts write them, through the package's writer
    X.write_xlsx(tmp_path / "IntrinsicParameters.xlsx", ["Param", "Value", "Desc"], intr_rows)
    X.write_xlsx(tmp_path / "ExtrinsicParameters.xlsx", ["Parameter", "Value", "Description"],
                 ext_rows + [[f"T_wc_{a}", v, ""] for a, v in zip("XYZ", (1.5, -2.25, 30.125))])
    ma.load_parameters(tmp_path / "IntrinsicParameters.xlsx", tmp_path / "ExtrinsicParameters.xlsx")
    K = ma.camera.matrix.copy()
    assert K.dtype == np.float32 and K[0, 0] == np.float32(1400.123456789) and K[1, 2] == np.float32(512.75)
    assert np.array_equal(ma.camera.dist_coeffs, np.array([-0.1234567, 0.0123, 1e-4, -2e-4, 0.0], np.float32))
    assert np.array_equal(ma.camera.R_world_to_cam, R.astype(np.float32))
    assert ma.camera.T_world_to_cam.ravel().tolist() == [1.5, -2.25, 30.125]
    # (b) the spellings `load_parameters` itself names, in an openpyxl-style workbook (shared strings, rich text)
    _openpyxl_style_workbook(tmp_path / "i2.xlsx", ["Parameter", "Value", "Desc"], intr_rows)
    _openpyxl_style_workbook(tmp_path / "e2.xlsx", ["Parameter", "Value", "Description"],
                             ext_rows + [[f"T{a}_wc", v, ""] for a, v in zip("xyz", (1.5, -2.25, 30.125))])
    df = X.read_xlsx(tmp_path / "e2.xlsx")
    assert list(df.columns) == ["Parameter", "Value", "Description"] and len(df) == len(ext_rows) + 3
    assert df["Parameter"][0] == "--- Camera Extrinsic Parameters ---" and df["Parameter"][5] == "R_wc_11"
    assert df["Value"][3] is None or pd.isna(df["Value"][3])          # the blank row survives as a row
    mb = MarkerAnalysis(cfg)
    mb.load_parameters(tmp_path / "i2.xlsx", tmp_path / "e2.xlsx")
    assert np.array_equal(mb.camera.matrix, K) and np.array_equal(mb.camera.T_world_to_cam, ma.camera.T_world_to_cam)
    assert np.array_equal(mb.camera.R_world_to_cam, ma.camera.R_world_to_cam)
    # errors keep the reference's types
    X.write_xlsx(tmp_path / "nokey.xlsx", ["A", "B"], [["fx", 1.0]])
    with pytest.raises(ValueError):
        ma.load_parameters(tmp_path / "nokey.xlsx", tmp_path / "e2.xlsx")
    with pytest.raises(FileNotFoundError):
        ma.load_parameters(tmp_path / "absent.xlsx", tmp_path / "e2.xlsx")
    (tmp_path / "junk.xlsx").write_bytes(b"not a zip")
    with pytest.raises(ValueError):
        X.read_xlsx(tmp_path / "junk.xlsx")
    # (c) result sheet: float64 values round-trip exactly, integer columns stay integers
    rng = np.random.default_rng(3)
    res = pd.DataFrame({"frameno": np.arange(101, 107), "row": [0, 1, 1, 2, 2, 2], "col": [0, 0, 1, 0, 1, 2]})
    for ccol in ("X", "Y", "Z", "dX", "dY", "dZ", "displacement"):
        res[ccol] = rng.normal(size=6) * 10.0 ** rng.integers(-8, 3, size=6)
    X.dataframe_to_xlsx(res, tmp_path / "marker_3d_coordinates.xlsx")
    back = X.read_xlsx(tmp_path / "marker_3d_coordinates.xlsx")
    assert list(back.columns) == ["frameno", "row", "col", "X", "Y", "Z", "dX", "dY", "dZ", "displacement"]
    assert back["frameno"].tolist() == res["frameno"].tolist() and all(isinstance(v, int) for v in back["row"])
    for ccol in ("X", "Y", "Z", "dX", "dY", "dZ", "displacement"):
        assert np.array_equal(back[ccol].to_numpy(dtype=np.float64), res[ccol].to_numpy())
    # special characters and leading blanks in strings
    X.write_xlsx(tmp_path / "s.xlsx", ["k"], [["a<b & c"], ["  padded "], [None], [True]])
    assert X.read_rows(tmp_path / "s.xlsx") == [["k"], ["a<b & c"], ["  padded "], [None], [True]]


def test_csv_writer_equals_pandas(tmp_path):
    """`_save_results` writes the batches' columns with its own writer: byte for byte what `df.to_csv(index=False)`
    (`marker_detection.py:464-468`) writes, including integral floats, exponents, NaN, inf and the empty table."""
    import pandas as pd
    from vbs_amd.marker_detection import _write_csv_columns, CSV_COLUMNS
    rng = np.random.default_rng(1)
    n = 2000
    cols = {"frameno": rng.integers(0, 50, n), "row": rng.integers(0, 6, n), "col": rng.integers(0, 24, n)}
    for c in CSV_COLUMNS[3:]:
        cols[c] = rng.random(n) * 10.0 ** rng.integers(-8, 8, n)
    cols["Ox"][0] = 100.0; cols["Oy"][1] = np.nan; cols["Cx"][2] = 1e22; cols["Cy"][3] = np.inf; cols["angle"][4] = -0.0
    _write_csv_columns(tmp_path / "a.csv", cols)
    pd.DataFrame(cols, columns=CSV_COLUMNS).to_csv(tmp_path / "b.csv", index=False)
    assert (tmp_path / "a.csv").read_bytes() == (tmp_path / "b.csv").read_bytes()
    _write_csv_columns(tmp_path / "e.csv", {c: np.zeros(0) for c in CSV_COLUMNS})
    pd.DataFrame({c: [] for c in CSV_COLUMNS}, columns=CSV_COLUMNS).to_csv(tmp_path / "f.csv", index=False)
    assert (tmp_path / "e.csv").read_bytes() == (tmp_path / "f.csv").read_bytes()


def test_shard_bounds():
    from vbs_amd.dist import shard_bounds
    for n, w in ((32768, 8), (10, 4), (7, 8), (0, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(32768, 8, 3) == (12288, 16384)


def _gloo_worker(rank, world, port, n_total, out_dir, chunk=3):
    import torch
    import torch.distributed as td
    sys.path.insert(0, ROOT)
    from vbs_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = D.shard_bounds(n_total, world, rank)
        m = 7
        full = torch.arange(n_total * m * 10, dtype=torch.float32).reshape(n_total, m, 10)
        ids = xy = None
        if rank == 0:
            ids = np.array([[0, 0], [1, 0], [1, 1], [2, 0], [2, 1], [2, 2], [3, 0]])
            xy = np.linspace(0, 1, 14).reshape(7, 2) * 1e3 + 0.123456789
        ids, xy = D.broadcast_reference(ids, xy, torch.device("cpu"))
        gathered = D.gather_tables(full[a:b].clone(), n_total)
        # the pass-by-pass gather (chunk 3: several passes, the last one ragged when the shards differ)
        g = D.TableGather(n_total, m, 10, torch.device("cpu"), chunk)
        local = full[a:b].clone()
        for off in range(0, g.n_max, chunk):
            g.push(off, local[off:off + chunk])
        piped = g.finish()
        ok = torch.equal(gathered, full) and torch.equal(piped, full) and ids.shape == (7, 2) and ids[5].tolist() == [2, 2] \
            and abs(xy[6, 1] - (1e3 + 0.123456789)) < 1e-12
        open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "bad")
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 9])
def test_gather_tables_two_ranks_gloo(tmp_path, n_total):
    """N>1 path on CPU: broadcast of the reference table + the single all-gather, even and ragged shards."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"rank{r}.txt").read() for r in range(2)] == ["ok", "ok"]


def test_table_gather_eight_ranks_ragged_shards_gloo(tmp_path):
    """The world size the driver's scaling run uses (8 ranks), on CPU over gloo, with ragged shards (21 frames: five ranks
    of 3 and three of 2) and a pass width (2) that leaves some ranks without rows in the last pass: the single collective
    (`gather_tables`) and the pass-by-pass one (`TableGather`) both rebuild the table in frame order on every rank."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_worker, args=(8, port, 21, str(tmp_path), 2), nprocs=8, join=True)
    assert [open(tmp_path / f"rank{r}.txt").read() for r in range(8)] == ["ok"] * 8


def _loader_asm(rotated=True, copy_in_flight=False, moved=False, touch_other=False):
    """A stand-in for the disassembly of k_blur16's loader: four load groups, a loop of (wait, staging, refill) x 4 - with
    the last refill in the latch block in front of the header when `rotated` - and four full waits."""
    sets = [range(12, -1, -4), range(28, 15, -4), range(44, 31, -4), range(60, 47, -4)]

    def L(k, first=None):
        regs = list(sets[k])
        if first is not None:
            regs[0] = first
        return ["\t;;#ASMSTART"] + [f"\tglobal_load_dwordx4 v[{r}:{r + 3}], v{70 + i}, s[6:7]" for i, r in enumerate(regs)] + \
               ["\t;;#ASMEND"]

    def W(n):
        return ["\t;;#ASMSTART", f"\ts_waitcnt vmcnt({n})", "\t;;#ASMEND"]

    def stage(k):
        out = []
        for r in sets[k]:
            out += [f"\tv_xor_b32_e32 v{r}, 0x80808080, v{r}", f"\tds_write_b128 v80, v[{r}:{r + 3}]"]
        return out

    body = []
    for k in range(4):
        body += [f"\tv_add_u32_e32 v{70 + k}, s5, v65"] + L(k)
    body += ["\ts_mov_b32 s30, 0"] + (["\tv_mov_b32_e32 v90, v60"] if copy_in_flight else []) + ["\ts_branch .LBB0_3"]
    loop = []
    for k in range(4):
        loop.append(W(12) + stage(k) + (["\tv_mov_b32_e32 v91, v4"] if touch_other and k == 2 else []))
    if rotated:
        body += [".LBB0_2:"] + L(3, first=100 if moved else None) + [".LBB0_3:"]
        for k in range(4):
            body += loop[k] + (L(k) if k < 3 else ["\ts_branch .LBB0_2"])
    else:
        body += [".LBB0_3:"]
        for k in range(4):
            body += loop[k] + L(k, first=100 if moved and k == 3 else None)
        body += ["\ts_cbranch_scc0 .LBB0_3"]
    body += W(0) * 4 + ["\ts_endpgm"]
    return "\n".join(["_Z8k_blur16ILb0EEvPKh:"] + body + [".Lfunc_end0:"]) + "\n"


def test_blur16_isa_check_accepts_the_layouts_and_names_the_hazards():
    """vbs_amd/_isa_check.py (run by build() on the real disassembly): a loader whose in-flight load registers are only
    touched between their wait and their refill passes, rotated loop or not; a copy of a register whose load is still in
    flight, a register set that moved between the prologue and the loop, and staging code that touches another set each
    fail with a message that names the instruction."""
    from vbs_amd._isa_check import check_blur16
    for rot in (True, False):
        assert check_blur16(_loader_asm(rotated=rot)) == []
        p = check_blur16(_loader_asm(rotated=rot, copy_in_flight=True))
        assert len(p) == 1 and "v_mov_b32_e32 v90, v60" in p[0]
        p = check_blur16(_loader_asm(rotated=rot, moved=True))
        assert p and "register set moved" in p[0]
        p = check_blur16(_loader_asm(rotated=rot, touch_other=True))
        assert len(p) == 1 and "v_mov_b32_e32 v91, v4" in p[0] and "another set" in p[0]
    assert check_blur16("no kernels here\n") == ["no k_blur16 instantiation found in the assembly"]


def test_bench_cpu_limits_and_profile_selection(tmp_path, monkeypatch):
    """bench.py's measurement hygiene: the cgroup CPU quota / cpuset are parsed (v2 `cpu.max` and v1
    `cfs_quota_us` forms), the parallel-CPU probe reports a sane figure and the worker count is the smallest limit; the
    newest profiles/ file of a kind is chosen by round tag (r3pre < r3a < r3i < r4a), not by spelling."""
    sys.path.insert(0, ROOT)
    import bench
    quota, cpuset = bench._cgroup_quota_cpus()
    assert quota is None or quota > 0
    assert cpuset is None or cpuset >= 1
    pr = bench.probe_parallel_cpus(2, seconds=0.05)
    assert pr["effective_cpus"] >= 0.5 and "1" in pr["rate_vs_one_process"]
    # profile selection
    prof = tmp_path / "profiles"
    prof.mkdir()
    for tag in ("r1a", "r3pre", "r3a", "r3i", "r4a", "r2b"):
        (prof / f"{tag}_pmc_traffic_c3.json").write_text("{}")
    (prof / "notes_pmc_traffic_c3.json").write_text("{}")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert os.path.basename(bench._newest("*_pmc_traffic_c3.json")) == "r4a_pmc_traffic_c3.json"
    os.remove(prof / "r4a_pmc_traffic_c3.json")
    assert os.path.basename(bench._newest("*_pmc_traffic_c3.json")) == "r3i_pmc_traffic_c3.json"
    os.remove(prof / "r3i_pmc_traffic_c3.json"); os.remove(prof / "r3a_pmc_traffic_c3.json")
    assert os.path.basename(bench._newest("*_pmc_traffic_c3.json")) == "r3pre_pmc_traffic_c3.json"
    assert bench._newest("*_nothing.json") is None


def test_status_text_names_every_per_frame_status():
    from vbs_amd import _lib as L
    assert "workspace" in L.status_text(L.VBS_ECAPACITY, 512) and "512" in L.status_text(L.VBS_ECAPACITY, 512)
    assert "hand-shake" in L.status_text(L.VBS_EINTERNAL)
    assert L.VBS_EINTERNAL == -5
    hdr = open(os.path.join(ROOT, "include", "vbs.h")).read()
    assert "#define VBS_EINTERNAL  -5" in hdr


def test_mjpeg_entropy_decode_host_half(tmp_path):
    """f4(c), the host half of the native Motion-JPEG decoder without a GPU: `vbs_mjpeg_probe` reads the frame geometry,
    `vbs_mjpeg_entropy_batch` the quantised coefficients - checked by a float inverse DCT of them in NumPy against Pillow's
    decode of the same gray frame (libjpeg's integer IDCT is within one grey level of the exact transform) - for plain,
    restart-interval and optimised-table streams; progressive streams and garbage are refused."""
    import ctypes as C
    import io
    from PIL import Image
    from scipy.fft import idctn
    from vbs_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:61, 0:83]
    img = np.clip(128 + 80 * np.sin(xx / 6.0) * np.cos(yy / 9.0) + rng.normal(0, 10, (61, 83)), 0, 255).astype(np.uint8)
    for opts in ({}, {"restart_marker_blocks": 3}, {"optimize": True}, {"quality": 100}, {"no_dht": True}):
        bio = io.BytesIO()
        Image.fromarray(img).save(bio, format="JPEG", **{"quality": 80, **{k: v for k, v in opts.items() if k != "no_dht"}})
        data = bio.getvalue()
        if opts.get("no_dht"):                                       # a camera's frame: the standard tables are implied
            i, keep = 2, bytearray(data[:2])
            while data[i + 1] != 0xDA:
                ln = 2 + ((data[i + 2] << 8) | data[i + 3])
                if data[i + 1] != 0xC4:
                    keep += data[i:i + ln]
                i += ln
            data = bytes(keep + data[i:])
        info = (C.c_int32 * 8)()
        assert lib.vbs_mjpeg_probe(data, len(data), info) == 0
        assert list(info)[:5] == [83, 61, 1, 1, 1] and info[6] == 11 * 8 * 64
        two = data + data                                             # two frames in one buffer, two threads
        offs = np.array([0, len(data)], dtype=np.int64)
        sizes = np.array([len(data)] * 2, dtype=np.int32)
        nblk, cap = info[6] // 64, info[6] // 2
        ent = np.zeros(2 * cap, np.uint32)
        tab = np.zeros((2, nblk), np.uint32)
        fb = np.zeros(2, np.int64)
        reg = np.zeros(4, np.int64)
        qt = np.zeros((2, 3, 64), np.uint16)
        st = np.full(2, 99, np.int32)
        assert lib.vbs_mjpeg_entropy_batch(two, offs.ctypes.data, sizes.ctypes.data, 2, info, ent.ctypes.data, tab.ctypes.data,
                                           fb.ctypes.data, reg.ctypes.data, qt.ctypes.data, st.ctypes.data, 2) == 0
        assert not st.any() and list(fb) == [0, cap] and list(reg[::2]) == [0, cap] and 0 < reg[1] == reg[3] <= cap
        coef = np.zeros((2, nblk, 64), np.int16)                      # the compact form, expanded as the device kernel does
        dense_blocks = 0
        for i in range(2):
            for b in range(nblk):
                start, cnt = int(fb[i]) + (int(tab[i, b]) >> 7), int(tab[i, b]) & 127
                if cnt == 127:
                    coef[i, b] = ent[start:start + 32].view(np.int16)
                    dense_blocks += 1
                else:
                    assert cnt <= 32
                    e = ent[start:start + cnt]
                    assert len(set((e >> 16).tolist())) == cnt and ((e & 0xFFFF) != 0).all()
                    coef[i, b, (e >> 16).astype(np.int64)] = (e & 0xFFFF).astype(np.uint16).view(np.int16)
        assert np.array_equal(coef[0], coef[1]) and (dense_blocks > 0) == (opts.get("quality") == 100)
        coef = coef.reshape(2, -1)
        blocks = coef[0].reshape(8, 11, 8, 8).astype(np.float64) * qt[0, 0].reshape(8, 8)
        pix = idctn(blocks, axes=(2, 3), norm="ortho") + 128.0
        pix = pix.transpose(0, 2, 1, 3).reshape(64, 88)[:61, :83]
        want = np.asarray(Image.open(io.BytesIO(data)).convert("L")).astype(np.float64)
        assert np.abs(np.clip(pix, 0, 255) - want).max() <= 1.0, opts
    bio = io.BytesIO()
    Image.fromarray(img).save(bio, format="JPEG", progressive=True)
    info = (C.c_int32 * 8)()
    assert lib.vbs_mjpeg_probe(bio.getvalue(), len(bio.getvalue()), info) != 0
    junk = bytes(rng.integers(0, 256, 500, dtype=np.uint8))
    assert lib.vbs_mjpeg_probe(junk, len(junk), info) != 0
    assert lib.vbs_mjpeg_probe(data[:40], 40, info) != 0
    st = np.zeros(1, np.int32)
    offs, sizes = np.zeros(1, np.int64), np.array([len(junk)], np.int32)
    lib.vbs_mjpeg_probe(data, len(data), info)
    assert lib.vbs_mjpeg_entropy_batch(junk, offs.ctypes.data, sizes.ctypes.data, 1, info, ent.ctypes.data, tab.ctypes.data,
                                       fb.ctypes.data, reg.ctypes.data, qt.ctypes.data, st.ctypes.data, 1) == 1 and st[0] != 0


def test_mjpeg_host_half_survives_corrupt_streams():
    """f4(c): the entropy decoder reads bytes a camera or a disk produced.  Valid frames with random bytes overwritten (headers,
    tables, scan), cut short, or with an over-subscribed Huffman table must come back as a status - or as a well-formed
    block table when the damage still parses - never as a write past the buffers (guard words behind every output)."""
    import ctypes as C
    import io
    from PIL import Image
    from vbs_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:40, 0:56]
    img = np.clip(np.stack([128 + 90 * np.sin(xx / 5.0), 128 + 90 * np.cos(yy / 4.0), 3.0 * (xx + yy)], axis=2)
                  + rng.normal(0, 15, (40, 56, 3)), 0, 255).astype(np.uint8)
    streams = []
    for opts in (dict(subsampling=2), dict(subsampling=0), dict(subsampling=1, restart_marker_blocks=2), dict(gray=True)):
        bio = io.BytesIO()
        im = Image.fromarray(img[:, :, 0] if opts.pop("gray", False) else img)
        im.save(bio, format="JPEG", quality=85, **opts)
        streams.append(bio.getvalue())
    GUARD = 0xA5A5A5A5
    checked = parsed = 0
    for base in streams:
        info = (C.c_int32 * 8)()
        assert lib.vbs_mjpeg_probe(base, len(base), info) == 0
        cap, nblk = info[6] // 2, info[6] // 64
        sos = base.index(b"\xff\xda")
        for it in range(700):
            d = bytearray(base)
            kind = it % 5
            if kind == 0:                                           # bytes anywhere
                for _ in range(int(rng.integers(1, 12))):
                    d[int(rng.integers(2, len(d)))] = int(rng.integers(0, 256))
            elif kind == 1:                                         # bytes in the headers and tables
                for _ in range(int(rng.integers(1, 6))):
                    d[int(rng.integers(2, sos + 12))] = int(rng.integers(0, 256))
            elif kind == 2:                                         # cut short
                d = d[:int(rng.integers(2, len(d)))]
            elif kind == 3:                                         # markers sprayed into the scan
                for _ in range(int(rng.integers(1, 5))):
                    k = int(rng.integers(sos, len(d) - 1))
                    d[k] = 0xFF
                    d[k + 1] = int(rng.choice([0x00, 0xD0, 0xD3, 0xD9, 0xC4, 0xFF]))
            else:                                                   # an over-subscribed code-length histogram in a DHT
                k = bytes(d).index(b"\xff\xc4") + 5
                d[k + int(rng.integers(0, 4))] = int(rng.integers(3, 256))
            d = bytes(d)
            ent = np.full(cap + 64, GUARD, np.uint32)
            tab = np.full(nblk + 16, GUARD, np.uint32)
            fb = np.zeros(1, np.int64)
            reg = np.zeros(2, np.int64)
            qt = np.zeros((1, 3, 64), np.uint16)
            st = np.zeros(1, np.int32)
            offs, sizes = np.zeros(1, np.int64), np.array([len(d)], np.int32)
            bad = lib.vbs_mjpeg_entropy_batch(d, offs.ctypes.data, sizes.ctypes.data, 1, info, ent.ctypes.data, tab.ctypes.data,
                                              fb.ctypes.data, reg.ctypes.data, qt.ctypes.data, st.ctypes.data, 1)
            assert bad in (0, 1) and (bad == 1) == (st[0] != 0)
            assert (ent[cap:] == GUARD).all() and (tab[nblk:] == GUARD).all()
            checked += 1
            if bad == 0:
                parsed += 1
                cnt, start = tab[:nblk] & 127, tab[:nblk] >> 7
                assert ((cnt <= 32) | (cnt == 127)).all()
                words = np.where(cnt == 127, 32, cnt)
                assert (start + words <= reg[1]).all() and 0 <= reg[1] <= cap and reg[0] == 0
            lib.vbs_mjpeg_probe(d, len(d), (C.c_int32 * 8)())
    assert checked == 2800 and 200 < parsed < 2700                  # both outcomes occur


def test_mjpeg_decoder_class_host_half_without_a_gpu(tmp_path):
    """`video_io.MjpegDeviceDecoder` on the CPU device (host buffers only): batches that do not divide the clip, the
    refusals (`ValueError`: the tracker then stays with Pillow), a frame without a header as an `IOError` that names it."""
    from vbs_amd.video_io import AviReader, MjpegDeviceDecoder, write_avi
    spec = S.config1()
    frames = S.make_frames(spec, range(7), seed=2, channels=3)
    p = str(tmp_path / "clip.avi")
    write_avi(p, frames, quality=70)
    dec = MjpegDeviceDecoder(AviReader(p), "cpu", batch=3, threads=2)
    assert (dec.width, dec.height) == (spec.width, spec.height)
    assert [dec.entropy(0), dec.entropy(1), dec.entropy(0), dec.entropy(1)] == [3, 3, 1, 0]
    assert int(dec._regions[0][1]) > 0 and int(dec._fb[0][0]) == 0
    for bad_kind in ("raw", "progressive"):
        q = str(tmp_path / f"{bad_kind}.avi")
        write_avi(q, frames[:2], **({"codec": "RAW"} if bad_kind == "raw" else {"progressive": True}))
        with pytest.raises(ValueError):
            MjpegDeviceDecoder(AviReader(q), "cpu", batch=2)
    b = bytearray(open(p, "rb").read())
    off, size = AviReader(p)._frames[4]
    b[off:off + 4] = bytes(4)
    q = str(tmp_path / "bad.avi")
    open(q, "wb").write(bytes(b))
    dec = MjpegDeviceDecoder(AviReader(q), "cpu", batch=4, threads=3)
    assert dec.entropy(0) == 4
    with pytest.raises(IOError, match="frame 4"):
        dec.entropy(1)


def test_avi_reader_follows_opendml_continuation_chunks(tmp_path):
    """A recording beyond 1 GB continues in 'AVIX' RIFF chunks (OpenDML; what cv2.VideoWriter writes): the reader walks
    every one of them, on a memory-mapped file; the frames are those of the same clip written as one chunk."""
    from vbs_amd.video_io import AviReader, MjpegDeviceDecoder, write_avi, CAP_PROP_FRAME_COUNT
    spec = S.config1()
    frames = S.make_frames(spec, range(7), seed=4, channels=3)
    one, split = str(tmp_path / "one.avi"), str(tmp_path / "split.avi")
    write_avi(one, frames, quality=80)
    write_avi(split, frames, quality=80, riff_frames=3)
    assert open(split, "rb").read().count(b"AVIX") == 2
    a, b = AviReader(one), AviReader(split)
    assert a.get(CAP_PROP_FRAME_COUNT) == b.get(CAP_PROP_FRAME_COUNT) == 7
    na, fa = a.read_batch(7, threads=1)
    nb, fb = b.read_batch(7, threads=2)
    assert na == nb == 7 and np.array_equal(fa, fb)
    dec = MjpegDeviceDecoder(AviReader(split), "cpu", batch=7, threads=2)      # the native decoder reads the mapping in place
    assert dec.entropy(0) == 7
    for rd in (a, b):
        rd.release()
        assert not rd.isOpened() and rd.read() == (False, None)
    assert not AviReader(str(tmp_path / "missing.avi")).isOpened()
    open(str(tmp_path / "empty.avi"), "wb").close()
    assert not AviReader(str(tmp_path / "empty.avi")).isOpened()
