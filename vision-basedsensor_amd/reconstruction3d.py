"""Drop-in for the reference's `code/Marker_Calibration/3d_reconstruction.py` (`Config`, `CONFIG`,
`CameraParameters`, `MarkerAnalysis`) with the undistort + closed-form 3-D solve + last-seen
displacement running on the MI355X (vbs_undistort_points / vbs_calculate_3d / vbs_solve3d /
vbs_displacement).

Published defects that a literal copy would inherit are fixed, nothing else is changed:
 * `Config.column_mapping` is a `default_factory` (the mutable default at `:28` raises at import);
 * no module-level FileHandler into a directory that does not exist yet (`:42`);
 * parameter tables: `.xlsx` is read by the package's own reader (`xlsx_io`: zipfile + xml.etree, no Excel engine
   needed) and the result sheet `marker_3d_coordinates.xlsx` (`:430-433`) is written by its writer; `.csv` / `.json`
   with the same `Parameter, Value` layout are accepted too; both spellings of the translation keys
   (`Tx_wc` `:121` / `T_wc_X` `extrinsic_calibration.py:135-151`) and of the key column
   (`Parameter` `:84` / `Param` `intrinsic_calibration.py:51`) are understood.
Plots (`:336-394`) are out of scope; `analyze_displacement` writes the statistics CSV only.
"""
from __future__ import annotations

import json
import logging
import traceback
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Tuple

import numpy as np
import pandas as pd

from . import _lib as L

logger = logging.getLogger(__name__)


@dataclass
class Config:
    """Configuration parameters for marker tracking and analysis (`:18-32`)."""
    marker_diameter_mm: float = 2.0
    warmup_frames: int = 100
    min_marker_size_px: float = 5.0
    max_displacement_px: float = 50.0
    data_dir: Path = Path("Results/data")
    output_dir: Path = Path("Results/data/results")
    plots_dir: Path = Path("Results/data/results/Displacement_Analysis_Plots")
    column_mapping: Dict[str, str] = field(
        default_factory=lambda: {"Cx": "u", "Cy": "v", "major_axis": "major_axis"})


CONFIG = Config()


class CameraParameters:
    """Container for camera intrinsic and extrinsic parameters (`:48-55`)."""

    def __init__(self):
        self.matrix: np.ndarray = None
        self.dist_coeffs: np.ndarray = None
        self.R_world_to_cam: np.ndarray = None
        self.T_world_to_cam: np.ndarray = None
        self.resolution: Tuple[int, int] = None


def _read_params(path: Path) -> pd.Series:
    path = Path(path)
    if path.suffix.lower() == ".json":
        return pd.Series(json.loads(path.read_text()))
    if path.suffix.lower() == ".csv":
        df = pd.read_csv(path)
    else:
        from .xlsx_io import read_xlsx
        df = read_xlsx(path)                       # what pd.read_excel(path) returns, without an Excel engine
    key = "Parameter" if "Parameter" in df.columns else "Param"
    if key not in df.columns or "Value" not in df.columns:
        raise ValueError(f"{path}: expected columns 'Parameter' (or 'Param') and 'Value', got {list(df.columns)}")
    # the extrinsics sheet carries title / date / blank rows (`extrinsic_calibration.py:135-141`): keep numeric values
    df = df[pd.to_numeric(df["Value"], errors="coerce").notna()]
    return df.set_index(key)["Value"].astype(float)


class MarkerAnalysis:
    """Marker tracking and 3-D displacement analysis (GPU implementation)."""

    def __init__(self, config: Config):
        self.config = config
        self.camera = CameraParameters()
        self._validate_paths()

    def _validate_paths(self) -> None:
        for path in (self.config.data_dir, self.config.output_dir, self.config.plots_dir):
            Path(path).mkdir(parents=True, exist_ok=True)

    # ---- `:70-130` ---------------------------------------------------------------------------------
    def load_parameters(self, intrinsic_path: Path, extrinsic_path: Path) -> None:
        try:
            p = _read_params(intrinsic_path)
            self.camera.matrix = np.array([[p["fx"], p.get("skew", 0), p["cx"]], [0, p["fy"], p["cy"]],
                                           [0, 0, 1]], dtype=np.float32)
            if self.camera.matrix[0, 0] <= 0 or self.camera.matrix[1, 1] <= 0:
                raise ValueError("Focal lengths must be positive")
            self.camera.dist_coeffs = np.array([p.get(k, 0) for k in ("k1", "k2", "p1", "p2", "k3")],
                                               dtype=np.float32)
            e = _read_params(extrinsic_path)
            self.camera.R_world_to_cam = np.array([[e[f"R_wc_{i}{j}"] for j in range(1, 4)]
                                                   for i in range(1, 4)], dtype=np.float32)
            if not np.allclose(self.camera.R_world_to_cam @ self.camera.R_world_to_cam.T, np.eye(3),
                               atol=1e-6):
                raise ValueError("Rotation matrix is not orthogonal")
            t = [e[a] if a in e.index else e[b] for a, b in (("Tx_wc", "T_wc_X"), ("Ty_wc", "T_wc_Y"),
                                                             ("Tz_wc", "T_wc_Z"))]
            self.camera.T_world_to_cam = np.array(t, dtype=np.float32).reshape(3, 1)
            logger.info("Camera parameters loaded successfully")
        except Exception as exc:
            logger.error(f"Failed to load camera parameters: {exc}")
            raise

    def set_camera(self, K, dist, R, T) -> None:
        """Set the camera from arrays (float32 casts as `load_parameters` applies)."""
        self.camera.matrix = np.asarray(K, dtype=np.float32).reshape(3, 3)
        if self.camera.matrix[0, 0] <= 0 or self.camera.matrix[1, 1] <= 0:
            raise ValueError("Focal lengths must be positive")
        d = np.zeros(5, dtype=np.float32)
        dd = np.asarray(dist, dtype=np.float32).ravel()[:5]
        d[:dd.size] = dd
        self.camera.dist_coeffs = d
        self.camera.R_world_to_cam = np.asarray(R, dtype=np.float32).reshape(3, 3)
        if not np.allclose(self.camera.R_world_to_cam @ self.camera.R_world_to_cam.T, np.eye(3), atol=1e-6):
            raise ValueError("Rotation matrix is not orthogonal")
        self.camera.T_world_to_cam = np.asarray(T, dtype=np.float32).reshape(3, 1)

    def _cam(self) -> L.Camera:
        c = self.camera
        return L.make_camera(c.matrix, c.dist_coeffs, c.R_world_to_cam, c.T_world_to_cam,
                             self.config.marker_diameter_mm)

    # ---- `:132-183` --------------------------------------------------------------------------------
    def load_marker_data(self, filepath: Path) -> pd.DataFrame:
        filepath = Path(filepath)
        if not filepath.exists():
            raise FileNotFoundError(f"Marker data file not found: {filepath}")
        try:
            try:
                import chardet
                with open(filepath, "rb") as f:
                    encoding = chardet.detect(f.read(30000))["encoding"] or "utf-8"
            except ImportError:
                encoding = "utf-8"
            df = pd.read_csv(filepath, sep=r"\s+|,|\t", encoding=encoding, engine="python",
                             skipinitialspace=True)
            df.columns = [c.strip() for c in df.columns]
            missing = (set(self.config.column_mapping.keys()) | {"frameno", "row", "col"}) - set(df.columns)
            if missing:
                raise ValueError(f"Missing required columns: {missing}")
            df = df.rename(columns=self.config.column_mapping)
            valid = df["major_axis"] >= self.config.min_marker_size_px
            if not valid.all():
                logger.warning(f"Filtered {len(df) - valid.sum()} markers for being too small")
                df = df[valid].copy()
            return df.sort_values("frameno", kind="stable").reset_index(drop=True)
        except Exception as exc:
            logger.error(f"Failed to load marker data: {exc}")
            raise

    # ---- `:185-238` --------------------------------------------------------------------------------
    def _undistort_points(self, points: np.ndarray) -> np.ndarray:
        from .engine import undistort_points
        return undistort_points(points, self._cam()).cpu().numpy().reshape(-1, 2)

    def _calculate_3d_position(self, u: float, v: float, diameter_px: float) -> np.ndarray:
        from .engine import calculate_3d
        xyz, ok = calculate_3d([[u, v, diameter_px]], self._cam())
        if not int(ok[0].item()):
            R = float(np.hypot(u - float(self.camera.matrix[0, 2]), v - float(self.camera.matrix[1, 2])))
            msg = "Marker too close to principal point" if R < 1e-6 else "Non-finite coordinates calculated"
            logger.warning(f"3D calculation failed: {msg}")
            raise ValueError(msg)
        return xyz[0].cpu().numpy()

    # ---- `:240-316` --------------------------------------------------------------------------------
    def _track_markers(self, df: pd.DataFrame) -> pd.DataFrame:
        """Rows `frameno,row,col,X,Y,Z,dX,dY,dZ,displacement` for every observation whose ID was seen
        before (after the warm-up), displacement measured against the last frame that ID was seen in."""
        import torch
        from .engine import undistort_points, calculate_3d, displacement_f64
        cols = ["frameno", "row", "col", "X", "Y", "Z", "dX", "dY", "dZ", "displacement"]
        if len(df) == 0:
            return pd.DataFrame(columns=cols)
        if self.config.warmup_frames > 0:
            df = df[df["frameno"] >= df["frameno"].min() + self.config.warmup_frames].copy()
        if len(df) == 0:
            return pd.DataFrame(columns=cols)
        cam = self._cam()
        uv = undistort_points(df[["u", "v"]].to_numpy(dtype=np.float64), cam)
        major = torch.as_tensor(df["major_axis"].to_numpy(dtype=np.float64), device=uv.device)
        xyz, ok = calculate_3d(torch.cat([uv, major.reshape(-1, 1)], dim=1), cam)
        # dense [frame, id] table (frames = the distinct framenos present, ids = distinct (row, col))
        frameno = df["frameno"].to_numpy()
        fvals, fidx = np.unique(frameno, return_inverse=True)
        keys = np.stack([df["row"].to_numpy(), df["col"].to_numpy()], axis=1)
        kvals, kidx = np.unique(keys, axis=0, return_inverse=True)
        kidx = kidx.reshape(-1)
        table = torch.zeros((len(fvals), len(kvals), L.TABLE_COLS), dtype=torch.float64, device=uv.device)
        fi = torch.as_tensor(fidx, device=uv.device)
        ki = torch.as_tensor(kidx, device=uv.device)
        table[fi, ki, 0] = (L.FLAG_TRACKED + L.FLAG_XYZ * ok.to(torch.float64))
        table[fi, ki, 3] = major
        table[fi, ki, 6:9] = xyz
        disp = displacement_f64(table, 0, -np.inf, self.config.max_displacement_px).cpu().numpy()
        table = table.cpu().numpy()
        sel = disp[fidx, kidx, 0] == 1
        out = np.concatenate([frameno[sel, None].astype(np.float64), keys[sel].astype(np.float64),
                              table[fidx[sel], kidx[sel], 6:9], disp[fidx[sel], kidx[sel], 1:5]], axis=1)
        res = pd.DataFrame(out, columns=cols)
        for c, src in (("frameno", "frameno"), ("row", "row"), ("col", "col")):
            res[c] = res[c].astype(df[src].dtype)
        return res

    # ---- `:318-403` (statistics only; plots are out of scope) ------------------------------------------
    def analyze_displacement(self, results_df: pd.DataFrame) -> None:
        if results_df.empty:
            logger.warning("No valid displacement data to analyze")
            return
        Path(self.config.plots_dir).mkdir(exist_ok=True)
        results_df = results_df.sort_values(["row", "col", "frameno"])
        results_df["cumulative_displacement"] = results_df.groupby(["row", "col"])["displacement"].cumsum()
        stats = results_df.groupby(["row", "col"]).agg({"displacement": ["mean", "std", "max"],
                                                        "cumulative_displacement": "last"})
        stats_path = Path(self.config.plots_dir) / "displacement_statistics.csv"
        stats.to_csv(stats_path)
        logger.info(f"Saved displacement statistics: {stats_path}")

    # ---- `:405-442` --------------------------------------------------------------------------------
    def run_analysis(self, input_csv: Path) -> None:
        try:
            logger.info("Starting marker analysis pipeline")
            if self.camera.matrix is None:
                base = Path(self.config.data_dir) / "PreprocessPara"
                self.load_parameters(base / "IntrinsicParameters.xlsx", base / "ExtrinsicParameters.xlsx")
            marker_df = self.load_marker_data(Path(input_csv))
            logger.info(f"Loaded {len(marker_df)} marker detections")
            results_df = self._track_markers(marker_df)
            if results_df.empty:
                raise ValueError("No valid 3D positions calculated")
            logger.info(f"Calculated 3D positions for {len(results_df)} marker observations")
            output_path = Path(self.config.output_dir) / "marker_3d_coordinates.xlsx"
            from .xlsx_io import dataframe_to_xlsx
            dataframe_to_xlsx(results_df, output_path)          # `results_df.to_excel(output_path, index=False)` :432
            logger.info(f"Saved 3D coordinates to {output_path}")
            self.analyze_displacement(results_df)
            logger.info("Analysis completed successfully")
        except Exception as exc:
            logger.error(f"Analysis failed: {exc}")
            logger.error(traceback.format_exc())
            raise


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    try:
        MarkerAnalysis(CONFIG).run_analysis(CONFIG.data_dir / "marker_locations_0.csv")
    except Exception as exc:
        logger.critical(f"Fatal error: {exc}")
        raise SystemExit(1)
