"""Path-compatible stand-in for the reference's `code/Marker_Calibration/3d_reconstruction.py`
(run it as a script, or load it with runpy / importlib since the name is not an identifier)."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from vbs_amd.reconstruction3d import Config, CONFIG, CameraParameters, MarkerAnalysis, logger  # noqa: E402,F401

if __name__ == "__main__":
    import logging
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    try:
        MarkerAnalysis(CONFIG).run_analysis(CONFIG.data_dir / "marker_locations_0.csv")
    except Exception as e:
        logger.critical(f"Fatal error: {e}")
        raise SystemExit(1)
