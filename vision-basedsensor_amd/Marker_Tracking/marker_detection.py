"""Path-compatible stand-in for the reference's `code/Marker_Tracking/marker_detection.py`:
`from marker_detection import MarkerTracker, find_marker, marker_center` keeps working when this
directory replaces the reference's on sys.path."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from vbs_amd.marker_detection import *  # noqa: E402,F401,F403
from vbs_amd.marker_detection import MarkerTracker, find_marker, marker_center  # noqa: E402,F401

if __name__ == "__main__":
    import runpy
    runpy.run_module("vbs_amd.marker_detection", run_name="__main__")
