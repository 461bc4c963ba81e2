"""Path-compatible stand-in for the reference's `code/Marker_Tracking/tracking.py`."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from vbs_amd.tracking import process_video, find_marker, marker_center  # noqa: E402,F401

if __name__ == "__main__":
    process_video()
