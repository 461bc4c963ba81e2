"""MI355X-native marker tracking -> 3D displacement (drop-in for the reference's
`code/Marker_Tracking` + `code/Marker_Calibration/3d_reconstruction.py` hot path).

Import as `vbs_amd` (see `/vbs_amd.py`: the directory name is not a Python identifier).
Compute runs in hand-written HIP kernels behind the C-ABI of `include/vbs.h` (`csrc/libvbs.so`);
there is no CPU fallback: without the library, or without a GPU, the operators raise.
"""
__version__ = "0.1.0"
