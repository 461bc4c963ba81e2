export VBS_LIB_SUFFIX=_dbg
for b in 0 4 16 32; do for n in 0 16 32 64; do echo "blur16_nseg=$b ncc_nseg=$n"; VBS_BLUR16_NSEG=$b VBS_NCC_NSEG=$n timeout -k 10 120 python tools/gpu_single_frame.py 2>/dev/null | grep "stage impl" ; done; done
