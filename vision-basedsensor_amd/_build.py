"""Build csrc/*.hip into csrc/libvbs.so for gfx950 with hipcc (in-tree, so the .so travels with gpurun)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libvbs.so")
SOURCES = ("api.hip", "k_blur.hip", "k_ncc.hip", "k_label.hip", "k_ccl.hip", "k_stage.hip", "k_stage_lat.hip", "k_solve.hip", "k_undistort.hip", "k_ids.hip", "host_csv.hip", "host_mjpeg.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# k_ncc.hip: no SLP pairings (they cost registers, 127 -> 108, and instructions) and no packed float32 instructions at all:
# next to matrix-core instructions a v_pk_fma_f32 costs more than the two v_fma_f32 it replaces (k_ncc_mfma 1.94 ->
# 1.90 us per frame with its float2 arithmetic split by the compiler)
FILE_FLAGS = {"k_ncc.hip": ["-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), suffix: str = "") -> str:
    """`extra_flags` / `suffix`: a second library next to the product one, e.g. tools/ build `libvbs_dbg.so` with
    -DVBS_DEBUG_KNOBS (phase-timing early exits read from the environment); the product library never has them."""
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "ccl_common.h"), os.path.join(CSRC, "stage_common.h"), os.path.join(CSRC, "track_common.h"), os.path.join(CSRC, "..", "..", "include", "vbs.h")]
    lib = LIB.replace(".so", suffix + ".so")
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", suffix + ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + list(extra_flags) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn, file=sys.stderr)
    _check_isa(force or any("k_blur.hip" in " ".join(j) for j in jobs), list(extra_flags), suffix, run)
    if force or jobs or _stale(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", lib] + objs)
    return lib


def _check_isa(rebuilt: bool, extra_flags, suffix, run):
    """k_blur16's loader keeps loads in flight across inline-asm statements: what the register allocator did with their
    destination registers is checked in the disassembly after every (re)build of k_blur.hip (vbs_amd/_isa_check.py says
    what is checked and why); a violation fails the build."""
    from . import _isa_check
    src = os.path.join(CSRC, "k_blur.hip")
    asm = os.path.join(CSRC, "k_blur" + suffix + ".s")
    stamp = asm + ".ok"
    if not rebuilt and os.path.exists(stamp) and os.path.getmtime(stamp) >= os.path.getmtime(src):
        return
    run([HIPCC] + FLAGS + FILE_FLAGS.get("k_blur.hip", []) + extra_flags + ["--cuda-device-only", "-S", src, "-o", asm])
    problems = _isa_check.check_blur16(open(asm).read())
    os.remove(asm)
    if problems:
        if os.path.exists(stamp):
            os.remove(stamp)
        raise RuntimeError("k_blur16 ISA check failed (vbs_amd/_isa_check.py):\n  " + "\n  ".join(problems))
    open(stamp, "w").write("ok\n")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
