"""Compile magnet_amd/csrc/*.hip for gfx950 into magnet_amd/csrc/libmagnet_hip.so (in-tree).

    python -m magnet_amd.build            # rebuild if sources are newer than the library
    python -m magnet_amd.build --force
    python -m magnet_amd.build --dev      # development library libmagnet_hip_dev.so: earlier kernel generations and run-time
                                          # ablations selectable through include/magnet_hip_dev.h (MAGNET_DEV_LIB=1 loads it);
                                          # the product library contains none of that

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmagnet_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", *os.environ.get("MAGNET_BUILD_DEFS", "").split()]   # MAGNET_BUILD_DEFS: extra -D... of an experiment build
# per-source additions: the lane-per-node read-out is one fully unrolled block; below the default pragma-unroll threshold
# hipcc unrolls it partially, indexes the row registers dynamically and puts them in scratch
# mpnn_edge_rows.hip: hipcc's SLP vectoriser pairs the epilogue's per-value fma into v_pk_fma_f32 and pays two v_mov per pair to line the
# operands up - packed fp32 beside MFMAs is slower than the scalar form to begin with
EXTRA_FLAGS = {"readout_kernels.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
               "mpnn_edge_rows.hip": ["-fno-slp-vectorize"], "mpnn_edge_stream.hip": ["-fno-slp-vectorize"], "mpnn_edge_rowsr.hip": ["-fno-slp-vectorize"], "mpnn_edge_rowsp.hip": ["-fno-slp-vectorize"], "mpnn_edge_bwd_wp.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
               "mpnn_edge_bwd_co.hip": ["-fno-slp-vectorize"]}


# earlier generations of the fused MPNN edge op (9-13): measured slower than or equal to k_mpnn_edge16 / k_mpnn_edge_stream32 and reached
# by no default path - compiled into the development library only (their entries: include/magnet_hip_dev.h, wrappers: dev_ops.py)
DEV_ONLY = ("mpnn_edge32t.hip", "mpnn_edge_ws.hip", "mpnn_edge_rows.hip", "mpnn_edge_rowsr.hip", "mpnn_edge_rowsp.hip")


def sources(dev: bool = True):
    every = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    return every if dev else [s for s in every if os.path.basename(s) not in DEV_ONLY]


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources(lib != LIB) + glob.glob(os.path.join(CSRC, "*.cuh"))
    return any(os.path.getmtime(p) > t for p in deps)


DEV_LIB = os.path.join(CSRC, "libmagnet_hip_dev.so")


def _compile(src: str, dev: bool = False) -> str:
    obj = os.path.splitext(src)[0] + (".dev.o" if dev else ".o")
    hdr = max(os.path.getmtime(p) for p in glob.glob(os.path.join(CSRC, "*.cuh")))
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr):
        return obj
    cmd = ["hipcc", *FLAGS, *(["-DMAGNET_DEV_VARIANTS"] if dev else []), *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src,
           "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True, dev: bool = False) -> str:
    lib = DEV_LIB if dev else LIB
    if not force and not needs_build(lib):
        return lib
    srcs = sources(dev)
    if force:
        for s in srcs:
            o = os.path.splitext(s)[0] + (".dev.o" if dev else ".o")
            if os.path.exists(o):
                os.remove(o)
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s_: _compile(s_, dev), srcs))
    r = subprocess.run(["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    if verbose:
        print(f"[magnet_amd.build] built {lib} from {len(srcs)} sources", file=sys.stderr)
    # the linked code objects are checked for the one hazard hipcc does not cover: an inline-asm VALU write that a v_mfma reads
    # fewer than two wait states later (magnet_amd/hazard_check.py) - a library with such a pair is not shipped
    # (not optional: without llvm-objdump the build fails - hazard_check.objdump() says where it looked)
    from . import hazard_check
    try:
        bad = hazard_check.check(lib)
    except Exception:
        os.remove(lib)
        raise
    if bad:
        os.remove(lib)
        lines = "\n".join(f"  {n[:60]}: {w}  ->  {m}" for n, _, _, w, m in bad[:8])
        raise RuntimeError(f"{len(bad)} hazard pair(s) without the wait states they need in {lib}:\n{lines}")
    if verbose:
        print("[magnet_amd.build] hazard check passed (VALU -> MFMA, MFMA -> VALU, transcendental -> VALU, VALU-SGPR -> VMEM, VALU -> lane ops, "
              "M0 -> LDS-DMA; loop seams included)", file=sys.stderr)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, dev="--dev" in sys.argv)
