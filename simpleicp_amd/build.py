"""Builds libsimpleicp_hip.so (gfx950 only) in-tree with hipcc.

    python -m simpleicp_amd.build        # or simpleicp_amd.build.build()

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libsimpleicp_hip.so"
SOURCES = [CSRC / "sicp_api.cpp", CSRC / "sicp_kernels.hip", CSRC / "sicp_grid.hip", CSRC / "sicp_tail.hip", CSRC / "sicp_lm.hip", CSRC / "sicp_io.cpp"]
HEADERS = [CSRC / "sicp_internal.h", CSRC / "sicp_lanes.h", CSRC / "sicp_solver.h", PKG.parent / "include" / "simpleicp_hip.h"]
# -amdgpu-mfma-vgpr-form: MFMA results land in ordinary VGPRs (gfx950's register file is unified), so the VALU work
# that consumes them (min trees of the matrix-pipe filter, Gram folds) needs no v_accvgpr_read per register
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden", "-Wall", "-pthread", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: cannot build libsimpleicp_hip.so")
    return exe


def is_stale():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), *FLAGS, "-o", str(LIB), *map(str, SOURCES)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libsimpleicp_hip.so")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
