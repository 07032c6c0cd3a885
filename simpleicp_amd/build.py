"""Builds libsimpleicp_hip.so (gfx950 only) in-tree with hipcc.

    python -m simpleicp_amd.build        # or simpleicp_amd.build.build()

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libsimpleicp_hip.so"
SOURCES = [CSRC / "sicp_api.cpp", CSRC / "sicp_clouds.cpp", CSRC / "sicp_search.cpp", CSRC / "sicp_icp.cpp", CSRC / "sicp_comm.cpp", CSRC / "sicp_kernels.hip", CSRC / "sicp_grid.hip", CSRC / "sicp_gridf.hip", CSRC / "sicp_tail.hip", CSRC / "sicp_reject.hip", CSRC / "sicp_lm.hip", CSRC / "sicp_io.cpp"]
HEADERS = [CSRC / "sicp_internal.h", CSRC / "sicp_host.h", CSRC / "sicp_lanes.h", CSRC / "sicp_solver.h", CSRC / "sicp_normals.h", CSRC / "sicp_grid_dev.h", PKG.parent / "include" / "simpleicp_hip.h"]
# -amdgpu-mfma-vgpr-form: MFMA results land in ordinary VGPRs (gfx950's register file is unified), so the VALU work
# that consumes them (min trees of the matrix-pipe filter, Gram folds) needs no v_accvgpr_read per register
COMPILE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
           "-fvisibility=hidden", "-Wall", "-pthread", "-mllvm", "-amdgpu-mfma-vgpr-form",
           # the first 16 dwords of a kernel's arguments (8 pointers) arrive in SGPRs with the wave instead of through a load
           # every dependent access queues behind; hot kernels order their parameters accordingly (falls back by itself
           # where the firmware does not preload)
           "-mllvm", "-amdgpu-kernarg-preload-count=16"]
LINK = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"]
OBJ = PKG / "_obj"
LIB_ASAN = OBJ / "libsimpleicp_hip_asan.so"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: cannot build libsimpleicp_hip.so")
    return exe


def is_stale(lib=LIB):
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def _run(cmd, verbose, what):
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed {what}")


def _objects(sources, suffix, extra, force, verbose):
    """One object per translation unit (compiled side by side), rebuilt when the source, a header or this file is newer."""
    OBJ.mkdir(exist_ok=True)
    deps = HEADERS + [Path(__file__)]
    jobs, objs = [], []
    for src in sources:
        obj = OBJ / (src.stem + suffix + ".o")
        objs.append(obj)
        if force or not obj.exists() or any(p.stat().st_mtime > obj.stat().st_mtime for p in [src] + deps):
            jobs.append([hipcc(), *COMPILE, *extra, "-c", "-o", str(obj), str(src)])
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as pool:
        list(pool.map(lambda c: _run(c, verbose, "compiling " + c[-1]), jobs))
    return objs


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    objs = _objects(SOURCES, "", [], force, verbose)
    _run([hipcc(), *LINK, "-o", str(LIB), *map(str, objs)], verbose, "linking libsimpleicp_hip.so")
    return LIB


def build_variant(name, defines, units=("sicp_tail",), verbose=False):
    """An instrumented build for measurements (e.g. -DSICP_SEL_FINE_TRACE): `units` recompiled with `defines`, everything else
    shared with the product build -> _obj/libsimpleicp_hip_<name>.so; load it with SICP_LIBRARY=<path>."""
    build(verbose=verbose)
    lib = OBJ / f"libsimpleicp_hip_{name}.so"
    special = [s for s in SOURCES if s.stem in units]
    objs = _objects(special, f".{name}", list(defines), True, verbose) + _objects([s for s in SOURCES if s.stem not in units], "", [], False, verbose)
    _run([hipcc(), *LINK, "-o", str(lib), *map(str, objs)], verbose, f"linking {lib.name}")
    return lib


def build_asan(force=False, verbose=False):
    """The same library with its HOST translation units (C ABI + .xyz I/O) under AddressSanitizer and UBSan; the device
    objects are shared with the product build.  Host units go through g++ and GCC's sanitizer runtime: the one shipped with
    ROCm's clang intercepts the HSA allocator for device-side ASan and cannot run next to an ordinary HIP process.
    Test infrastructure (tests/test_asan.py): load it with SICP_LIBRARY=<path> in a process that has asan_runtime() preloaded."""
    if not force and not is_stale(LIB_ASAN):
        return LIB_ASAN
    gxx = shutil.which("g++")
    if not gxx:
        raise RuntimeError("g++ not found")
    rocm = Path(hipcc()).resolve().parents[1]
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    OBJ.mkdir(exist_ok=True)
    objs = []
    for src in (s for s in SOURCES if s.suffix == ".cpp"):
        obj = OBJ / (src.stem + ".asan.o")
        objs.append(obj)
        _run([gxx, "-std=c++17", "-fPIC", "-fvisibility=hidden", "-pthread", "-D__HIP_PLATFORM_AMD__", f"-I{rocm / 'include'}", *san,
              "-c", "-o", str(obj), str(src)], verbose, f"compiling {src} for ASan")
    objs += _objects([s for s in SOURCES if s.suffix != ".cpp"], "", [], False, verbose)
    _run([gxx, "-shared", "-fPIC", "-pthread", "-fsanitize=address,undefined", "-o", str(LIB_ASAN), *map(str, objs),
          f"-L{rocm / 'lib'}", "-lamdhip64", f"-Wl,-rpath,{rocm / 'lib'}", "-ldl"], verbose, "linking the ASan build")
    return LIB_ASAN


def asan_runtime():
    r = subprocess.run([shutil.which("g++") or "g++", "-print-file-name=libasan.so"], capture_output=True, text=True)
    path = Path(r.stdout.strip())
    if r.returncode != 0 or not path.is_absolute() or not path.exists():
        raise RuntimeError("libasan.so not found")
    return path.resolve()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
