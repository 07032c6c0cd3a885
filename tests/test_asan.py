"""The library's HOST translation units (sicp_api.cpp, sicp_io.cpp) under AddressSanitizer + UBSan: the build links them
against the unchanged device objects, a child interpreter with the sanitizer runtime preloaded drives the .xyz I/O and the
ABI's argument checks (CPU) and one whole ICP run (GPU box)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def run_exercise(*args, timeout=600):
    from simpleicp_amd import build
    lib = build.build_asan()
    env = dict(os.environ)
    env.update(SICP_LIBRARY=str(lib), LD_PRELOAD=str(build.asan_runtime()),
               # leaks: the interpreter's own allocations are not ours to judge; protect_shadow_gap: the HIP runtime maps
               # its SVM apertures where ASan would like an inaccessible gap
               ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "native" / "asan_exercise.py"), *args], env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "asan exercise OK" in r.stdout, f"exit {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-6000:]


def test_host_units_under_asan():
    run_exercise()


@pytest.mark.gpu
def test_icp_run_under_asan():
    run_exercise("--gpu")
