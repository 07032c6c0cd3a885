#!/usr/bin/env bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5i
timeout 150 python - > gpurun_out/r5i/op_profile.txt 2>&1 <<'PY'
import cProfile, pstats, sys, time, signal, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from conftest import load_cloud
import operator_flow
cache = {}
def clouds(stem):
    from pathlib import Path
    stem = Path(stem).stem
    if stem not in cache: cache[stem] = load_cloud(stem)
    return cache[stem]
pr = cProfile.Profile()
def dump(*a):
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
    sys.stdout.flush(); os._exit(0)
signal.signal(signal.SIGALRM, dump); signal.alarm(100)
t0 = time.time()
pr.enable()
operator_flow.reference_loop("webots", clouds)
pr.disable()
print("webots operator loop", time.time() - t0, "s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
PY
tail -60 gpurun_out/r5i/op_profile.txt
