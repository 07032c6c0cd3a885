#!/usr/bin/env bash
# round 6, call 9: k_hsel_all's last meeting replaced by a ticket -- large-Q tests + Q sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c9; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "windowed_rejection or longest_barrier or barrier_timeout or one_launch or q_sweep or large_q or duplicate" -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
timeout 600 python scripts/q_sweep.py 1e7 16384 32768 100000 196608 1000000 > $O/q_sweep.txt 2>&1; cat $O/q_sweep.txt
timeout 600 python scripts/steady_sweep.py 1e7 32768 100000 1000000 > $O/steady_sweep.txt 2>&1; cat $O/steady_sweep.txt
