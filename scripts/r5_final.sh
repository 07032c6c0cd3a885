#!/usr/bin/env bash
# the round's last GPU call: smoke, the records (scripts/final_measure.sh), then the suite without the two full-size files
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/final_r5c
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_r5c/smoke.txt 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/final_r5c/smoke.txt
TESTS_TIMEOUT=430 TESTS_ARGS="tests/ --ignore=tests/test_gpu_c5size.py --deselect tests/test_gpu_fullsize.py::test_select_in_range_at_full_size --deselect tests/test_gpu_fullsize.py::test_large_q_iterations_equal_oracle_at_full_size --deselect tests/test_gpu_fullsize.py::test_mid_q_iteration_equals_oracle_at_full_size --deselect tests/test_gpu_fullsize.py::test_normals_at_one_million_queries_equal_oracle" bash scripts/final_measure.sh r5c
