#!/usr/bin/env bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_terrestrial.py tests/test_gpu_operators.py tests/test_gpu_fuzz.py tests/test_gpu_run.py -q --maxfail=20 -p no:cacheprovider > $O/pytest_a.txt 2>&1
echo "a rc $?" >> $O/pytest_a.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --maxfail=20 -p no:cacheprovider -k "normals_vs_oracle or iteration_vs_oracle or knnk_bit_exact or many_queries_search_flavours or filtered_scan_equals or knn_sweep_random" > $O/pytest_b.txt 2>&1
echo "b rc $?" >> $O/pytest_b.txt
AB_EARLY=5 timeout 600 python scripts/match_ab.py 1e7 1e6 "near:SICP_NN16=near" "near-eager:SICP_NN16=near,SICP_BOXES=3" "near-sub4:SICP_NN16=near,SICP_SUB_TARGET=4" "near-sub4-eager:SICP_NN16=near,SICP_SUB_TARGET=4,SICP_BOXES=3" "near-sub8-noboxes:SICP_NN16=near,SICP_SUB_TARGET=8,SICP_BOXES=0" "far:SICP_NN16=far" > $O/match_ab_q1m.txt 2>&1
timeout 600 python scripts/q_cross.py 1e7 > $O/q_cross.txt 2>&1
( SICP_BOXES=0 timeout 200 python scripts/cold_match.py; echo "--- boxes eager"; SICP_BOXES=3 timeout 200 python scripts/cold_match.py ) > $O/cold_match.txt 2>&1
timeout 300 python scripts/dataset_profile.py webots bunny dragon multisensor > $O/dataset_profile.txt 2>&1
timeout 300 python scripts/datasets_run.py > $O/datasets_run.txt 2>&1
timeout 400 python bench.py --config T --no-cpu-baseline --throughput-q 0 --out $O/bench_T.json > $O/bench_T.line 2> $O/bench_T.err
timeout 400 python bench.py --config C3 --no-cpu-baseline --throughput-q 0 --out $O/bench_C3.json > $O/bench_C3.line 2> $O/bench_C3.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/calib_trace -- $REPO/scripts/ubench/gather_calib > $REPO/$O/calib_trace.out 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/iter0_trace -- python $REPO/scripts/cold_iter0.py 1e7 1e6 3 > $REPO/$O/iter0_trace.out 2>&1
cd $REPO
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r5b/calib_trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
for f in glob.glob("gpurun_out/r5b/iter0_trace/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "grid_nn" in r["Kernel_Name"] or "postmatch" in r["Kernel_Name"] or "hsel" in r["Kernel_Name"] or "lm_all" in r["Kernel_Name"] or "slot" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open("gpurun_out/r5b/iter0_sequence.txt", "w") as o:
        for r in rows[-40:]:
            o.write(f"{r['Kernel_Name'][:70]:70s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us\n")
    import os; os.remove(f)
PY
tail -n 3 $O/pytest_a.txt $O/pytest_b.txt
