#!/usr/bin/env bash
# Run on the GPU box (through gpurun): kernel-trace stats + HBM traffic counters for bench.py.
# usage: scripts/gpu_profile.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
# PMC passes are separate runs without any trace domain (gpurun refuses --pmc with sys/hip traces).
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-parity --no-end-to-end $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq1" -- $BENCH > "$OUT/bench_pmc_sq1.json" 2> "$OUT/pmc_sq1.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/bench_pmc_sq2.json" 2> "$OUT/pmc_sq2.err"
ls "$OUT"/*/* | head
