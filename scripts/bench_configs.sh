set -u
mkdir -p gpurun_out/cfgs
T0=$(date +%s)
for C in C1 C2 C3; do
  timeout 100 python bench.py --config $C --out gpurun_out/cfgs/bench_$C.json > /dev/null 2> gpurun_out/cfgs/bench_$C.err
  echo "$C rc=$? at $(( $(date +%s) - T0 )) s" >> gpurun_out/cfgs/steps.log
done
cat gpurun_out/cfgs/steps.log
