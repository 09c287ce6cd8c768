#!/bin/bash
# rocprofv3 kernel statistics of the headline bench command (configs[1]), run on the GPU box from the repo root.
# Output: gpurun_out/prof_mil/r03_bench_kernel_stats.csv (+ the bench line of the profiled run); copy to profiles/.
set -e
R=$PWD
mkdir -p $R/gpurun_out/prof_mil
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_mil/run
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mil/run -- \
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sublines > $R/gpurun_out/prof_mil/bench.json 2> $R/gpurun_out/prof_mil/bench.err
f=$(ls $R/gpurun_out/prof_mil/run/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_mil/r03_bench_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_mil/r03_bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{r["Name"][:86]:86s} calls {int(r["Calls"]):5d}  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
