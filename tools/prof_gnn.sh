#!/bin/bash
# rocprofv3 kernel statistics of the graph-path bench step as bench.py runs it (hipGraph replay) (run on the GPU box from the repo root).
# Output: gpurun_out/prof_gnn/r03_gnn_bench_kernel_stats.csv (copy to profiles/).
set -e
R=$PWD
mkdir -p $R/gpurun_out/prof_gnn
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_gnn/run
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gnn/run -- \
  python3 $R/bench.py --config gnn --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_gnn/bench.json 2> $R/gpurun_out/prof_gnn/bench.err
f=$(ls $R/gpurun_out/prof_gnn/run/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_gnn/r03_gnn_bench_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_gnn/r03_gnn_bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d}  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
