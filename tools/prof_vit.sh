#!/bin/bash
# rocprofv3 kernel statistics of the ViT-S/16 encoder bench (configs[4]) as bench.py runs it (run on the GPU box from the repo root).
# Output: gpurun_out/prof_vit/r03_vit_bench_kernel_stats.csv (copy to profiles/).
set -e
R=$PWD
mkdir -p $R/gpurun_out/prof_vit
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_vit/run
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vit/run -- \
  python3 $R/bench.py --config vit --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_vit/bench.json 2> $R/gpurun_out/prof_vit/bench.err
f=$(ls $R/gpurun_out/prof_vit/run/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_vit/r03_vit_bench_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_vit/r03_vit_bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d}  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
