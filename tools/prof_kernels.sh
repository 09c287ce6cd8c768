#!/bin/bash
# rocprofv3 kernel statistics of one bench.py configuration as the driver runs it (run on the GPU box from the repo root):
#   tools/prof_kernels.sh mil|gnn|vit|teacher|knn [extra bench.py arguments]
# Output: gpurun_out/prof_<cfg>/r04_<cfg>_bench_kernel_stats.csv (+ the bench line of the profiled run); copy to profiles/.
set -e
R=$PWD
cfg=${1:-mil}; shift || true
case $cfg in
  mil) ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sublines" ;;
  gnn) ARGS="--config gnn --steps 30 --warmup 5 --no-cpu-baseline" ;;
  vit) ARGS="--config vit --steps 10 --warmup 2 --no-cpu-baseline" ;;
  teacher) ARGS="--config teacher --steps 50 --warmup 5 --no-cpu-baseline" ;;
  knn) ARGS="--config knn --steps 10 --warmup 2 --no-cpu-baseline" ;;
  *) echo "unknown configuration $cfg"; exit 2 ;;
esac
D=$R/gpurun_out/prof_$cfg
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rm -rf $D/run
rocprofv3 --kernel-trace --stats --output-format csv -d $D/run -- \
  python3 $R/bench.py $ARGS "$@" > $D/bench.json 2> $D/bench.err
f=$(ls $D/run/*/*kernel_stats.csv | head -1)
cp $f $D/r04_${cfg}_bench_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$D/r04_${cfg}_bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d}  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
