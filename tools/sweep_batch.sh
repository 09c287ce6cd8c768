#!/bin/bash
# Developer tool: per-step batch sweep of the sub-benchmarks / the headline (run on the GPU box from the repo root):
#   tools/sweep_batch.sh heads   teacher + GNN at 256 / 512 / 668 / 1024 bags (graphs) per step
#   tools/sweep_batch.sh mil     configs[1] at 32 / 48 / 64 bags per step
mkdir -p gpurun_out
if [ "${1:-heads}" = "mil" ]; then
  for b in 32 48 64; do
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sublines --bags-per-step $b > gpurun_out/r4_sw_m$b.json 2>/dev/null
    python - <<PY
import json
m=json.load(open("gpurun_out/r4_sw_m$b.json"))
print("bags/step=$b mil", round(m["value"],1), round(m["ms_per_step"],3), round(m["roofline"]["frac"],4), flush=True)
PY
  done
  exit 0
fi
for b in 256 512 668 1024; do
  python bench.py --config teacher --steps 40 --warmup 5 --no-cpu-baseline --teacher-bags-per-step $b > gpurun_out/r4_sw_t$b.json 2>/dev/null
  python bench.py --config gnn --steps 30 --warmup 5 --no-cpu-baseline --graphs-per-step $b > gpurun_out/r4_sw_g$b.json 2>/dev/null
  python - <<PY
import json
t=json.load(open("gpurun_out/r4_sw_t$b.json")); g=json.load(open("gpurun_out/r4_sw_g$b.json"))
print("B=$b teacher", round(t["value"]), round(t["ms_per_step"],4), "tuned", round(t["tuned"]["value"]), "| gnn", round(g["value"]), round(g["ms_per_step"],4), "spmm frac", round(g["roofline"]["frac"],3), flush=True)
PY
done
