#!/bin/bash
# MFMA utilisation and effective shader clock of the graph step's exact-fp32 GEMM kernels from rocprofv3 PMC counters (run on the
# GPU box from the repo root; eager launch).  As tools/collect_mfma.sh: effective clock = GRBM_GUI_ACTIVE / 8 / kernel wall time;
# SQ_VALU_MFMA_BUSY_CYCLES counts matrix-core busy cycles over all SIMDs (32 per v_mfma_f32_16x16x4_f32, 2048 FLOP each).
# Output: gpurun_out/mfma/r04_pmc_mfma_gnn.json
set -e
R=$PWD
mkdir -p $R/gpurun_out/mfma
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/mfma/pmc_gnn
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/mfma/pmc_gnn -- \
  python3 $R/bench.py --config gnn --steps 4 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/mfma/pmc_gnn.log 2>&1
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("$R/gpurun_out/mfma/pmc_gnn/*/*counter_collection.csv")[0]
def group(k):
    if "gemm_f32p_kernel" in k: return "gemm_f32p (persistent 256 x 128 x 32: input projection forward / weight gradient, K = 512 products)"
    if "gemm_rowpanel_kernel" in k: return "gemm_rowpanel (K <= 128: GCNConv.lin forward / data gradient, attention hidden layer)"
    if "gemm_tn_skinny_kernel" in k: return "gemm_tn_skinny (128 x 128 weight gradients over all nodes)"
    if "gemm_f32_kernel" in k: return "gemm_f32 (64 x 64 x 16: what is left)"
    return None
agg = collections.defaultdict(lambda: {"ns": 0.0, "GRBM_GUI_ACTIVE": 0.0, "SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "ids": set()})
for r in csv.DictReader(open(f)):
    g = group(r["Kernel_Name"])
    if not g: continue
    a = agg[g]
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in a["ids"]:
        a["ids"].add(r["Dispatch_Id"]); a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for g, a in agg.items():
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    out[g] = {"dispatches": len(a["ids"]), "total_ms": a["ns"] / 1e6, "effective_clock_GHz": cyc / a["ns"],
              "mfma_busy_fraction": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
              "executed_f32_TFLOPs": a["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0 * 2048.0 / (a["ns"] * 1e-9) / 1e12}
json.dump({"workload": "bench.py --config gnn (256 graphs x 196 nodes per step), eager launch, profiled pass", "kernels": out},
          open("$R/gpurun_out/mfma/r04_pmc_mfma_gnn.json", "w"), indent=1)
print(json.dumps(out))
PY
