#!/bin/bash
# MFMA utilisation and effective shader clock of the convolution kernels from rocprofv3 PMC counters (run on the GPU
# box from the repo root).  GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, "DVFS give-back"):
# effective clock = GRBM_GUI_ACTIVE / 8 / kernel wall time.  SQ_VALU_MFMA_BUSY_CYCLES counts matrix-core busy cycles
# over all SIMDs (16 per v_mfma_f32_16x16x32_bf16): utilisation = busy / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).
# Output: gpurun_out/mfma/r04_pmc_mfma.json
set -e
R=$PWD
mkdir -p $R/gpurun_out/mfma
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/mfma/pmc -- \
  python3 $R/bench.py --steps 2 --warmup 1 --bags-per-step ${BAGS:-64} --no-cpu-baseline --no-sublines > $R/gpurun_out/mfma/pmc.log 2>&1
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("$R/gpurun_out/mfma/pmc/*/*counter_collection.csv")[0]
def group(k):
    if "conv_halo" in k: return "conv_halo (3x3 stride-1 forward / data gradient, >=128 channels)"
    if "conv_pgemm" in k: return "conv_pgemm (stride-2 3x3 and 1x1 forward, stride-2 data gradients with >=128 outputs)"
    if "conv_igemm" in k: return "conv_igemm (generic: data gradients with 64 outputs, 1x1 data gradients)"
    if "conv3x3_c64" in k: return "conv3x3_c64p (64->64 forward / data gradient)"
    if "wgrad_c64_kernel" in k: return "wgrad_c64 (64->64 weight gradient)"
    if "wgrad_s2_kernel" in k: return "wgrad_s2 (stride-2 3x3 weight gradients, input staged once per tile)"
    if "wgrad_c128b_kernel" in k: return "wgrad_c128b (>=128-channel 3x3 weight gradients, 64 output channels per block)"
    if "wgrad_c128_kernel" in k: return "wgrad_c128 (128->128 weight gradient, 32 output channels per block)"
    if "conv_wgrad_kernel" in k: return "conv_wgrad (other weight gradients)"
    if "conv_stem" in k: return "conv_stem (forward + weight gradient)"
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
              "executed_bf16_TFLOPs": a["SQ_VALU_MFMA_BUSY_CYCLES"] / 16.0 * 16384.0 / (a["ns"] * 1e-9) / 1e12}
json.dump({"bags_per_step": int("${BAGS:-64}"), "note": "profiled pass (clocks ~3% below an un-profiled run)", "kernels": out},
          open("$R/gpurun_out/mfma/r04_pmc_mfma.json", "w"), indent=1)
print(json.dumps(out))
PY
