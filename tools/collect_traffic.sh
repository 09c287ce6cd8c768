#!/bin/bash
# HBM traffic of the conv kernels from rocprofv3 PMC counters (run on the GPU box from the repo root):
#   FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2).  Units are KiB; on gfx950
#   FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so it is DOUBLED
#   (/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Output: gpurun_out/traffic/r02_pmc_traffic.json,
#   stamped with the kernel source hash bench.py checks before it reports `roofline.traffic`.
set -e
R=$PWD
B=${BAGS:-32}
mkdir -p $R/gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic/$c -- \
    python3 $R/bench.py --steps 2 --warmup 1 --bags-per-step $B --no-cpu-baseline > $R/gpurun_out/traffic/$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$R")
import bench
out, steps = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/traffic/%s/*/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(lambda: [0.0, set()])
    adam = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "adam_step_kernel" in k:
            adam.add(r["Dispatch_Id"])
        # isic_conv2d_igemm_bf16 dispatches the generic implicit GEMM or the halo-resident 3x3 kernels
        name = ("conv_igemm" if ("conv_igemm" in k or "conv3x3_c64" in k or "conv_halo" in k or "conv_pgemm" in k) else
                "conv_wgrad" if ("wgrad" in k and "table" not in k and "reduce" not in k) else None)
        if name and r["Counter_Name"] == c:
            agg[name][0] += float(r["Counter_Value"]); agg[name][1].add(r["Dispatch_Id"])
    steps[c] = len(adam)                      # one AdamW launch per optimizer step (settle + warm-up + timed + instrumented)
    for k, (v, ids) in agg.items():
        out.setdefault(k, {})[c] = {"kib_total": v, "dispatches": len(ids)}
assert steps["FETCH_SIZE"] == steps["WRITE_SIZE"] and steps["FETCH_SIZE"] > 0, steps
res = {"kernel_source_hash": bench.kernel_source_hash(), "bags_per_step": $B, "patches": 64, "image_size": 224,
       "steps_profiled": steps["FETCH_SIZE"], "raw": out}
ig = out["conv_igemm"]
launches = 38 * steps["FETCH_SIZE"]   # bench counts isic_conv2d_igemm_bf16 calls: 19 forward + 19 data gradient per step
res["conv_igemm_hbm_bytes_per_launch"] = (2.0 * ig["FETCH_SIZE"]["kib_total"] + ig["WRITE_SIZE"]["kib_total"]) * 1024.0 / launches
res["note"] = "FETCH_SIZE doubled (gfx950 wide-read correction); per isic_conv2d_igemm_bf16 launch, 38 launches per step"
json.dump(res, open("$R/gpurun_out/traffic/r02_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
