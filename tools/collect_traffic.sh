#!/bin/bash
# HBM traffic of the conv kernels from rocprofv3 PMC counters (run on the GPU box from the repo root):
#   FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2).  Units are KiB; on gfx950
#   FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so it is DOUBLED
#   (/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Output: gpurun_out/traffic/*.csv
set -e
R=$PWD
mkdir -p $R/gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic/$c -- \
    python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic/$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/traffic/%s/*/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        # isic_conv2d_igemm_bf16 dispatches the generic implicit GEMM or, for the 64 -> 64 3x3 layers, the halo kernels
        name = ("conv_igemm" if ("conv_igemm" in k or "conv3x3_c64" in k) else
                "conv_wgrad" if ("conv_wgrad_kernel" in k or "wgrad_c64_kernel" in k or "wgrad_c128_kernel" in k) else None)
        if name and r["Counter_Name"] == c:
            agg[name][0] += float(r["Counter_Value"]); agg[name][1].add(r["Dispatch_Id"])
    for k, (v, ids) in agg.items():
        out.setdefault(k, {})[c] = {"kib_total": v, "dispatches": len(ids)}
res = {"bags_per_step": 16, "patches": 64, "image_size": 224, "raw": out}
ig = out["conv_igemm"]
# per C-ABI launch: a stride-2 data gradient is 4 kernel dispatches; bench counts isic_conv2d_igemm_bf16 calls (38 / step)
steps = 5  # 2 settle + 1 warm-up + 2 timed
launches = 38 * steps
res["conv_igemm_hbm_bytes_per_launch"] = (2.0 * ig["FETCH_SIZE"]["kib_total"] + ig["WRITE_SIZE"]["kib_total"]) * 1024.0 / launches
res["note"] = "FETCH_SIZE doubled (gfx950 wide-read correction); per isic_conv2d_igemm_bf16 launch, 38 launches per step"
json.dump(res, open("$R/gpurun_out/traffic/r01_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
