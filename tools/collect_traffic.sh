#!/bin/bash
# HBM traffic of the three roofline entries bench.py reports, from rocprofv3 PMC counters (run on the GPU box from
# the repo root):  mil = isic_conv2d_igemm_bf16 (conv forward + data gradient), gnn = isic_spmm_csr_f32,
# vit = isic_gemm_f16.
#   FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2).  Units are KiB; on gfx950
#   FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so it is DOUBLED
#   (/opt/skills/guides/MI355X_MICROARCH.md, HBM section).
# Output: profiles/r04_pmc_traffic.json (tracked; copied back through gpurun_out/traffic/), one section per entry,
# each stamped with the hash of the kernel sources behind it (bench.TRAFFIC_SOURCES) and its workload: bench.py
# refuses a section whose stamp differs from the code it runs, and tests/test_host_cpu.py fails on a stale one.
#   SECTIONS="mil gnn vit" (default) selects what to re-collect; other sections of an existing file are kept.
set -e
R=$PWD
B=${BAGS:-64}
SECTIONS=${SECTIONS:-"mil gnn vit"}
OUT=$R/gpurun_out/traffic
mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
for s in $SECTIONS; do
  case $s in
    mil) ARGS="--config mil --steps 2 --warmup 1 --bags-per-step $B --no-cpu-baseline --no-sublines" ;;
    gnn) ARGS="--config gnn --steps 4 --warmup 1 --no-cpu-baseline" ;;
    vit) ARGS="--config vit --steps 2 --warmup 1 --no-cpu-baseline" ;;
  esac
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/$s.$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$s.$c -- \
      python3 $R/bench.py $ARGS > $OUT/$s.$c.log 2>&1
    echo "[traffic] $s $c done"
  done
done
python3 - <<PY
import csv, glob, json, collections, os, sys
sys.path.insert(0, "$R")
import bench
path = bench.TRAFFIC_PROFILE
try:
    res = json.load(open(path))
except Exception:
    res = {}
# (kernel-name substrings of the entry, C-ABI launches per step or None = one launch per dispatch, step-marker kernel)
SPEC = {
    "mil": (("conv_igemm", "conv3x3_c64", "conv_halo", "conv_pgemm"), 35, "adam_step_kernel",
            {"bags_per_step": $B, "patches": 64, "image_size": 224}),
    "gnn": (("spmm_",), None, "adam_step_kernel", {"graphs_per_step": 668, "nodes": 196, "hidden": 128, "knn_k": 8}),
    "vit": (("gemm_f16_kernel",), None, None, {"images_per_step": 2048, "image_size": 224}),
}
for s in "$SECTIONS".split():
    subs, per_step, marker, workload = SPEC[s]
    tot, disp, steps = {}, {}, {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("$OUT/%s.%s/*/*counter_collection.csv" % (s, c))[0]
        v, ids, mk = 0.0, set(), set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if marker and marker in k:
                mk.add(r["Dispatch_Id"])
            if r["Counter_Name"] == c and any(t in k for t in subs):
                v += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
        tot[c], disp[c], steps[c] = v, len(ids), len(mk)
    assert disp["FETCH_SIZE"] == disp["WRITE_SIZE"] > 0, (s, disp)
    launches = per_step * steps["FETCH_SIZE"] if per_step else disp["FETCH_SIZE"]
    res[s] = {"kernel_source_hash": bench.kernel_source_hash(bench.TRAFFIC_SOURCES[s]), "workload": workload,
              "hbm_bytes_per_launch": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / launches,
              "fetch_kib_total": tot["FETCH_SIZE"], "write_kib_total": tot["WRITE_SIZE"], "dispatches": disp["FETCH_SIZE"],
              "launches": launches, "steps_profiled": steps["FETCH_SIZE"] if marker else None,
              "note": "FETCH_SIZE doubled (gfx950 wide-read correction); separate --pmc passes per counter"}
json.dump(res, open(path, "w"), indent=1, sort_keys=True)
os.makedirs("$OUT", exist_ok=True)
json.dump(res, open("$OUT/r04_pmc_traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res)[:1500])
PY
