#!/bin/bash
# HBM-side traffic of whole functions (every kernel they launch): imgfd_canny_dev on 4K frames, imgfd_fhog_dev and imgfd_surf_dev on
# 4096^2 tiles.  Separate PMC passes (FETCH_SIZE / WRITE_SIZE never together, no tracing beside them), the gfx950 correction of
# MI355X_MICROARCH.md (FETCH_SIZE doubled).  Writes gpurun_out/traffic/function_traffic.json -- copy it to profiles/function_traffic.json:
# bench.py reads it for the roofline entries of configs 3 / 4 and ignores an entry whose kernel sources have changed since.
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/traffic"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
run() {  # name, counter, command...
  name=$1; ctr=$2; shift 2
  rm -rf /tmp/ft_$name_$ctr
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/ft_${name}_$ctr -o p -- "$@" > /dev/null 2>&1
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  BATCH=8 ITERS=1 run canny $ctr env BATCH=8 ITERS=1 python $R/scripts/canny_time.py
  run fhog $ctr env TILES=4 python $R/scripts/fhog_time.py
  run surf $ctr env TILES=8 python $R/scripts/surf_dev_time.py
done
cd "$R"
python - <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
# calls of the function each script makes (warm-up + timed) and the units (frames / tiles) per call
plan = {"canny": (3 + 1, 8, "4K frame (3840x2160)"), "fhog": (2 + 5, 4, "4096x4096 RGB tile"), "surf": (2 + 4, 8, "4096x4096 RGB tile")}
skip = ("synth", "at::", "elementwise", "copyBuffer")
out = {}
for name, (calls, units, what) in plan.items():
    tot = {}
    kernels = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        s = 0.0
        for fn in glob.glob(f"/tmp/ft_{name}_{ctr}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"]
                if r["Counter_Name"] != ctr or any(x in k for x in skip):
                    continue
                v = float(r["Counter_Value"])
                s += v
                kk = k.replace("void ", "").split("(")[0][:40]
                kernels.setdefault(kk, {}).setdefault(ctr, 0.0)
                kernels[kk][ctr] += v
        tot[ctr] = s
    fetch = 2 * 1024 * tot["FETCH_SIZE"] / (calls * units)
    write = 1024 * tot["WRITE_SIZE"] / (calls * units)
    px = 3840 * 2160 if name == "canny" else 4096 * 4096
    out[name] = {"unit": what, "calls_measured": calls, "units_per_call": units, "fetch_bytes_per_unit": fetch, "write_bytes_per_unit": write,
                 "traffic_bytes_per_unit": fetch + write, "traffic_bytes_per_pixel": round((fetch + write) / px, 3),
                 "per_kernel_bytes_per_pixel": {k: {"fetch": round(2 * 1024 * v.get("FETCH_SIZE", 0) / (calls * units) / px, 3),
                                                    "write": round(1024 * v.get("WRITE_SIZE", 0) / (calls * units) / px, 3)} for k, v in sorted(kernels.items())},
                 "collected_with": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing), FETCH_SIZE x 2 (gfx950): scripts/gpu_pmc_functions.sh",
                 "kernel_source_sha1": bench.kernel_source_hash(bench.FUNCTION_SOURCES[name])}
json.dump(out, open("gpurun_out/traffic/function_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, v["traffic_bytes_per_pixel"], "B/px", json.dumps(v["per_kernel_bytes_per_pixel"]))
PY
