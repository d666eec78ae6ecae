#!/bin/bash
# PMC passes on the structure-tensor kernel (20 B/px doorway, batch 32, scripts/k3_variants.py): separate passes, no tracing.
# Writes gpurun_out/k3/k3_pmc.txt and gpurun_out/k3/k3_traffic.json (copy the latter to profiles/k3_traffic.json: bench.py
# reads it for roofline.traffic and ignores it when the kernel sources have changed since -- kernel_source_sha1).
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/k3"; mkdir -p "$O"
export TMPDIR=/tmp
KERN="${1:-fir_tensor}"
cd /tmp
pmc() {
  name=$1; shift
  BATCHES=32 ITERS=4 timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$O/pmc_$name" -o p -- python $R/scripts/k3_variants.py > "$O/pmc_$name.log" 2>&1
  f=$(find "$O/pmc_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$KERN" "$O/pmc_$name.json" <<'PY'
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if sys.argv[2] not in k: continue
    agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
        out[c] = sum(v) / len(v)
    out["kernel"] = k
json.dump(out, open(sys.argv[3], "w"))
PY
  rm -rf "$O/pmc_$name"
}
{
echo "=== PMC pass 1"; pmc a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
echo "=== PMC pass 2"; pmc b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY
echo "=== PMC pass 3"; pmc c FETCH_SIZE GRBM_GUI_ACTIVE
echo "=== PMC pass 4"; pmc d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
} > "$O/k3_pmc.txt" 2>&1
cd "$R"
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
O = "gpurun_out/k3"
c = json.load(open(f"{O}/pmc_c.json")); d = json.load(open(f"{O}/pmc_d.json"))
B, px = 32, 3840 * 2160
fetch = 2 * 1024 * c["FETCH_SIZE"]          # KiB; doubled per the gfx950 note of MI355X_MICROARCH.md (HBM section)
write = 1024 * d["WRITE_SIZE"]
alg = 20 * px * B
json.dump({"kernel": c["kernel"], "workload": "scripts/k3_variants.py, 32 frames 3840x2160 per launch (the roofline entry's launch)", "batch": B,
           "collected_with": "rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, no tracing): scripts/gpu_pmc_k3.sh",
           "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": d["WRITE_SIZE"], "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE"),
           "TCC_HIT_sum": d.get("TCC_HIT_sum"), "TCC_MISS_sum": d.get("TCC_MISS_sum"),
           "correction": "gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced reads -> doubled; WRITE_SIZE taken as reported",
           "fetch_bytes": int(fetch), "write_bytes": int(write), "traffic_bytes_per_launch": int(fetch + write), "traffic_bytes_per_frame": (fetch + write) / B,
           "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((fetch + write) / alg, 4),
           "kernel_source_sha1": bench.kernel_source_hash()}, open(f"{O}/k3_traffic.json", "w"), indent=1)
print(open(f"{O}/k3_traffic.json").read())
PY
exit 0
