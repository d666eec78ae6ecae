#!/bin/bash
# The 1/2/4/8-GPU run sheet (north_star: "Mpixels/s at 1/2/4/8 GPUs reported next to the reference CPU path"): bench.py --gpus N for
# N in $GPUS, frames resident in HBM and -- second sweep -- delivered from pinned host memory (--h2d), the default workload (weak
# scaling: 32 frames per rank and pass) and configs[4] (the 10 000-frame stream, strong scaling), every rank's own rate kept.
#   bash scripts/scale_8gpu.sh [outfile.json]          on an 8-GPU node
#   GPUS="1 2" SHARE=1 FRAMES=12 BATCH=3 STEPS=1 INNER=1 bash scripts/scale_8gpu.sh   functional check on ONE GPU (ranks share device 0, gloo only)
# Asserts: the N = 1 value of the sweep agrees with a plain default line taken in the same call (within 10 %: box noise), every line
# says how many ranks its one RCCL collective saw (= N; 0 with SHARE=1, where no RCCL communicator can form), feature counts scale
# with the frames.  Efficiency is left to the reader of the JSON (the driver computes its own).
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale_8gpu.json}
GPUS=${GPUS:-"1 2 4 8"}
mkdir -p "$(dirname "$OUT")"
EXTRA=""; [ "${SHARE:-0}" = 1 ] && EXTRA="--share-device"
COMMON="--no-cpu --no-extra ${STEPS:+--steps $STEPS} ${INNER:+--inner $INNER} ${BATCH:+--batch $BATCH}"
TMP=$(mktemp -d)
line() { grep '^{' | tail -1; }
echo "[scale] reference line (N = 1, default flags except --no-cpu --no-extra)" >&2
python bench.py $COMMON 2>"$TMP/ref.err" | line > "$TMP/ref.json" || { tail -5 "$TMP/ref.err" >&2; exit 1; }
for mode in resident h2d; do
  H=""; [ $mode = h2d ] && H="--h2d"
  for n in $GPUS; do
    echo "[scale] default workload, $mode, N = $n" >&2
    python bench.py --gpus $n $EXTRA $H $COMMON 2>"$TMP/e" | line > "$TMP/c2_${mode}_$n.json" || { tail -5 "$TMP/e" >&2; exit 1; }
    echo "[scale] configs[4] stream (${FRAMES:-10000} frames), $mode, N = $n" >&2
    python bench.py --gpus $n $EXTRA $H --config 5 --frames ${FRAMES:-10000} --no-cpu ${BATCH:+--batch $BATCH} --max-parity-frames 4 2>"$TMP/e" | line > "$TMP/c5_${mode}_$n.json" || { tail -5 "$TMP/e" >&2; exit 1; }
  done
done
python - "$TMP" "$OUT" "$GPUS" "${SHARE:-0}" <<'PY'
import json, sys, os
tmp, out, gpus, share = sys.argv[1], sys.argv[2], [int(g) for g in sys.argv[3].split()], sys.argv[4] == "1"
ref = json.load(open(os.path.join(tmp, "ref.json")))
res = {"reference_line": {k: ref[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step")}, "runs": []}
for mode in ("resident", "h2d"):
    for cfg in ("c2", "c5"):
        for n in gpus:
            r = json.load(open(os.path.join(tmp, f"{cfg}_{mode}_{n}.json")))
            assert r["n_gpus"] == n, (cfg, mode, n, r["n_gpus"])
            seen = r["config"].get("rccl_ranks_seen")
            assert seen == (0 if share else n), f"{cfg} {mode} N={n}: the RCCL collective saw {seen} ranks"
            assert len(r["config"]["per_rank"]) == n
            res["runs"].append({"workload": "default (configs[1] + Canny, weak)" if cfg == "c2" else "configs[4] stream (strong)", "delivery": mode, "n_gpus": n,
                                "value": r["value"], "unit": r["unit"], "scaling": r["scaling"], "ms_per_step": r["ms_per_step"],
                                "feature_counts": r["config"]["feature_counts"], "rccl_ranks_seen": seen, "collectives": r["config"]["collectives"],
                                "per_rank": r["config"]["per_rank"]})
one = next(x for x in res["runs"] if x["delivery"] == "resident" and x["n_gpus"] == 1 and x["workload"].startswith("default"))
rel = abs(one["value"] - ref["value"]) / ref["value"]
res["n1_vs_reference_line_rel_diff"] = round(rel, 4)
assert rel < 0.10, f"N = 1 of the sweep ({one['value']}) and the plain default line ({ref['value']}) differ by {rel:.1%}"
# configs[4] is the SAME stream whatever N: its counts must not depend on the sharding, resident or delivered
c5 = [x["feature_counts"] for x in res["runs"] if x["workload"].startswith("configs[4]")]
assert all(c == c5[0] for c in c5), "configs[4]: feature counts depend on N or on the delivery"
json.dump(res, open(out, "w"), indent=1)
print(f"[scale] wrote {out}: " + ", ".join(f"{x['delivery']}/{'c2' if x['workload'].startswith('default') else 'c5'}/N{x['n_gpus']}={x['value'] / 1e3:.1f} Gpx/s" for x in res["runs"]))
PY
rc=$?; rm -rf "$TMP"; exit $rc
