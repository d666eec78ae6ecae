#!/bin/bash
# configs.host_api and configs.5_h2d of bench.py alone (the boundary R users hit; config 5 with delivery)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/host
timeout 600 python - <<'PY' 2>&1 | grep -v "Librccl\|ROCm version\|Hostname" | tee gpurun_out/host/host_api.json
import json, bench
print(json.dumps({"host_api": bench.host_api_config(0)}))
print(json.dumps({"5_h2d": bench.stream_h2d_config(0)}))
PY
