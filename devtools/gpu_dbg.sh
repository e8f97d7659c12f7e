#!/bin/bash
# timing-only sweep of one debug env var (no parity tests): devtools/gpu_dbg.sh VAR v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train > gpurun_out/dbg_$v.json 2> gpurun_out/dbg_$v.err || { tail -5 gpurun_out/dbg_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/dbg_$v.json").read().strip().splitlines()[-1])
print("$VAR=$v", d["value"], {k: round(x*1e3,1) for k,x in d["kernels_avg_ms"].items() if x and k.startswith("gru")})
PY
done
