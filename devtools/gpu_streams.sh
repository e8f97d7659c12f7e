#!/bin/bash
# bench.py at several --streams values (inference leg only matters); prints value / ms_per_step per setting
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for s in ${STREAMS:-1 2 3 4}; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --streams $s > gpurun_out/streams_$s.json 2> gpurun_out/streams_$s.err || { tail -5 gpurun_out/streams_$s.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/streams_$s.json").read().strip().splitlines()[-1])
print("streams", $s, d["value"], d["ms_per_step"], "train", d.get("train",{}).get("value"))
PY
done
