#!/bin/bash
# inference pipeline over 2 / 3 / 4 slot streams at the round's final kernels (round 3 measured: 3 the same, 4 slower)
set -e
O=gpurun_out/r4ak; mkdir -p $O
for ns in 2 3 2 3 4; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-train --no-cpu-baseline --sustain-seconds 0 --streams $ns > $O/b$ns.json 2> $O/b$ns.err
  python - $O/b$ns.json $ns <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("streams", sys.argv[2], "utt/s", d["value"], "ms", d["ms_per_step"], "serial", d["single_stream"]["ms_per_step"])
PY
done
