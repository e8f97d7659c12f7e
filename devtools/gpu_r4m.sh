#!/bin/bash
# round 4: timing knock-outs of the conv1 forward kernel (strip form, one unit at a time; SIR_CONV1 = 16 + knock bits)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4m
mkdir -p $O
cd $R
for c in 0 2 17 18 19 20 22 23 24 31; do
  SIR_CONV1=$c timeout -k 10 300 python bench.py --steps 30 --warmup 10 --repeats 1 --no-cpu-baseline --no-train --sustain-seconds 0 --streams 1 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$c" $O/tmp.json <<'PY' | tee -a $O/knock_conv1.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("SIR_CONV1=%s conv1 %.1f us" % (sys.argv[1], 1e3*k["conv1_bn_relu_pool"]))
PY
done
