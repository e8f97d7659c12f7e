#!/bin/bash
# round 4, twelfth GPU session: conv1 variants (0 tile-by-tile, 1 strip + pipelined halves, 2 strip, one unit at a time)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4l
mkdir -p $O
cd $R
SIR_CONV1=4 timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu > $O/tests_c1.log 2>&1 || { tail -40 $O/tests_c1.log; exit 1; }
tail -2 $O/tests_c1.log
for c in 0 3 4 1 0 3 4 1; do
  SIR_CONV1=$c timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$c" $O/tmp.json <<'PY' | tee -a $O/ab_conv1.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("SIR_CONV1=%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  conv1 %.4f ms" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], k["conv1_bn_relu_pool"]))
PY
done
