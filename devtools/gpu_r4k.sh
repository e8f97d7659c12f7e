#!/bin/bash
# round 4, eleventh GPU session: conv1 forward with the column strip resident in LDS (SIR_CONV1=1) against the tile-by-tile kernel; 3 streams
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4k
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/tests_c1.log 2>&1 || { tail -40 $O/tests_c1.log; exit 1; }
tail -2 $O/tests_c1.log
for c in 0 1 0 1; do
  SIR_CONV1=$c timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$c" $O/tmp.json <<'PY' | tee -a $O/ab_conv1.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("SIR_CONV1=%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  conv1 %.4f ms" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], k["conv1_bn_relu_pool"]))
PY
done
for s in 2 3 4; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 12 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 --streams $s > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$s" $O/tmp.json <<'PY' | tee -a $O/ab_streams_infer.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("--streams %s infer %.1f utt/s  %.4f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
done
for c in 0 1; do
  SIR_CONV1=$c timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag conv1_$c --kernels train_conv1 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_conv1.txt
done
