#!/bin/bash
# round 4: conv1 forward on the fp16 matrix pipe (f16x3, SIR_F16 bit 6) against the f32-MFMA kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4x
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_robustness_gpu.py -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for v in 15 15; do
  SIR_WINO2=$v timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$v" $O/tmp.json <<'PY' | tee -a $O/ab_conv1_f16.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("SIR_F16=%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  conv2 %.1f conv3 %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["conv2_mfma_bn_relu_pool"], 1e3*k["conv3_mfma_bn_relu_pool"]))
PY
done
for v in 15; do
  SIR_WINO2=$v timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag f16_$v --kernels conv > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_conv1_f16.txt
done
