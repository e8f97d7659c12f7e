#!/bin/bash
# round 4: conv1 forward with VGPR-form MFMAs (no accumulator copies) and a packed-f32 epilogue
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4o
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for c in 1 2; do
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$c" $O/tmp.json <<'PY' | tee -a $O/conv1.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("run %s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  conv1 %.1f us  feat %.1f conv2 %.1f" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["conv1_bn_relu_pool"], 1e3*k["feat_frames"], 1e3*k["conv2_mfma_bn_relu_pool"]))
PY
done
timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag conv1pk --kernels conv1 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
tail -1 $O/tmp.json | tee -a $O/conv1.txt
