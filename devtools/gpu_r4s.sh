#!/bin/bash
# round 4: Winograd producer / consumer kernel, f16x3: the fp16 split moved from the producers (the critical waves) to the consumers
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4s
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_robustness_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for c in 1 2; do
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$c" $O/tmp.json <<'PY' | tee -a $O/conv.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("run %s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  conv2 %.1f conv3 %.1f conv1 %.1f gemm0 %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["conv2_mfma_bn_relu_pool"], 1e3*k["conv3_mfma_bn_relu_pool"], 1e3*k["conv1_bn_relu_pool"], 1e3*k["gemm_ih_l0"]))
PY
done
timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag split_cons --kernels conv > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
tail -1 $O/tmp.json | tee -a $O/conv.txt
make -C speech-intent-recognizer_amd/csrc tools > $O/make_tools.log 2>&1 || { tail -20 $O/make_tools.log; exit 1; }
timeout -k 10 400 speech-intent-recognizer_amd/lib/bench_conv wino2 > $O/bench_conv_wino2.txt 2>&1 || { tail -20 $O/bench_conv_wino2.txt; exit 1; }
grep -E "^conv|f16x3|stamps" $O/bench_conv_wino2.txt | head -60
