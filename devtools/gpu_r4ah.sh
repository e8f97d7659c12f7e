#!/bin/bash
# round 4: Winograd kernel with magic-number divisions in the per-task address set-up against the division sequences, both built on the box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4ah
mkdir -p $O
cd $R
LIB=speech-intent-recognizer_amd/lib
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_robustness_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
cp $LIB/libsir_hip.so $O/lib_new.so
touch speech-intent-recognizer_amd/csrc/model_infer.hip speech-intent-recognizer_amd/csrc/model_train.hip speech-intent-recognizer_amd/csrc/gru_quad.hip speech-intent-recognizer_amd/csrc/gru_pair.hip
make -C speech-intent-recognizer_amd/csrc EXTRA=-DSIR_W2_NP4 > $O/make.log 2>&1 || { tail -20 $O/make.log; exit 1; }
cp $LIB/libsir_hip.so $O/lib_old.so
for v in new old old new; do
  cp $O/lib_$v.so $LIB/libsir_hip.so
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$v" $O/tmp.json <<'PY' | tee -a $O/ab_div.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  conv2 %.1f conv3 %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["conv2_mfma_bn_relu_pool"], 1e3*k["conv3_mfma_bn_relu_pool"]))
PY
  timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag $v --kernels conv2,conv3,bwd_gru_d > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_div.txt
done
cp $O/lib_new.so $LIB/libsir_hip.so
rm -f $O/lib_new.so $O/lib_old.so
make -C speech-intent-recognizer_amd/csrc tools > $O/make_tools.log 2>&1 || { tail -20 $O/make_tools.log; exit 1; }
timeout -k 10 500 speech-intent-recognizer_amd/lib/bench_conv wino2 > $O/bench_conv_wino2.txt 2>&1 || { tail -20 $O/bench_conv_wino2.txt; exit 1; }
grep -E "^conv|f16x3 arith|f16x3 producer|^    \[" $O/bench_conv_wino2.txt | head -40
