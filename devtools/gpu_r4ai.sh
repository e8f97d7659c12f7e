#!/bin/bash
# round 4: eight producer waves (four waves per SIMD): producer priority 3 (default) / 1 / 0, against the four-producer build -- four libraries built on the box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4ai
mkdir -p $O
cd $R
LIB=speech-intent-recognizer_amd/lib
cp $LIB/libsir_hip.so $O/lib_np8p3.so
for v in "np8p1:-DSIR_W2_PRIO=1" "np8p0:-DSIR_W2_PRIO=0" "np4:-DSIR_W2_NP4"; do
  tag=${v%%:*}; fl=${v#*:}
  touch speech-intent-recognizer_amd/csrc/model_infer.hip speech-intent-recognizer_amd/csrc/model_train.hip
  make -C speech-intent-recognizer_amd/csrc EXTRA="$fl" > $O/make_$tag.log 2>&1 || { tail -20 $O/make_$tag.log; exit 1; }
  cp $LIB/libsir_hip.so $O/lib_$tag.so
done
for v in np4 np8p3 np8p1 np8p0 np4 np8p1 np8p0; do
  cp $O/lib_$v.so $LIB/libsir_hip.so
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$v" $O/tmp.json <<'PY' | tee -a $O/ab_np.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  conv2 %.1f conv3 %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["conv2_mfma_bn_relu_pool"], 1e3*k["conv3_mfma_bn_relu_pool"]))
PY
  timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag $v --kernels conv2,conv3 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_np.txt
done
cp $O/lib_np8p3.so $LIB/libsir_hip.so
rm -f $O/lib_*.so
