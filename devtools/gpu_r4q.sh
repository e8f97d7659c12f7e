#!/bin/bash
# round 4: A/B of the AGPR-pinned recurrence kernels against the compiler-managed build, both built ON the box (one box, alternating)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4q
mkdir -p $O
cd $R
LIB=speech-intent-recognizer_amd/lib
cp $LIB/libsir_hip.so $O/lib_agpr.so
touch speech-intent-recognizer_amd/csrc/gru_quad.hip
make -C speech-intent-recognizer_amd/csrc EXTRA=-DSIR_GQ_BUILTIN_MFMA > $O/make.log 2>&1 || { tail -20 $O/make.log; exit 1; }
cp $LIB/libsir_hip.so $O/lib_builtin.so
for v in agpr builtin agpr builtin; do
  cp $O/lib_$v.so $LIB/libsir_hip.so
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$v" $O/tmp.json <<'PY' | tee -a $O/ab_agpr.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  gru l0 %.1f l1 %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["gru_recurrence_l0"], 1e3*k["gru_recurrence_l1"]))
PY
  timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag $v --kernels gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_agpr.txt
done
rm -f $O/lib_agpr.so $O/lib_builtin.so
