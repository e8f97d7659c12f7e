#!/bin/bash
# round 4, second GPU session: the whole GPU suite on the f16x3 forward (GEMM + Winograd convs), bench_conv with the f16x3 lines,
# in-model A/B of SIR_F16, one full bench line (run via gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > $O/gpu_tests.log 2>&1; rc=$?
tail -4 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert |trajectory" $O/gpu_tests.log | head -40; exit 1; fi
grep -E "trajectory|clips, .* predicted classes" $O/gpu_tests.log | head
timeout -k 10 400 speech-intent-recognizer_amd/lib/bench_conv wino2 > $O/bench_conv_wino2.txt 2>&1 || { tail -20 $O/bench_conv_wino2.txt; exit 1; }
grep -E "^conv|f16x3|rotating" $O/bench_conv_wino2.txt | head -40
for m in 3 0 3 0; do
  SIR_F16=$m timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$m" $O/tmp.json <<'PY' | tee -a $O/ab_f16_infer.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("SIR_F16=%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  kernels %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], json.dumps(d["kernels_avg_ms"])))
PY
done
for m in 3 0 3 0; do
  SIR_F16=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag f16_$m --kernels train_ > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_f16_train.jsonl
done
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: infer", d["value"], "ms", d["ms_per_step"], "train", d["train"]["value"], d["train"]["ms_per_step"])
print(json.dumps(d["train"].get("dropin_epoch"), indent=1))
print(json.dumps(d["roofline"].get("by_symbol")))
PY
