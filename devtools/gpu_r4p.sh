#!/bin/bash
# round 4: recurrence kernels with their resident W_hh fragments pinned in AGPRs (inline-asm MFMAs, no per-step register copies)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4p
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_robustness_gpu.py tests/test_nccl_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for c in 1 2; do
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$c" $O/tmp.json <<'PY' | tee -a $O/gru.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("run %s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  gru l0 %.1f l1 %.1f us  conv2 %.1f" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["gru_recurrence_l0"], 1e3*k["gru_recurrence_l1"], 1e3*k["conv2_mfma_bn_relu_pool"]))
PY
done
for c in 1 2; do
timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag agpr --kernels gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
tail -1 $O/tmp.json | tee -a $O/gru.txt
done
