#!/bin/bash
# round 4: forward recurrence with the gate arithmetic / global accesses in the coalesced thread layout (SIR_GQ_ROLES=1) against the MFMA layout
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4r
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_robustness_gpu.py tests/test_surface_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for v in 0 1 0 1; do
  SIR_GQ_ROLES=$v timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$v" $O/tmp.json <<'PY' | tee -a $O/ab_roles.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("SIR_GQ_ROLES=%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  gru l0 %.1f l1 %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["gru_recurrence_l0"], 1e3*k["gru_recurrence_l1"]))
PY
  SIR_GQ_ROLES=$v timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag roles$v --kernels train_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_roles.txt
done
