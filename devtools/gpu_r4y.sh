#!/bin/bash
# round 4: first-poll delay of the two recurrence kernels re-swept after the AGPR pinning (SIR_GQ_DELAY, SIR_BQ_DELAY; x 64 cycles)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4y
mkdir -p $O
cd $R
for dly in 8 4 6 10 12 8; do
  SIR_GQ_DELAY=$dly timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$dly" $O/tmp.json <<'PY' | tee -a $O/ab_delay.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("SIR_GQ_DELAY=%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  gru l0 %.1f l1 %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], 1e3*k["gru_recurrence_l0"], 1e3*k["gru_recurrence_l1"]))
PY
done
for dly in 8 4 6 10 12 8; do
  SIR_BQ_DELAY=$dly SIR_GQ_DELAY=$dly timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag delay$dly --kernels gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_delay.txt
done
