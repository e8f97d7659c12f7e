#!/bin/bash
# round 4, sixth GPU session: four-k BPTT (consecutive k, wave-split exchange), poll-delay sweep, host-thread cap in the probe, bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4f
mkdir -p $O
cd $R
SIR_BPTT=4 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_robustness_gpu.py tests/test_nccl_gpu.py -x -q -m gpu > $O/tests_k4.log 2>&1 || { tail -30 $O/tests_k4.log; exit 1; }
tail -2 $O/tests_k4.log
for m in 0 4 0 4 37 38 40; do
  SIR_BPTT=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag bptt$m --kernels bwd_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_bptt_k4.jsonl
done
for dly in 16 4 8 12 20 16; do
  SIR_GQ_DELAY=$dly timeout -k 10 300 python bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-train --sustain-seconds 0 > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  python - "$dly" $O/tmp.json <<'PY' | tee -a $O/ab_gq_delay.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]
print("SIR_GQ_DELAY=%s infer %.1f utt/s  %.4f ms/step  serial %.4f ms  gru l0 %.4f l1 %.4f" % (sys.argv[1], d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], k["gru_recurrence_l0"], k["gru_recurrence_l1"]))
PY
done
timeout -k 10 400 python devtools/dataloader_probe.py 2048 staged_only,staged_train,none,train_step > $O/probe3.jsonl 2> $O/probe3.err || { tail -20 $O/probe3.err; exit 1; }
cat $O/probe3.jsonl
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: infer", d["value"], "ms", d["ms_per_step"], "serial", d["single_stream"]["ms_per_step"], "train", d["train"]["value"], d["train"]["ms_per_step"], "aug", d["train_aug"]["value"])
de=d["train"].get("dropin_epoch", {})
for k in ("dataloader","dataloader_staged","hbm_feature_store","waveform_store"):
    print(k, json.dumps(de.get(k)))
print(json.dumps(d["roofline"]))
PY
