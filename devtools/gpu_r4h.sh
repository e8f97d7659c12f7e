#!/bin/bash
# round 4, eighth GPU session: BPTT on the matrix cores (gru_bwd_quad_kernel, SIR_BPTT=5) -- parity, A/B against the four-k FMA kernel,
# knock-outs, poll delay; fork-server DataLoader route through train() and the bench's dropin_epoch block
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R
SIR_BPTT=5 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_robustness_gpu.py tests/test_nccl_gpu.py -x -q -m gpu > $O/tests_bq.log 2>&1 || { tail -40 $O/tests_bq.log; exit 1; }
tail -2 $O/tests_bq.log
for m in 4 5 4 5; do
  SIR_BPTT=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag bptt$m --kernels bwd_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_bptt_quad.jsonl
done
for k in 1 2 4 6; do
  SIR_BPTT=5 SIR_BQ_DBG=$k timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag bq_dbg$k --kernels bwd_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_bptt_quad.jsonl
done
for dly in 0 4 12 16 8; do
  SIR_BPTT=5 SIR_BQ_DELAY=$dly timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag bq_delay$dly --kernels bwd_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_bptt_quad.jsonl
done
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests_pipe.log 2>&1 || { tail -40 $O/tests_pipe.log; exit 1; }
tail -2 $O/tests_pipe.log
SIR_BPTT=5 timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: infer", d["value"], "ms", d["ms_per_step"], "serial", d["single_stream"]["ms_per_step"], "train", d["train"]["value"], d["train"]["ms_per_step"], "aug", d["train_aug"]["value"])
print({k: v for k, v in d["kernels_avg_ms"].items() if "gru" in k})
de=d["train"].get("dropin_epoch", {})
for k in ("dataloader","dataloader_forkserver","hbm_feature_store","waveform_store"):
    print(k, json.dumps(de.get(k)))
PY
