#!/bin/bash
# round 4, evidence at HEAD: full GPU suite, smoke, bench line (default settings), rocprofv3 kernel traces of both legs (one stream each:
# `--streams 1` for the inference pipeline, SIR_BWD_STREAMS=0 for the backward, so that kernel durations are un-overlapped), PMC passes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4z
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -3 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert " $O/gpu_tests.log | head -40; exit 1; fi
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 900 python bench.py --steps 100 --warmup 20 > $O/bench_n1.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - $O/bench_n1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: infer", d["value"], "ms", d["ms_per_step"], "serial", d["single_stream"]["ms_per_step"], "train", d["train"]["value"], d["train"]["ms_per_step"], "aug", d["train_aug"]["value"], "rccl1", d["train"].get("rccl_world1",{}).get("ms_per_step"))
print(json.dumps(d["kernels_avg_ms"]))
print(json.dumps(d["train"]["kernels_avg_ms"]))
print(json.dumps(d["roofline"]))
de=d["train"].get("dropin_epoch", {})
for k in ("dataloader","dataloader_forkserver","hbm_feature_store","waveform_store"):
    v=de.get(k) or {}
    print(k, v.get("utts_per_s"), v.get("frac_of_step_rate"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o infer -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-train --sustain-seconds 0 --streams 1 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
SIR_BWD_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --train-steps 30 --no-cpu-baseline --no-dist-leg --no-dropin --no-host-feed --no-augment --sustain-seconds 0 --streams 1 > $O/prof_train.log 2>&1 || { tail -20 $O/prof_train.log; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train2s -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --train-steps 30 --no-cpu-baseline --no-dist-leg --no-dropin --no-host-feed --no-augment --sustain-seconds 0 --streams 1 > $O/prof_train2s.log 2>&1 || { tail -20 $O/prof_train2s.log; exit 1; }
find $O/prof -name "*kernel_stats.csv" | head
export SIR_BWD_STREAMS=0
cd $R && bash devtools/gpu_pmc.sh > $O/pmc.log 2>&1; tail -5 $O/pmc.log
