#!/bin/bash
# confirmation at the round's last commit: full GPU suite, smoke, the default bench line with its wall time
set -o pipefail
O=gpurun_out/r4am; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -3 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert " $O/gpu_tests.log | head -40; exit 1; fi
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
t0=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
t1=$(date +%s)
echo "default bench.py wall: $((t1 - t0)) s"
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: infer", d["value"], "ms", d["ms_per_step"], "serial", d["single_stream"]["ms_per_step"], "train", d["train"]["value"], d["train"]["ms_per_step"], "steps", d["steps"], "warmup", d["warmup"])
print(json.dumps(d["roofline"])[:600])
PY
