#!/bin/bash
# one GPU-box session: parity tests, smoke, bench, rocprofv3 kernel traces of the inference and the training leg (run via gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu -s > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/gpu_tests.log | head -40; fi
if [ $rc -gt 1 ]; then echo "pytest crashed rc=$rc"; exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps ${STEPS:-100} --warmup 20 > gpurun_out/bench.log 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.log
if [ "${PROF:-1}" = "1" ]; then
cd /tmp && export TMPDIR=/tmp
# inference leg alone, strictly serial (one stream) so that per-kernel averages are the kernels' own durations
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o infer -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-train --sustain-seconds 0 --streams 1 > $R/gpurun_out/prof.log 2>&1 || { tail -20 $R/gpurun_out/prof.log; exit 1; }
# training legs (plain + augmented) with a short inference part
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o train -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --train-steps 30 --no-cpu-baseline --no-dist-leg --no-dropin --sustain-seconds 0 --streams 1 > $R/gpurun_out/prof_train.log 2>&1 || { tail -20 $R/gpurun_out/prof_train.log; exit 1; }
ls -R $R/gpurun_out/prof | head -20
fi
