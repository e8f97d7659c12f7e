#!/bin/bash
# A/B of the GRU backward GEMMs inside the training step: producer / consumer kernel (SIR_TN2 = 7, the default) against the first
# kernel (SIR_TN2 = 0); per-kernel durations from rocprofv3's kernel trace (run via gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in ${MASKS:-0 7}; do
SIR_TN2=$m timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tn2 -o tn2_m$m -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --train-steps 20 --no-cpu-baseline --no-dist-leg --no-dropin --no-augment --no-host-feed --sustain-seconds 0 --streams 1 > $R/gpurun_out/tn2_m$m.log 2>&1 || { tail -20 $R/gpurun_out/tn2_m$m.log; exit 1; }
echo "SIR_TN2=$m"; python3 -c "
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'gemm_tn' in r[0]: print('   ', r[0][:44], r[1], 'avg', round(float(r[3])/1000,1), 'min', round(float(r[5])/1000,1), 'max', round(float(r[6])/1000,1))
" $R/gpurun_out/prof_tn2/tn2_m${m}_kernel_stats.csv
done
