#!/bin/bash
# round 4: the consumer-side fp16 split again, this time with the fine-grained producer stamps (harness only, -DSIR_W2_CSPLIT)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4ag
mkdir -p $O
cd $R
touch devtools/kernel_ab/bench_conv.hip
make -C speech-intent-recognizer_amd/csrc tools EXTRA=-DSIR_W2_CSPLIT > $O/make_tools.log 2>&1 || { tail -20 $O/make_tools.log; exit 1; }
timeout -k 10 500 speech-intent-recognizer_amd/lib/bench_conv wino2 > $O/bench_conv_csplit.txt 2>&1 || { tail -20 $O/bench_conv_csplit.txt; exit 1; }
grep -E "^conv|f16x3 arith|f16x3 knock|f16x3 producer|stamps group|^    \[" $O/bench_conv_csplit.txt | head -24
