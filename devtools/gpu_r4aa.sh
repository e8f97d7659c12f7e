#!/bin/bash
# round 4: where does a producer wave of the Winograd kernel spend its step?  (fine-grained s_memtime stamps, harness only)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4aa
mkdir -p $O
cd $R
make -C speech-intent-recognizer_amd/csrc tools > $O/make_tools.log 2>&1 || { tail -20 $O/make_tools.log; exit 1; }
timeout -k 10 500 speech-intent-recognizer_amd/lib/bench_conv wino2 > $O/bench_conv_wino2.txt 2>&1 || { tail -20 $O/bench_conv_wino2.txt; exit 1; }
grep -E "^conv|producer wave|^    \[" $O/bench_conv_wino2.txt | head -60
