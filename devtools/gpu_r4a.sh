#!/bin/bash
# round 4, first GPU session: (1) f16x3 GEMM A/B on random and on the real projection operands, (2) BPTT variants and knock-outs,
# (3) two-stream backward A/B -- all inside ONE box session (run via gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_robustness_gpu.py -x -q -m gpu > $O/tests_default.log 2>&1 || { tail -30 $O/tests_default.log; exit 1; }
tail -2 $O/tests_default.log
SIR_BWD_STREAMS=1 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_nccl_gpu.py -x -q -m gpu > $O/tests_streams.log 2>&1 || { tail -30 $O/tests_streams.log; exit 1; }
tail -2 $O/tests_streams.log
SIR_BPTT=2 timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/tests_pk.log 2>&1 || { tail -30 $O/tests_pk.log; exit 1; }
tail -2 $O/tests_pk.log
timeout -k 10 300 python devtools/dump_gemm_operands.py $O/ops > $O/ops.txt 2>&1 || { tail -20 $O/ops.txt; exit 1; }
cat $O/ops.txt
B=speech-intent-recognizer_amd/lib/bench_gemm
timeout -k 10 200 $B 6400 1024 > $O/gemm_random_k1024.txt 2>&1 || { tail -20 $O/gemm_random_k1024.txt; exit 1; }
timeout -k 10 200 $B 6400 1024 $O/ops/A_l0.f32 $O/ops/B_l0.f32 $O/ops/bias_l0.f32 > $O/gemm_real_l0.txt 2>&1 || { tail -20 $O/gemm_real_l0.txt; exit 1; }
timeout -k 10 200 $B 6400 512 $O/ops/A_l1.f32 $O/ops/B_l1.f32 $O/ops/bias_l1.f32 > $O/gemm_real_l1.txt 2>&1 || { tail -20 $O/gemm_real_l1.txt; exit 1; }
rm -rf $O/ops/*.f32
tail -14 $O/gemm_real_l0.txt; tail -14 $O/gemm_real_l1.txt
for m in 0 1 2 3 0 1 17 18 19 20 24; do
  SIR_BPTT=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 3 --tag bptt$m --kernels bwd_gru_l > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_bptt.jsonl
done
for m in 0 1 0 1; do
  SIR_BWD_STREAMS=$m timeout -k 10 200 python devtools/train_only.py --steps 20 --repeats 5 --tag streams$m --kernels bwd_ > $O/tmp.json 2> $O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  tail -1 $O/tmp.json | tee -a $O/ab_streams.jsonl
done
