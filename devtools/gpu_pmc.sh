#!/bin/bash
# PMC passes (separate from kernel-trace): MFMA busy, wave stalls, HBM traffic.  Run via gpurun.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc/$tag -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --streams 1 --repeats 1 > $R/gpurun_out/pmc/$tag.log 2>&1 || { echo "pmc group failed: $grp"; tail -5 $R/gpurun_out/pmc/$tag.log; }
  echo "done $grp"
done
# HBM traffic of the training-only kernels (three steps)
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc/train_$grp -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --repeats 1 --train-steps 3 --no-cpu-baseline --no-augment --no-host-feed --streams 1 > $R/gpurun_out/pmc/train_$grp.log 2>&1 || { echo "pmc training pass failed: $grp"; tail -5 $R/gpurun_out/pmc/train_$grp.log; }
  echo "done training $grp"
done
ls -R $R/gpurun_out/pmc | head -40
