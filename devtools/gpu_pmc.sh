#!/bin/bash
# PMC passes (separate from kernel-trace): MFMA busy, wave stalls, LDS conflicts, HBM traffic -- for the inference AND the
# training kernels (one bench process per counter group: 3 serial inference steps + 3 training steps).  Run via gpurun;
# then  python profiles/pmc_summary.py gpurun_out/pmc profiles/rNN/pmc  and  python profiles/pmc_to_traffic.py gpurun_out/pmc profiles/rNN
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $R/gpurun_out/pmc
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --repeats 1 --train-steps 3 --no-cpu-baseline --no-augment --no-host-feed --no-dist-leg --no-dropin --sustain-seconds 0 --streams 1"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  # the program itself goes directly after `--` (no env / bash -c hop: the profiler has initialised the GPU by then)
  timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc/$tag -o pmc -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc/$tag.log 2>&1 || { echo "pmc group failed: $grp"; tail -5 $R/gpurun_out/pmc/$tag.log; }
  echo "done $grp"
done
ls -R $R/gpurun_out/pmc | head -40
