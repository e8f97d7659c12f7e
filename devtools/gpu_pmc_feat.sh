#!/bin/bash
# PMC passes over the feature kernel only (devtools/feat_only.py).  Run via gpurun.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_feat
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/devtools/feat_only.py 200
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $O/$tag -o pmc -- python3 $R/devtools/feat_only.py 5 > $O/$tag.log 2>&1 || { echo "pmc group failed: $grp"; tail -5 $O/$tag.log; }
  echo "done $grp"
done
python3 $R/profiles/pmc_summary.py $O $O/summary > /dev/null 2>&1
grep -i "feat_" $O/summary/pmc_summary.csv | head -3
head -1 $O/summary/pmc_summary.csv
