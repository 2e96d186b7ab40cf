#!/bin/bash
# SQ counters of the d = 128 step kernel (BASELINE config 5 on one GPU, tools/config5.py), one rocprofv3 --pmc pass per set.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_config5_${TAG:-x}
mkdir -p $O; rm -f $O/summary.txt
export STEPS=${STEPS:-4}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d /tmp/pc5_$tag -o k --output-format csv -- python3 $R/tools/config5.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pc5_$tag "${FILT:-k_pcn_mm}" >> $O/summary.txt 2>&1
done
cat $O/summary.txt
