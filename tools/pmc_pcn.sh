#!/bin/bash
# SQ counters of a pCN step kernel: D (default 128), FILT (kernel-name filter, default k_pcn_mm), TAG (output dir suffix) (separate rocprofv3 --pmc passes, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_${TAG:-mm}
mkdir -p $O; rm -f $O/summary.txt
export D=${D:-128}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d /tmp/pmm_$tag -o k --output-format csv -- python3 $R/tools/kbench.py pcn > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmm_$tag "${FILT:-k_pcn_mm}" >> $O/summary.txt 2>&1
done
cat $O/summary.txt | tail -60
