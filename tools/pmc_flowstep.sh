#!/bin/bash
# SQ counters of the flow-proposal step kernels (separate rocprofv3 --pmc passes, no trace domains).  FILT: kernel-name filter.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_flowstep_${TAG:-x}
mkdir -p $O; rm -f $O/summary.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d /tmp/pfs_$tag -o k --output-format csv -- python3 $R/tools/flowstep_bench.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pfs_$tag "${FILT:-k_pcn_flow_fused}" >> $O/summary.txt 2>&1
done
cat $O/summary.txt
