#!/bin/bash
# Round-6 profile on the GPU box (one call): everything tools/prof_r05.sh collects (driver's bench command under rocprofv3
# --kernel-trace --stats, its JSON line, HBM traffic and SQ counters of the fused d = 32 step, of the flow16 step at d = 64 and of
# configs[4]'s step, the step-by-dimension tables), plus SQ counters + traffic of the flow16 step with the reference's default flow
# class at configs[4]'s dimension (autoregressive, d = 128), where round 5 measured 1.20 x the algorithmic traffic.
TAG=${1:-r06}
bash $GRAFT_REPO_ROOT/tools/prof_r05.sh $TAG > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA")
rm -f $O/pmc_flow16_maf_d128.txt
for set in "${sets[@]}" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  D=128 KIND=maf rocprofv3 --pmc $set -d /tmp/pf16m_$tag -o k --output-format csv -- python3 $R/tools/flow16_bench.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pf16m_$tag k_pcn_flow16 >> $O/pmc_flow16_maf_d128.txt 2>&1
done
NOISE=f64 TOP=30 python3 $R/tools/config5.py 2>&1 | grep -v amdgpu.ids > $O/config5_run.txt
ls $O
