#!/bin/bash
# Round-5 profile on the GPU box (one call): everything tools/prof_r04.sh collects for the headline kernel (kernel-trace stats of the
# driver's bench command, its JSON line, HBM traffic and SQ counters of the fused d = 32 step), plus the flow-proposal step above 32
# dimensions (k_pcn_flow16: traffic + SQ counters at d = 64, coupling), configs[4]'s step (k_pcn_mm at d = 128: SQ counters, traffic)
# and the step-by-dimension table.  Only summaries are kept under gpurun_out/<tag>/.
TAG=${1:-r05}
bash $GRAFT_REPO_ROOT/tools/prof_r04.sh $TAG > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA")
rm -f $O/pmc_flow16_d64.txt $O/pmc_config5_step.txt
for set in "${sets[@]}" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  D=64 KIND=coupling rocprofv3 --pmc $set -d /tmp/pf16_$tag -o k --output-format csv -- python3 $R/tools/flow16_bench.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pf16_$tag k_pcn_flow16 >> $O/pmc_flow16_d64.txt 2>&1
  NOWARM=1 STEPS=4 NOISE=f64 rocprofv3 --pmc $set -d /tmp/pc5_$tag -o k --output-format csv -- python3 $R/tools/config5.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pc5_$tag k_pcn_mm >> $O/pmc_config5_step.txt 2>&1
done
DIMS=8,16,20,32,48,64,100,128 python3 $R/tools/flow_dims.py 2>&1 | grep -v amdgpu.ids > $O/flow_dims.txt
NU=5 DIMS=32,64,128 python3 $R/tools/flow_dims.py 2>&1 | grep -v amdgpu.ids > $O/flow_dims_tpcn.txt
python3 $R/tools/maf_draw_time.py 2>&1 | grep -v amdgpu.ids > $O/maf_draw.txt
ls $O
