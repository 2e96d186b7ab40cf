cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05A; mkdir -p $O; rm -f $O/pmc_config5_step.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  NOWARM=1 STEPS=4 NOISE=f64 rocprofv3 --pmc $set -d /tmp/pc5_$tag -o k --output-format csv -- python3 $R/tools/config5.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pc5_$tag k_pcn_mm >> $O/pmc_config5_step.txt 2>&1
done
