#!/bin/bash
# SQ wave-state and instruction counters of the cluster BPTT kernel (tools/run_gru_fwd.py BWD=1: one bidirectional H = 512 layer,
# 2048 trials, 20 steps).  One counter group per rocprofv3 pass, the program directly after `--`.  usage: tools/pmc_cluster_sq.sh out.txt [GRID]
R=$PWD; OUT=${1:-$R/gpurun_out/pmc_cluster_sq.txt}; export GRID=${2:-0}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  rm -rf /tmp/c_sq
  BWD=1 rocprofv3 --pmc $grp --output-format csv -d /tmp/c_sq -o x -- python3 $R/tools/run_gru_fwd.py > /tmp/c_sq.log 2>&1
  echo "prof rc=$?" >> $OUT
  python3 $R/tools/pmc_table.py $(find /tmp/c_sq -name '*counter_collection.csv') --kernel _bwd_kernel >> $OUT
done
cat $OUT
