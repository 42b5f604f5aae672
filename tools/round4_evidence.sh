#!/bin/bash
# Round-4 evidence in one GPU call: rocprofv3 kernel stats (serial streams) of the configs[3] shard in bf16x3 AND fp32 mode and of
# configs[1], step timelines with the side stream, PMC traffic of the cluster kernels in both precisions and of the configs[1]
# weight-gradient group, SQ counters of the LDS-DMA weight-gradient kernel.  Outputs under gpurun_out/<tag>/.
# usage: tools/round4_evidence.sh [tag] [commit]
R=$PWD; TAG=${1:-r4}; COMMIT=${2:-unknown}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export XPS_BENCH_PREWARM_STEPS=20
prof() {  # name, env assignments, bench args
  local name=$1 envs=$2; shift 2
  rm -rf /tmp/p_$name
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o x -- python3 $R/bench.py "$@" > /tmp/p_$name.log 2>&1
  cp $(find /tmp/p_$name -name '*kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv
  cp $(find /tmp/p_$name -name '*kernel_trace.csv' | head -1) /tmp/${name}_trace.csv
  tail -1 /tmp/p_$name.log | cut -c1-200
}
pmc() {  # name, counters, env, script
  rm -rf /tmp/c_$1
  env $3 rocprofv3 --pmc $2 --output-format csv -d /tmp/c_$1 -o x -- python3 $R/$4 > /tmp/c_$1.log 2>&1
  cp $(find /tmp/c_$1 -name '*counter_collection.csv' | head -1) /tmp/c_$1.csv
}
prof h512_serial XPS_OVERLAP_WGRAD=0 --workload configs3 --steps 10 --warmup 3 --headline-only --no-probes --no-cpu-baseline &&
prof h512_fp32_serial XPS_OVERLAP_WGRAD=0 --workload configs3 --precision fp32 --steps 10 --warmup 3 --headline-only --no-probes --no-cpu-baseline &&
prof cfg2_serial XPS_OVERLAP_WGRAD=0 --workload configs1 --steps 20 --warmup 5 --headline-only --no-probes --no-cpu-baseline &&
prof h512_overlap XPS_OVERLAP_WGRAD=1 --workload configs3 --steps 10 --warmup 3 --headline-only --no-probes --no-cpu-baseline &&
python3 $R/tools/timeline.py /tmp/h512_overlap_trace.csv > $O/h512_step_timeline.txt &&
python3 $R/tools/prof_summary.py $O/h512_serial_kernel_stats.csv 33 "configs[3] shard, bf16x3, serial streams (XPS_OVERLAP_WGRAD=0), 20 pre-warm + 3 warm-up + 10 timed steps" > $O/h512_serial_summary.md &&
python3 $R/tools/prof_summary.py $O/h512_fp32_serial_kernel_stats.csv 33 "configs[3] shard, fp32 MFMA mode (--precision fp32), serial streams, 20 pre-warm + 3 warm-up + 10 timed steps" > $O/h512_fp32_serial_summary.md &&
python3 $R/tools/prof_summary.py $O/cfg2_serial_kernel_stats.csv 45 "configs[1], serial streams (XPS_OVERLAP_WGRAD=0), 20 pre-warm + 5 warm-up + 20 timed steps" > $O/cfg2_serial_summary.md &&
pmc gru_f FETCH_SIZE BWD=1 tools/run_gru_fwd.py && pmc gru_w WRITE_SIZE BWD=1 tools/run_gru_fwd.py &&
pmc gru32_f FETCH_SIZE "BWD=1 XPS_GEMM_PRECISION=fp32" tools/run_gru_fwd.py && pmc gru32_w WRITE_SIZE "BWD=1 XPS_GEMM_PRECISION=fp32" tools/run_gru_fwd.py &&
pmc tn_f FETCH_SIZE X=1 tools/run_wgrad_group.py && pmc tn_w WRITE_SIZE X=1 tools/run_wgrad_group.py &&
python3 $R/tools/pmc_traffic.py $O/pmc_traffic.json \
  gru_cluster_fwd_kernel_bf16x3=gru_cluster_fwd_kernel:1342177280:/tmp/c_gru_f.csv:/tmp/c_gru_w.csv \
  gru_cluster_bwd_kernel_bf16x3=_bwd_kernel:1677721600:/tmp/c_gru_f.csv:/tmp/c_gru_w.csv \
  gru_cluster_fwd_kernel=gru_cluster_fwd_kernel:1342177280:/tmp/c_gru32_f.csv:/tmp/c_gru32_w.csv \
  gru_cluster_bwd_kernel=_bwd_kernel:1677721600:/tmp/c_gru32_f.csv:/tmp/c_gru32_w.csv \
  gemm_tn_grouped_kernel_bf16x3=gemm_tn_grouped_kernel:252844032:/tmp/c_tn_f.csv:/tmp/c_tn_w.csv:+gemm_tn_grouped_reduce > $O/pmc.log 2>&1
python3 - <<PY
import json
p = "$O/pmc_traffic.json"
d = json.load(open(p)); d["_commit"] = "$COMMIT"
d["_how"] = d.get("_how", "").replace("round-3 kernels", "round-4 kernels") + "; keys without a precision suffix: fp32-MFMA mode (XPS_GEMM_PRECISION=fp32)"
json.dump(d, open(p, "w"), indent=1)
PY
# SQ counters of the weight-gradient kernel, LDS-DMA loop and register-staged loop (two passes each: 8 SQ slots per pass)
for m in 1 0; do
  pmc sq1_$m "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" XPS_GEMM_DMA=$m tools/run_dma_tn.py
  pmc sq2_$m "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" XPS_GEMM_DMA=$m tools/run_dma_tn.py
  { echo "## gemm_big_tn_kernel, dW_ih 1536 x 1024 x 40960 on split4 operands, XPS_GEMM_DMA=$m"; echo '```'; python3 $R/tools/pmc_table.py /tmp/c_sq1_$m.csv /tmp/c_sq2_$m.csv --kernel gemm_big_tn_kernel; echo '```'; } >> $O/pmc_gemm_big_raw.md
done
rm -rf /tmp/p_tn; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tn -o x -- python3 $R/tools/run_dma_tn.py > /tmp/p_tn.log 2>&1
grep gemm_big_tn $(find /tmp/p_tn -name '*kernel_stats.csv' | head -1) > $O/dma_tn_kernel_stats.txt
tail -3 $O/h512_serial_summary.md; tail -1 $O/h512_step_timeline.txt; tail -30 $O/pmc.log; cat $O/pmc_gemm_big_raw.md | head -60; cat $O/dma_tn_kernel_stats.txt
