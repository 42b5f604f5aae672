#!/bin/bash
# Round-3 evidence in one GPU call: default bench line, rocprofv3 kernel stats (serial streams) + step timelines of both
# workloads, PMC traffic of the cluster kernels and the configs[1] weight-gradient group.  Outputs under gpurun_out/r3/.
# usage: tools/round3_evidence.sh [tag]
R=$PWD; TAG=${1:-r3}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() {  # name, env assignments, bench args
  local name=$1 envs=$2; shift 2
  rm -rf /tmp/p_$name
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o x -- python3 $R/bench.py "$@" > /tmp/p_$name.log 2>&1
  cp $(find /tmp/p_$name -name '*kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv
  cp $(find /tmp/p_$name -name '*kernel_trace.csv' | head -1) /tmp/${name}_trace.csv
  tail -1 /tmp/p_$name.log | cut -c1-200
}
pmc() {  # name, counter, env, script
  rm -rf /tmp/c_$1
  env $3 rocprofv3 --pmc $2 --output-format csv -d /tmp/c_$1 -o x -- python3 $R/$4 > /tmp/c_$1.log 2>&1
  cp $(find /tmp/c_$1 -name '*counter_collection.csv' | head -1) /tmp/c_$1.csv
}
prof h512_serial XPS_OVERLAP_WGRAD=0 --workload configs3 --steps 10 --warmup 3 --headline-only --no-cpu-baseline &&
prof cfg2_serial XPS_OVERLAP_WGRAD=0 --workload configs1 --steps 20 --warmup 5 --headline-only --no-cpu-baseline &&
prof h512_overlap XPS_OVERLAP_WGRAD=1 --workload configs3 --steps 10 --warmup 3 --headline-only --no-cpu-baseline &&
prof cfg2_overlap XPS_OVERLAP_WGRAD=1 --workload configs1 --steps 20 --warmup 5 --headline-only --no-cpu-baseline &&
python3 $R/tools/timeline.py /tmp/cfg2_overlap_trace.csv > $O/cfg2_step_timeline.txt &&
python3 $R/tools/timeline.py /tmp/h512_overlap_trace.csv > $O/h512_step_timeline.txt &&
python3 $R/tools/prof_summary.py $O/h512_serial_kernel_stats.csv 163 "configs[3] shard, serial streams (XPS_OVERLAP_WGRAD=0), 150 pre-warm + 3 warm-up + 10 timed steps" > $O/h512_serial_summary.md &&
python3 $R/tools/prof_summary.py $O/cfg2_serial_kernel_stats.csv 425 "configs[1], serial streams (XPS_OVERLAP_WGRAD=0), 400 pre-warm + 5 warm-up + 20 timed steps" > $O/cfg2_serial_summary.md &&
pmc gru_f FETCH_SIZE BWD=1 tools/run_gru_fwd.py && pmc gru_w WRITE_SIZE BWD=1 tools/run_gru_fwd.py &&
pmc tn_f FETCH_SIZE X=1 tools/run_wgrad_group.py && pmc tn_w WRITE_SIZE X=1 tools/run_wgrad_group.py &&
python3 $R/tools/pmc_traffic.py $O/pmc_traffic.json \
  gru_cluster_fwd_kernel_bf16x3=gru_cluster_fwd_kernel:1342177280:/tmp/c_gru_f.csv:/tmp/c_gru_w.csv \
  gru_cluster_bwd_kernel_bf16x3=_bwd_kernel:1677721600:/tmp/c_gru_f.csv:/tmp/c_gru_w.csv \
  gemm_tn_grouped_kernel_bf16x3=gemm_tn_grouped_kernel:252844032:/tmp/c_tn_f.csv:/tmp/c_tn_w.csv:+gemm_tn_grouped_reduce > $O/pmc.log 2>&1
tail -3 $O/h512_serial_summary.md; tail -1 $O/cfg2_step_timeline.txt; tail -1 $O/h512_step_timeline.txt; tail -25 $O/pmc.log
