#!/bin/bash
# End-of-round evidence in one GPU call: default bench line, rocprofv3 kernel stats (serial streams) of both workloads, step
# timelines with the side stream.  Outputs under gpurun_out/final/.
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
prof() {  # name, env assignments, bench args
  local name=$1 envs=$2; shift 2
  rm -rf /tmp/p_$name
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o x -- python3 $R/bench.py "$@" > /tmp/p_$name.log 2>&1
  cp $(find /tmp/p_$name -name '*kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv
  cp $(find /tmp/p_$name -name '*kernel_trace.csv' | head -1) /tmp/${name}_trace.csv
  tail -1 /tmp/p_$name.log | cut -c1-160
}
prof cfg2_serial XPS_OVERLAP_WGRAD=0 --steps 20 --warmup 5 --headline-only --no-cpu-baseline
prof h512_serial XPS_OVERLAP_WGRAD=0 --hidden 512 --channels 30 --steps 10 --warmup 3 --headline-only --no-cpu-baseline
prof cfg2_overlap XPS_OVERLAP_WGRAD=1 --steps 20 --warmup 5 --headline-only --no-cpu-baseline
prof h512_overlap XPS_OVERLAP_WGRAD=1 --hidden 512 --channels 30 --steps 10 --warmup 3 --headline-only --no-cpu-baseline
python3 $R/tools/timeline.py /tmp/cfg2_overlap_trace.csv > $O/cfg2_step_timeline.txt
python3 $R/tools/timeline.py /tmp/h512_overlap_trace.csv > $O/h512_step_timeline.txt
tail -1 $O/cfg2_step_timeline.txt; tail -1 $O/h512_step_timeline.txt
