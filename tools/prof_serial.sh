#!/bin/bash
# serial-stream per-kernel profile of the configs[3] shard under the caller's environment switches; prints the top kernels.
# usage (GPU box): [VAR=..] bash tools/prof_serial.sh <outname>
R=$PWD; O=$R/gpurun_out/${1:-serial}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export XPS_OVERLAP_WGRAD=0 XPS_BENCH_PREWARM_STEPS=20
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o x -- python3 $R/bench.py --steps 10 --warmup 3 --headline-only --no-probes --no-cpu-baseline > /tmp/ps.log 2>&1
f=$(find /tmp/ps -name '*kernel_stats.csv' | head -1)
cp $f $O/kernel_stats.csv
python3 $R/tools/prof_summary.py $f 33 "configs[3] shard, serial streams" > $O/summary.md
head -${2:-24} $O/summary.md | cut -c1-150
