#!/bin/bash
# per-kernel A/B of the configs[3] shard: rocprofv3 --kernel-trace --stats with XPS_SPLIT4=0 / 1 (serial streams)
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for s in 0 1; do
  export XPS_SPLIT4=$s XPS_OVERLAP_WGRAD=0
  rm -rf /tmp/prof_$s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$s -o h512 -- python3 $R/bench.py --hidden 512 --channels 30 --steps 10 --warmup 3 --headline-only --no-cpu-baseline > /tmp/prof_$s.log 2>&1
  f=$(find /tmp/prof_$s -name '*kernel_stats.csv' | head -1)
  cp $f $R/gpurun_out/h512_split4_${s}_kernel_stats.csv
  tail -1 /tmp/prof_$s.log | cut -c1-200
done
