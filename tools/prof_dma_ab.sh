#!/bin/bash
# per-kernel A/B of the configs[3] shard: rocprofv3 --kernel-trace --stats with XPS_GEMM_DMA=0 / 1 (serial streams), then the
# overlapped step time of both.  usage (GPU box): bash tools/prof_dma_ab.sh <outdir under gpurun_out>
set -e
R=$PWD
O=$R/gpurun_out/${1:-dma_ab}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in 0 1; do
  export XPS_GEMM_DMA=$s XPS_OVERLAP_WGRAD=0 XPS_BENCH_PREWARM_STEPS=20
  rm -rf /tmp/prof_$s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$s -o h512 -- python3 $R/bench.py --steps 10 --warmup 3 --headline-only --no-cpu-baseline > /tmp/prof_$s.log 2>&1
  f=$(find /tmp/prof_$s -name '*kernel_stats.csv' | head -1)
  cp $f $O/dma_${s}_kernel_stats.csv
  python3 $R/tools/prof_summary.py $f 33 "configs[3] shard, serial streams, XPS_GEMM_DMA=$s" > $O/dma_${s}_summary.md
  tail -1 /tmp/prof_$s.log | cut -c1-160
done
unset XPS_OVERLAP_WGRAD XPS_BENCH_PREWARM_STEPS
for s in 0 1 0 1; do
  XPS_GEMM_DMA=$s python3 $R/bench.py --steps 30 --warmup 5 --headline-only --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('overlapped step, XPS_GEMM_DMA=$s:', d['ms_per_step'], 'ms')"
done
