#!/bin/bash
# A/B on one box: XPS_SPLIT4=0 (fp32 operands everywhere) vs 1 (pre-split dgi / dghn / layer inputs / weights where supported)
for rep in 1 2; do
for s in 0 1; do
  for cfg in "" "--hidden 512 --channels 30"; do
    printf "XPS_SPLIT4=$s cfg=${cfg:-cfg2} : "
    XPS_SPLIT4=$s python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'], 'roofline_us', d.get('roofline',{}).get('launch_us'))"
  done
done
done
