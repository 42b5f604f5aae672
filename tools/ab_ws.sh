#!/bin/bash
for rep in 1 2 3; do
for v in 0 1; do
  for cfg in ""; do
    printf "XPS_PROJ_WS=$v cfg=${cfg:-cfg2} : "
    XPS_PROJ_WS=$v python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'])"
  done
done
done
