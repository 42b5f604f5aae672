#!/bin/bash
for rep in 1 2 3 4 5 6; do
for v in 0 4; do
    printf "XPS_BENCH_PIN_CORES=$v : "
    XPS_BENCH_PIN_CORES=$v python bench.py --gpus 1 --steps 20 --warmup 5 --headline-only --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'])"
done
done
