#!/bin/bash
for rep in 1 2 3; do
for m in 524288 0; do
    printf "XPS_SPLIT4_WEIGHTS_MIN=$m cfg2 : "
    XPS_SPLIT4_WEIGHTS_MIN=$m python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'])"
done
done
