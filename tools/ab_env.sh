#!/bin/bash
# usage: ab_env.sh VAR v1 v2 ... : bench headline (cfg 2) under each value of an environment switch, three interleaved repeats
VAR=$1; shift
for rep in 1 2 3; do
for v in "$@"; do
    printf "$VAR=$v cfg2 : "
    env $VAR=$v python bench.py --headline-only --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'], 'tn', d['roofline']['launch_us'])"
done
done
