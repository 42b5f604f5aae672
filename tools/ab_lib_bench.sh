#!/bin/bash
# headline step under the shipped library and under an A/B build (argument), XPS_SPLIT4=0 in both, interleaved
for rep in 1 2 3 4; do
for lib in "" "$1"; do
    printf "lib=${lib:-shipped} : "
    XPS_SPLIT4=0 XPS_LIB_OVERRIDE=${lib:+$PWD/$lib} python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'], 'tn launch', d['roofline']['launch_us'])"
done
done
