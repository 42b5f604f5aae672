python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python tools/host_backward.py 2>&1 | tail -13 | head -6
for rep in 1 2 3 4; do r=$(python bench.py --no-cpu-baseline --steps 50 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); echo "bf16x3 $r"; done
