python -m pytest tests/test_gpu_seq2seq.py tests/test_gpu_training.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2 3; do r=$(python bench.py --no-cpu-baseline --steps 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); echo "bf16x3 $r"; done
