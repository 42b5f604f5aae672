python -m pytest tests/test_gpu_nn_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -2
python bench.py --hidden 512 --channels 30 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
python bench.py --hidden 512 --channels 30 --no-cpu-baseline --steps 10 --warmup 3 --precision fp32 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
