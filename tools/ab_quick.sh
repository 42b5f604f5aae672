python -m pytest tests/test_gpu_nn_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for rep in 1 2 3 4; do r=$(python bench.py --no-cpu-baseline --steps 50 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); echo "bf16x3 $r"; done
python tools/bench_gemm.py 2>/dev/null | tail -14
