python -m pytest tests/test_gpu_nn_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do r=$(python bench.py --no-cpu-baseline --steps 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); echo "bf16x3 $r"; done
r=$(python bench.py --precision fp32 --no-cpu-baseline --steps 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); echo "fp32 $r"
python tools/bench_gemm.py 2>/dev/null | head -10
