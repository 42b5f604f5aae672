for p in fp32 bf16x3; do
  rm -rf gpurun_out/cli_$p
  XPS_GEMM_PRECISION=$p python scripts/train_seq2seq.py -pt SYN -p True --synthetic 3 --iters 1 --folds 3 --epochs 100 --hidden 128 --seed 3 --out gpurun_out/cli_$p > gpurun_out/cli_$p.log 2>&1
  python -c "import numpy as np; a=np.load('gpurun_out/cli_$p/accs/SYN/SYN_pooled_accs.npy'); print('$p', 'fold accuracies', np.round(a, 4).tolist(), 'mean', round(float(a.mean()), 4))"
done
