export XPS_GEMM_PRECISION=bf16x3
for st in 0 1024 2048 8192; do for tn in 384 768 1536; do
  r=$(XPS_GEMM_SMALL_TILE_BLOCKS=$st XPS_TN_BLOCKS=$tn python bench.py --no-cpu-baseline --steps 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "small_tile_blocks=$st tn_blocks=$tn $r"
done; done
