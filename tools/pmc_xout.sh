#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the 1-D BPTT launch with XPS_GRU_XOUT=0 / 1
R=$PWD; O=$R/gpurun_out/${1:-xout}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/x_${v}_$c
    XPS_GRU_XOUT=$v BWD=1 rocprofv3 --pmc $c --output-format csv -d /tmp/x_${v}_$c -o x -- python3 $R/tools/run_gru_fwd.py > /tmp/x_${v}_$c.log 2>&1
    cp $(find /tmp/x_${v}_$c -name '*counter_collection.csv' | head -1) /tmp/x_${v}_$c.csv
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_xout_$v.json gru_cluster_bwd_kernel_bf16x3=_bwd_kernel:1677721600:/tmp/x_${v}_FETCH_SIZE.csv:/tmp/x_${v}_WRITE_SIZE.csv > $O/pmc_$v.log 2>&1
  echo "XPS_GRU_XOUT=$v"; cat $O/pmc_xout_$v.json | grep -A6 bwd_kernel
done
