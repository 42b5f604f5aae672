#!/bin/bash
# tools/bench_proj.py under each library build given as argument ("" = the shipped one)
for lib in "" "$@"; do
  echo "== ${lib:-shipped}"
  XPS_LIB_OVERRIDE=${lib:+$PWD/$lib} python tools/bench_proj.py 2>&1 | grep "^M="
done
