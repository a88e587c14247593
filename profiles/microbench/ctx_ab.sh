#!/bin/bash
# A/B of the context-matching launch switches (TCL_CTX_XS pixel-range split, TCL_CTX_FCH channels per thread in the bwd feats pass)
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "ctx or context or smoke or cm_module" > gpurun_out/t_ctx.log 2>&1 || { tail -20 gpurun_out/t_ctx.log; exit 1; }
tail -2 gpurun_out/t_ctx.log
for cfg in "1 0" "2 0" "4 0" "8 0" "16 0" "0 0" "0 1" "0 2"; do
  set -- $cfg
  echo "XS=$1 FCH=$2"
  TCL_CTX_XS=$1 TCL_CTX_FCH=$2 timeout -k 10 120 python bench_ops.py --ops ctx 2>/dev/null | grep -o "\"op\": \"[^\"]*\", \"ms\": [0-9.]*" || exit 1
done
