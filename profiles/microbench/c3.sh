#!/bin/bash
(while sleep 30; do echo "[hb]"; done) & HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python -m master_thesis_amd.harness_c3 --sequences 32 --bf16 > gpurun_out/c3_bf16.log 2>&1 || { tail -5 gpurun_out/c3_bf16.log; exit 1; }
tail -1 gpurun_out/c3_bf16.log
timeout -k 10 300 python -m master_thesis_amd.harness_c3 --sequences 1 --micro-batch 1 --forward-only > gpurun_out/c1.log 2>&1 || { tail -5 gpurun_out/c1.log; exit 1; }
tail -1 gpurun_out/c1.log
