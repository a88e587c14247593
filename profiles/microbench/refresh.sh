#!/bin/bash
# end-of-round refresh: tests, smoke, bench line, kernel stats and PMC byte passes for bench.py and bench_ops.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rf
(while sleep 30; do echo "[hb]"; done) & HB=$!
trap "kill $HB" EXIT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/rf/tests.log 2>&1 || { tail -30 gpurun_out/rf/tests.log; exit 1; }
tail -2 gpurun_out/rf/tests.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/rf/smoke.log 2>&1 || { tail gpurun_out/rf/smoke.log; exit 1; }
tail -1 gpurun_out/rf/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/rf/bench.json 2> gpurun_out/rf/bench.err || { tail gpurun_out/rf/bench.err; exit 1; }
cat gpurun_out/rf/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rf/c2 -o p -- python3 bench.py --no-cpu-baseline --no-variants > gpurun_out/rf/c2.log 2>&1 || { tail -5 gpurun_out/rf/c2.log; exit 1; }
python3 ab/stats.py gpurun_out/rf/c2 | head -8
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rf/ops -o p -- python3 bench_ops.py --ops warp,affine,ctx,corr_model,argmax,misc > gpurun_out/rf/ops.log 2>&1 || { tail -5 gpurun_out/rf/ops.log; exit 1; }
grep '^{' gpurun_out/rf/ops.log > gpurun_out/rf/ops.jsonl
python3 ab/stats.py gpurun_out/rf/ops | head -40
timeout -k 10 200 python bench_ops.py --ops movement,sched > gpurun_out/rf/n34.log 2>&1 || { tail -5 gpurun_out/rf/n34.log; exit 1; }
grep '^{' gpurun_out/rf/n34.log > gpurun_out/rf/n34.jsonl
bash ab/pmc_bytes.sh rf/c2pmc bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/rf/c2pmc.txt 2>&1 || { tail gpurun_out/rf/c2pmc.txt; exit 1; }
bash ab/pmc_bytes.sh rf/opspmc bench_ops.py --iters 2 --ops warp,affine,ctx,corr_model,argmax,misc > gpurun_out/rf/opspmc.txt 2>&1 || { tail gpurun_out/rf/opspmc.txt; exit 1; }
echo refresh done
