#!/bin/bash
# round-3 evidence refresh on the final tree: tests, smoke, the bench line, kernel stats, PMC byte passes over the cold legs, MFMA
# counters of the contractions, cold rows of bench_ops.py.  Everything lands under gpurun_out/r3final/ (copied into profiles/ by hand).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final
mkdir -p $O
(while sleep 30; do echo "[hb]"; done) & HB=$!
trap "kill $HB" EXIT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cut -c1-600 $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o p -- python3 bench.py --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2> $O/c2.err || { tail -5 $O/c2.err; exit 1; }
python3 profiles/microbench/stats.py $O/c2 | head -12
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ops -o p -- python3 bench_ops.py --cold --ops warp,affine,ctx,corr_model,argmax,misc > $O/ops.log 2>&1 || { tail -5 $O/ops.log; exit 1; }
grep '^{' $O/ops.log > $O/ops_cold.jsonl
python3 profiles/microbench/stats.py $O/ops | head -40
timeout -k 10 200 python bench_ops.py --ops movement,sched > $O/n34.log 2>&1 || { tail -5 $O/n34.log; exit 1; }
grep '^{' $O/n34.log > $O/ops_n34.jsonl
cat $O/ops_n34.jsonl
bash profiles/microbench/pmc_bytes.sh r3final/legs bench.py --legs-only corr,warp,c4 > $O/legs_bytes.txt 2>&1 || { tail $O/legs_bytes.txt; exit 1; }
tail -32 $O/legs_bytes.txt
bash profiles/microbench/pmc.sh r3final/mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" bench.py --legs-only corr > $O/mfma.txt 2>&1 || { tail $O/mfma.txt; exit 1; }
grep -A9 "gemm_" $O/mfma.txt | head -90
echo refresh done
