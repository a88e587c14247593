#!/bin/bash
# usage: ab/prof.sh <name> <python script + args...>   -> gpurun_out/<name>/ csv + printed stats
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -o p -- python3 "$@" > gpurun_out/$name.log 2>&1 || { tail -5 gpurun_out/$name.log; exit 1; }
grep -v "simple_timer\|rocprofv3\|generateRocpd\|tool.cpp" gpurun_out/$name.log | tail -15
python3 profiles/microbench/stats.py gpurun_out/$name
