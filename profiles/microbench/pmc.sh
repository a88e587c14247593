#!/bin/bash
# usage: ab/pmc.sh <name> "<counters>" <python script + args...>
name=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/$name -o p -- python3 "$@" > gpurun_out/$name.log 2>&1 || { tail -5 gpurun_out/$name.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/$name/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'tcl::' not in k: continue
    print(k)
    for c, x in sorted(v.items()): print('     %-40s %14.0f  (n=%d)' % (c, sum(x) / len(x), len(x)))
PY
