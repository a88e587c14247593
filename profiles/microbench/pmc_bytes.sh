#!/bin/bash
# usage: profiles/microbench/pmc_bytes.sh <name> <python script + args...>  -> two PMC passes (FETCH_SIZE, WRITE_SIZE), per-kernel averages in KiB
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${name}_$c -o p -- python3 "$@" > gpurun_out/${name}_$c.log 2>&1 || { tail -5 gpurun_out/${name}_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob('gpurun_out/${name}_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c: agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if 'tcl::' in k: res[k][c + '_kib_avg'] = sum(v) / len(v); res[k]['launches'] = len(v)
json.dump(res, open('gpurun_out/${name}_bytes.json', 'w'), indent=1)
for k, v in res.items(): print(k[:90].ljust(90), {a: round(b, 1) for a, b in v.items()})
PY
