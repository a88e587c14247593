#!/bin/bash
# usage: ab/ops_ab.sh "<ops>" lib1 lib2 ... ("tree" = in-tree lib)
ops=$1; shift
for l in "$@"; do
  if [ $l = tree ]; then unset TCL_LIB_PATH; else export TCL_LIB_PATH=$PWD/ab/lib_$l.so; fi
  echo "== $l"; timeout -k 5 100 python3 bench_ops.py --ops $ops | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  %-70s %8.4f ms  %7.1f %s  frac %.3f' % (r['op'][:70], r['ms'], r['achieved'], r['unit'], r['frac']))" || exit 1
done
