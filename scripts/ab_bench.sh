#!/bin/bash
# same-box A/B of two builds of the library: scripts/ab_bench.sh <base.so> [bench.py args...]  (new = the in-tree build)
base=$1; shift
for i in 1 2; do
  DT_HIP_LIB=$base python bench.py "$@" --no-cpu-baseline --no-legs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['value'], d['ms_per_step'])"
  python bench.py "$@" --no-cpu-baseline --no-legs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'])"
done
