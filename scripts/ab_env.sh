#!/bin/bash
# same-box A/B of an environment switch: scripts/ab_env.sh VAR=VALUE [bench.py args...]  (base = with the switch set)
sw=$1; shift
for i in 1 2; do
  env "$sw" python bench.py "$@" --no-cpu-baseline --no-legs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['value'], d['ms_per_step'])"
  python bench.py "$@" --no-cpu-baseline --no-legs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'])"
done
