#!/bin/bash
# usage: scratch/ab.sh "ENV1=.." "ENV2=.." ... ; runs bench 3 rounds interleaved, prints img/s
for r in 1 2 3; do
  for e in "$@"; do
    v=$(env $e python bench.py --steps 40 --warmup 4 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('hipgraph'))")
    echo "$e -> $v"
  done
done
