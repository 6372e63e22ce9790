#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for n in 3 1000 10000 30000 100000 300000; do
  for ser in 0 1; do
    PF_GRAPH_SERIAL=$ser python bench.py --elems $n --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('n=$n serial=$ser ms/iter', round(d['ms_per_step'],4))"
  done
done
