#!/bin/bash
# Same-box A/B of argument sets of tools/bench_configs.py, variants taken in turn, round after round:
#   tools/ab_args.sh <rounds> "<args of variant 1>" "<args of variant 2>" ...   -> one JSON line per (round, variant, config)
rounds=$1; shift
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    python tools/bench_configs.py $v 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(json.dumps({'round': $r, 'variant': '''$v''', 'config': d['config'], 'ms': d['ms']}), flush=True)"
  done
done
