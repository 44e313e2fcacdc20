#!/bin/bash
# Same-box A/B of environment switches on configs of tools/bench_configs.py, variants taken in turn, round after round:
#   tools/ab_env.sh "<bench_configs arguments>" <rounds> "NAME=VAL NAME=VAL" "NAME=VAL ..." ...    ("-" = no switch)
# -> one line per (round, variant, config): variant | config | ms
args=$1; rounds=$2; shift 2
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
    env $envs python tools/bench_configs.py $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v |', d['config'][:10], '|', d['ms'])"
  done
done
