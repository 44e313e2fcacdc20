#!/bin/bash
# Same-box A/B of the HIP-graph replay of small level-by-level frames: tools/ab_small_frames.sh <rounds> [frame filters] -> JSON lines (variants in turn)
rounds=${1:-3}; shift
for r in $(seq 1 "$rounds"); do
  for g in 0 1; do
    LASGUN_GRAPH=$g python tools/small_frames.py "$@" 2>/dev/null | sed "s/^{/{\"round\": $r, /"
  done
done
