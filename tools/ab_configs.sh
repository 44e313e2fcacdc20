#!/bin/bash
# Same-box A/B of library variants on chosen configs of tools/bench_configs.py:
#   tools/ab_configs.sh "<config filters>" <rounds> <lib> [<lib> ...]     ("main" = the shipped library; others: lasgun_amd/liblasgun_hip_<lib>.so,
#   made by tools/build_variant.sh) -> one JSON line per (round, library, config) on stdout, tagged with "lib" and "round"
filters=$1; rounds=$2; shift 2
for r in $(seq 1 "$rounds"); do
  for lib in "$@"; do
    if [ "$lib" = main ]; then unset LASGUN_HIP_LIB; else export LASGUN_HIP_LIB=lasgun_amd/liblasgun_hip_$lib.so; fi
    python tools/bench_configs.py $filters 2>/dev/null | sed "s/^{/{\"lib\": \"$lib\", \"round\": $r, /"
  done
done
