# usage: bash tools/ab_libs.sh name1 name2 ...   (A/B of lasgun_amd/liblasgun_hip_<name>.so on the bench; 60 s cap per run)
for n in "$@"; do
  echo "=== $n"
  LASGUN_HIP_LIB=$PWD/lasgun_amd/liblasgun_hip_$n.so timeout -k 5 60 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|kernel_ms_avg": [0-9.]*' | tr '\n' ' '; echo
done
