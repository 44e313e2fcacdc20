for w in 2 3 4 5; do
  echo "=== waves=$w"
  LASGUN_HIP_LIB=$PWD/lasgun_amd/liblasgun_hip_w$w.so timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|kernel_ms_avg": [0-9.]*' | tr '\n' ' '; echo
done
