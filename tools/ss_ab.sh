set -o pipefail
mkdir -p gpurun_out
o=gpurun_out/ss_par2.jsonl; : > $o
for v in "LASGUN_SS_MEGA=1" "LASGUN_SS_MEGA=0" "LASGUN_SS_SERIAL=1"; do
  env $v timeout -k 10 400 python tools/ss_probe.py 256 512 1024 >> $o 2>> gpurun_out/ss_par2.err || exit 1
done
SS_PROBE_SCENES=simple_ss3,playground_ss2,simplecows_ss2,spheres1024_ss2 LASGUN_SS_MEGA=1 timeout -k 10 300 python tools/ss_probe.py 2048 >> $o 2>> gpurun_out/ss_par2.err &&
SS_PROBE_SCENES=simple_ss3,playground_ss2,simplecows_ss2,spheres1024_ss2 LASGUN_SS_MEGA=0 timeout -k 10 300 python tools/ss_probe.py 2048 >> $o 2>> gpurun_out/ss_par2.err
