set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
LASGUN_HIP_LIB=$PWD/lasgun_amd/liblasgun_hip_stamps.so timeout -k 10 200 python tools/stamp_mesh.py metal
LASGUN_HIP_LIB=$PWD/lasgun_amd/liblasgun_hip_stamps.so timeout -k 10 200 python tools/stamp_mesh.py glass
