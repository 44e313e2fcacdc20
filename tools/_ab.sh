set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
timeout -k 10 200 python tools/size_sweep.py glass 640,768,896,1024,1280,1536 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,$5}'
timeout -k 10 200 python tools/size_sweep.py simple2 512,1024,1536,2048 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,$5}'
timeout -k 10 200 python tools/size_sweep.py simple1 256,512,1024,2048 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,$5}'
timeout -k 10 200 python tools/size_sweep.py plastic 256,512,1024 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,$5}'
