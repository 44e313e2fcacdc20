set -u
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
O=gpurun_out/r4c10; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python tools/bench_configs.py --progressive=100 "3 sph" "2G" "2P" "1a" > $O/progressive.jsonl 2>$O/progressive.err; cat $O/progressive.jsonl
timeout -k 10 300 python tools/bench_configs.py --progressive=7 "3 sph" "2G" >> $O/progressive.jsonl 2>$O/progressive.err; tail -2 $O/progressive.jsonl
