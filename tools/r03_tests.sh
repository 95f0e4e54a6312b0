#!/bin/bash
# GPU box: new round-3 tests first (fast feedback), then the full parity suite, smoke, bench N=1, and the 2-rank gloo rehearsals
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_tests
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_segments_gpu.py tests/test_router_merge_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; rc=$?
tail -25 $O/pytest_new.log
[ $rc -ne 0 ] && exit $rc
if [ "$1" != "quick" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_segments_gpu.py --deselect tests/test_router_merge_gpu.py > $O/pytest.log 2>&1; rc=$?
tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
fi
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
cat $O/bench_n1.json
RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --rows 2000000 --sustained-seconds 0 > $O/bench_w2.json 2> $O/bench_w2.err || { tail $O/bench_w2.err; exit 1; }
cat $O/bench_w2.json
RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --rows 1000000 --scaling strong --sustained-seconds 0 > $O/bench_s2.json 2> $O/bench_s2.err || { tail $O/bench_s2.err; exit 1; }
cat $O/bench_s2.json
