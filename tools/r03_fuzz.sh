#!/bin/bash
# GPU box: guard / fuzz tests of the suite, then seeded fuzz beyond the suite's bounded sample
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_fuzz
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_guard_pages_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -8 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/fuzz_parity.py 301 400 segments > $O/segments.log 2>&1; rc1=$?
tail -1 $O/segments.log | cut -c1-300
timeout -k 10 300 python tools/fuzz_parity.py 302 400 > $O/plain.log 2>&1; rc2=$?
tail -1 $O/plain.log | cut -c1-300
timeout -k 10 300 python tools/fuzz_parity.py 303 60 big > $O/big.log 2>&1; rc3=$?
tail -1 $O/big.log | cut -c1-300
exit $((rc1 + rc2 + rc3))
