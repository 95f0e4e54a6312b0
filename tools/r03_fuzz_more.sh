#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_fuzz_more; mkdir -p $O
rc=0
timeout -k 10 500 python tools/fuzz_parity.py $1 1400 segments > $O/segments_$1.log 2>&1 || rc=1; tail -1 $O/segments_$1.log | cut -c1-200
timeout -k 10 300 python tools/fuzz_parity.py $2 2500 > $O/plain_$2.log 2>&1 || rc=1; tail -1 $O/plain_$2.log | cut -c1-200
exit $rc
