#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_fuzz_final; mkdir -p $O
rc=0
timeout -k 10 330 python tools/fuzz_parity.py 811 800 segments > $O/segments.log 2>&1 || rc=1; tail -1 $O/segments.log | cut -c1-200
timeout -k 10 200 python tools/fuzz_parity.py 812 1200 > $O/plain.log 2>&1 || rc=1; tail -1 $O/plain.log | cut -c1-200
timeout -k 10 330 python tools/fuzz_parity.py 813 150 big > $O/big.log 2>&1 || rc=1; tail -1 $O/big.log | cut -c1-200
exit $rc
