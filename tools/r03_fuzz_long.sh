#!/bin/bash
# GPU box: long seeded fuzz on the final code (progress lines keep the call alive)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_fuzz_long; mkdir -p $O
rc=0
timeout -k 10 420 python tools/fuzz_parity.py 411 1100 segments > $O/segments_a.log 2>&1 || rc=1; tail -1 $O/segments_a.log | cut -c1-200
timeout -k 10 420 python tools/fuzz_parity.py 412 1100 segments > $O/segments_b.log 2>&1 || rc=1; tail -1 $O/segments_b.log | cut -c1-200
timeout -k 10 300 python tools/fuzz_parity.py 413 2000 > $O/plain.log 2>&1 || rc=1; tail -1 $O/plain.log | cut -c1-200
timeout -k 10 420 python tools/fuzz_parity.py 414 300 big > $O/big.log 2>&1 || rc=1; tail -1 $O/big.log | cut -c1-200
timeout -k 10 120 python tools/r03_ws_poison.py > $O/ws_poison.log 2>&1 || rc=1; tail -1 $O/ws_poison.log
exit $rc
