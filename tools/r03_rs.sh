#!/bin/bash
# GPU box: the row-split wide-row kernel: parity, then timings against RR_WIDE_RS=0 (the 8-wave kernel)
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_rs; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_flat_search_gpu.py -m gpu -k "wide or fixture or config" -x -q > $O/pytest.log 2>&1; rc=$?
tail -6 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for shape in "4000000 1024" "2000000 2048" "3000000 1536"; do
  for rs in 0 1 0 1; do
    tag=$(echo $shape | tr ' ' x)
    RR_WIDE_RS=$rs python tools/shape_bench.py $shape 256 10 fp16 20 > $O/shape_${tag}_rs$rs.json 2>/dev/null
    python -c "import json; j=json.load(open('$O/shape_${tag}_rs$rs.json')); print('$shape rs=$rs scan frac', j['roofline']['frac'], 'b2b_ms', j['back_to_back_ms'], 'e2e frac', j['back_to_back_frac_of_8TBps'])"
  done
done
