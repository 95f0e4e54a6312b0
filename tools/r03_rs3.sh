#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_rs3; mkdir -p $O
for shape in "2000000 4096" "1000000 8192" "2000000 3072"; do
  for rs in 0 1 0 1; do
    tag=$(echo $shape | tr ' ' x)
    RR_WIDE_RS_MAXD=8192 RR_WIDE_RS=$rs python tools/shape_bench.py $shape 256 10 fp16 12 > $O/shape_${tag}_rs$rs.json 2>/dev/null
    python -c "import json; j=json.load(open('$O/shape_${tag}_rs$rs.json')); print('$shape rs=$rs scan frac', j['roofline']['frac'], 'b2b_ms', j['back_to_back_ms'], 'e2e frac', j['back_to_back_frac_of_8TBps'], j['sanity_top1'])"
  done
done
for nq in 160 192 208; do
  for rs in 0 1; do
    RR_WIDE_RS_MINQ=129 RR_WIDE_RS=$rs python tools/shape_bench.py 4000000 1024 $nq 10 fp16 12 > $O/q${nq}_rs$rs.json 2>/dev/null
    python -c "import json; j=json.load(open('$O/q${nq}_rs$rs.json')); print('4M x 1024 nq=$nq rs=$rs scan frac', j['roofline']['frac'], 'b2b_ms', j['back_to_back_ms'], j['sanity_top1'])"
  done
done
