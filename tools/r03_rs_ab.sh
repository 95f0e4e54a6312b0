#!/bin/bash
# GPU box: same-device A/B of row-split kernel variants (libraries built with RR_EXTRA_DEFINES, e.g. -DRR_RS_AHEAD=4 / 6)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_rs_ab; mkdir -p $O
for shape in "4000000 1024" "2000000 2048" "2000000 4096"; do
  for L in $LIBS $LIBS; do
    tag=$(echo $shape | tr ' ' x)
    RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$L.so python tools/shape_bench.py $shape 256 10 fp16 20 > $O/shape_${tag}$L.json 2>/dev/null
    python -c "import json; j=json.load(open('$O/shape_${tag}$L.json')); print('$shape lib=$L scan frac', j['roofline']['frac'], 'b2b_ms', j['back_to_back_ms'], j['sanity_top1'])"
  done
done
