#!/bin/bash
# GPU box: one query and 256 queries over every row-width class (corpus of ~4 GB each): looking for widths that fall off the curve
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_dimsweep; mkdir -p $O
for d in ${DIMS:-128 256 384 512 640 768 896 1024 1152 1280 1536 1664 1792 2048 2304 2560 3072 3584 4096 5120 6144 8192}; do
  n=$(( 2000000000 / d )); [ $n -gt 10000000 ] && n=10000000
  for nq in 1 256; do
    f=$O/shape_${d}_b${nq}.json
    timeout -k 10 200 python tools/shape_bench.py $n $d $nq ${K:-32} ${DT:-fp16} 10 > $f 2> $f.err || { echo "d=$d nq=$nq FAILED: $(tail -1 $f.err)"; continue; }
    python - "$f" "d=$d n=$n b=$nq" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "scan frac", j["roofline"]["frac"], "b2b frac", j["back_to_back_frac_of_8TBps"], "b2b_ms", j["back_to_back_ms"])
PY
  done
done
