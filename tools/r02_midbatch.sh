#!/bin/bash
# GPU box: wide-row searches at batch sizes between 1 and 256 (which waves hold queries decides the MFMA spread)
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_midbatch
mkdir -p $O
for shape in "2000000 4096" "2000000 2048" "4000000 1024" "10000000 768"; do
  for nq in ${NQS:-1 16 17 32 64 65 128 129 192 256}; do
    f=$O/shape_$(echo $shape | tr ' ' x)_b${nq}.json
    timeout -k 10 300 python tools/shape_bench.py $shape $nq 32 fp16 10 > $f 2> $f.err || { tail $f.err; exit 1; }
    python - "$f" "$shape b=$nq" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[2], "median_ms", j.get("median_ms"), "scan frac", r["frac"], "kernel", r.get("kernel"))
PY
  done
done
