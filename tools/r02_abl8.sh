#!/bin/bash
# GPU box: timing-only ablations of the 8-wave wide-row kernel (RR_WIDE8_ABL libraries _e0 baseline, _e1 half the LDS fragment
# reads, _e2 half the LDS reads + every query load twice = the resource mix of a row-split 2 x 4 wave layout).  Results are wrong
# by design; only the scan-launch time is read.
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_abl8; mkdir -p $O
for shape in "4000000 1024" "2000000 2048" "2000000 4096"; do
  for L in _e0 _e1 _e2; do
    [ -f ragroute_amd/libragroute_hip$L.so ] || continue
    f=$O/shape_$(echo $shape | tr ' ' x)$L.json
    RR_WIDE_WAVES=8 RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$L.so timeout -k 10 200 python tools/shape_bench.py $shape 256 32 fp16 20 > $f 2> $f.err || { tail -3 $f.err; continue; }
    python - "$f" "$shape lib=$L" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[2], "scan frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "b2b_ms", j["back_to_back_ms"])
PY
  done
done
