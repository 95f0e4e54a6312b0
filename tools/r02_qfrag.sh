#!/bin/bash
# GPU box: wide-row kernels reading fragment-order queries (main build) vs the row-major block (libragroute_hip_q0.so,
# -DRR_WIDE_QFRAG=0): parity first, then the shape lines of both on the same device.
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_qfrag
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_flat_search_gpu.py tests/test_guard_pages_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for shape in "4000000 1024" "2000000 2048" "2000000 4096"; do
  for nq in 256 1; do
    for L in "" _q0; do
      [ -n "$L" ] && [ ! -f ragroute_amd/libragroute_hip$L.so ] && continue
      f=$O/shape_$(echo $shape | tr ' ' x)_b${nq}$L.json
      RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$L.so timeout -k 10 300 python tools/shape_bench.py $shape $nq 32 fp16 20 > $f 2> $f.err || { tail $f.err; exit 1; }
      python - "$f" "$shape b=$nq lib=${L:-main}" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[2], "ms", j["ms_per_step"], "scan frac", r["frac"], "e2e", r.get("end_to_end_frac"))
PY
    done
  done
done
