#!/bin/bash
# GPU box: K rotation of the wide-row kernels (main) against none (libragroute_hip_r0.so = -DRR_WIDE_KROT=0): parity, then timing
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_krot; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_flat_search_gpu.py tests/test_fuzz_gpu.py tests/test_guard_pages_gpu.py tests/test_baseline_configs_gpu.py tests/test_router_merge_gpu.py tests/test_end_to_end_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
RR_WIDE_PD=0 timeout -k 10 300 python -m pytest tests/test_flat_search_gpu.py -x -q -m gpu -k "wide or generic or dims" > $O/pytest_pd0.log 2>&1; rc=$?
echo "round-1 kernel: $(tail -1 $O/pytest_pd0.log)"
[ $rc -ne 0 ] && exit $rc
for shape in ${SHAPES:-"2000000 2048" "2000000 4096" "1000000 8192" "2000000 1792"}; do
  for nq in ${NQS:-1 16 64 256}; do
    for L in "" _r0; do
      [ -n "$L" ] && [ ! -f ragroute_amd/libragroute_hip$L.so ] && continue
      f=$O/shape_$(echo $shape | tr ' ' x)_b${nq}${L}.json
      RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$L.so timeout -k 10 200 python tools/shape_bench.py $shape $nq 32 fp16 20 > $f 2> $f.err || { tail -3 $f.err; continue; }
      python - "$f" "$shape b=$nq lib=${L:-main}" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "scan frac", j["roofline"]["frac"], "b2b_ms", j["back_to_back_ms"], "b2b frac", j["back_to_back_frac_of_8TBps"])
PY
    done
  done
done
