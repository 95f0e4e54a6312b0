#!/bin/bash
# GPU box: parity suite, then config 2 (1M x 768) and headline (10M) timings with kernel traces
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_c2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_baseline_configs_gpu.py::test_config5_80m_bf16_k100 > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for n in 1000000 10000000; do
  python tools/quickperf.py $n 768 256 > $O/quick_$n.log 2>&1 || { tail $O/quick_$n.log; exit 1; }
  grep "ms/batch" $O/quick_$n.log
  RR_CHUNK_GROWTH=8 python tools/quickperf.py $n 768 256 > $O/quick_g8_$n.log 2>&1
  echo "growth 8: $(grep 'ms/batch' $O/quick_g8_$n.log)"
done
python tools/quickperf.py 10000000 768 1 | grep "ms/batch"
rocprofv3 --kernel-trace --stats -d $O/c2_trace -o t --output-format csv -- python3 tools/quickperf.py 1000000 768 256 > $O/c2_prof.log 2>&1 || exit 1
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/r02_c2/c2_trace/t_kernel_trace.csv")) if "rr" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
for r in rows[-9:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if t0 is None: t0=s
    print(f'{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  {r["Kernel_Name"][:70]}')
PY
