#!/bin/bash
# GPU box: timing-only ablations of the wide-row step (development; results of the ablated libraries are wrong by design)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_abl
mkdir -p $O
for d in ${DIMS:-4096 1024}; do
  for v in "" ${ABLS:-_abl1 _abl2 _abl4 _abl8 _abl6 _abl14 _abl30}; do
    RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$v.so timeout -k 10 200 python tools/generic_perf.py $d 2000000 > $O/perf${v}_d$d.log 2>&1
    echo "lib$v $(grep 'nq=' $O/perf${v}_d$d.log | tr '\n' ' ')"
  done
done
