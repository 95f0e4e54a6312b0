#!/bin/bash
# GPU box: same-device A/B of the main library against variant libraries (LIBS="_x _y"; built with RR_LIB_SUFFIX / RR_EXTRA_DEFINES)
# on the shapes that matter: headline, config 2, wide rows.  Alternates the libraries per shape (A B A B) against drift.
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_ab; mkdir -p $O
for shape in ${SHAPES:-"10000000 768" "1000000 768" "4000000 1024" "2000000 4096"}; do
  for rep in 1 2; do
    for L in main ${LIBS}; do
      S=$L; [ "$L" = main ] && S=""
      f=$O/shape_$(echo $shape | tr ' ' x)_${L}_$rep.json
      RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$S.so timeout -k 10 200 python tools/shape_bench.py $shape 256 ${K:-32} fp16 ${ITERS:-30} > $f 2> $f.err || { tail -3 $f.err; continue; }
      python - "$f" "$shape lib=$L rep=$rep" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[2], "scan frac", r["frac"], "b2b_ms", j["back_to_back_ms"], "b2b frac", j["back_to_back_frac_of_8TBps"])
PY
    done
  done
done
