#!/bin/bash
# GPU box: SQ / TA counters of the scan launches at 4M x 1024, 256 queries: row-split kernel against the 8-wave kernel (RR_WIDE_RS=0)
set -o pipefail
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_rs_pmc; mkdir -p $O
for rs in 1 0; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
             "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
             "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    RR_WIDE_RS=$rs timeout -k 10 150 rocprofv3 --pmc $set -d $O/rs${rs}_p$i -o t --output-format csv -- python3 $R/tools/shape_bench.py 4000000 1024 256 10 fp16 4 > $O/rs${rs}_p$i.log 2>&1
    echo "rs=$rs pass $i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_rs_pmc"
for rs in (1,0):
    acc=collections.defaultdict(float); n=collections.defaultdict(int)
    for f in glob.glob(f"{O}/rs{rs}_p*/t_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "flat_scan_wide" in k and "Lb1E" not in k.split("kernel")[1][:12]:
                if ("rs_kernel" in k) == (rs==1) or ("wide8" in k and rs==0):
                    acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("rs",rs,{k:round(v/max(1,n[k]),1) for k,v in sorted(acc.items())})
PY
