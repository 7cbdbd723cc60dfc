#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/r2n_a -o a -- python3 $R/tools/emul_once.py > $O/r2n_a.log 2>&1 || { tail -5 $O/r2n_a.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r2n_b -o b -- python3 $R/tools/emul_once.py > $O/r2n_b.log 2>&1 || { tail -5 $O/r2n_b.log; exit 1; }
cd $R
python - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/r2n_a","gpurun_out/r2n_b"):
    per=collections.defaultdict(float); n=0
    for f in glob.glob(d+"/*counter_collection.csv"):
        for x in csv.DictReader(open(f)):
            if "rns_gemm256" in x["Kernel_Name"]:
                per[x["Counter_Name"]]+=float(x["Counter_Value"])
    print(d, {k: "%.3e"%v for k,v in per.items()})
PY
exit 0
