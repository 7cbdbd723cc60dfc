#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r2g_sq -o s -- python3 $R/tools/emul_once.py > $O/r2g_sq.log 2>&1 || { tail -5 $O/r2g_sq.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $O/r2g_f -o f -- python3 $R/tools/emul_once.py > $O/r2g_f.log 2>&1 || { tail -5 $O/r2g_f.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $O/r2g_w -o w -- python3 $R/tools/emul_once.py > $O/r2g_w.log 2>&1 || { tail -5 $O/r2g_w.log; exit 1; }
cd $R
python - <<'PY'
import csv,glob,collections
def load(d):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d+"/*counter_collection.csv"):
        for x in csv.DictReader(open(f)):
            k=x["Kernel_Name"].split("(")[0][:40]
            per[k][x["Counter_Name"]]+=float(x["Counter_Value"])
            per[k]["n_"+x["Counter_Name"]]+=1
            per[k]["t_"+x["Counter_Name"]]+=(int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3
    return per
for d in ("gpurun_out/r2g_sq","gpurun_out/r2g_f","gpurun_out/r2g_w"):
    per=load(d)
    for k,v in per.items():
        if "rns" in k or "gemm_nt_kernel<1, 128, 4" in k:
            print(d[-3:],k,{c:(round(x/1e9,3) if x>1e7 else round(x,1)) for c,x in v.items()})
PY
find $O -name "*kernel_trace.csv" -size +30M -delete
exit 0
