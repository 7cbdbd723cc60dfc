#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2p_k -o k -- python3 $R/tools/emul_once.py rns_group=8 > $O/r2p_k.log 2>&1 || { tail -5 $O/r2p_k.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/r2p_a -o a -- python3 $R/tools/emul_once.py rns_group=8 > $O/r2p_a.log 2>&1 || { tail -5 $O/r2p_a.log; exit 1; }
cd $R
python - <<'PY'
import csv,glob,collections
for f in glob.glob("gpurun_out/r2p_k/*kernel_stats.csv"):
    for x in list(csv.DictReader(open(f)))[:14]:
        print("%-60s calls %6s total %10.3f ms avg %9.1f us  %5s%%" % (x["Name"][:60], x["Calls"], float(x["TotalDurationNs"])/1e6, float(x["AverageNs"])/1e3, x["Percentage"]))
per=collections.defaultdict(float)
for f in glob.glob("gpurun_out/r2p_a/*counter_collection.csv"):
    for x in csv.DictReader(open(f)):
        if "rns_gemm256" in x["Kernel_Name"]:
            per[x["Counter_Name"]]+=float(x["Counter_Value"])
print({k: "%.3e"%v for k,v in per.items()})
PY
exit 0
