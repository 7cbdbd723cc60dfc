#!/bin/bash
# HBM-side read traffic (FETCH_SIZE alone: FETCH_SIZE + WRITE_SIZE in one pass exceed the hardware) of the residue GEMM with the two workgroup orders (rns_interleave 0 / 1)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for il in 0 1; do
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r2r_$il -o c -- python3 $R/tools/emul_once.py rns_interleave=$il > $O/r2r_$il.log 2>&1 || { tail -5 $O/r2r_$il.log; exit 1; }
done
cd $R
python - <<'PY'
import csv,glob,collections,json
out={}
for il in (0,1):
    per=collections.defaultdict(float); n=0; dur=0
    for f in glob.glob("gpurun_out/r2r_%d/*counter_collection.csv"%il):
        for x in csv.DictReader(open(f)):
            if "rns_gemm256" in x["Kernel_Name"] and int(x["Grid_Size"])>=13107200:   # the long launches of the candidate solve
                per[x["Counter_Name"]]+=float(x["Counter_Value"])
                if x["Counter_Name"]=="FETCH_SIZE": n+=1; dur+=int(x["End_Timestamp"])-int(x["Start_Timestamp"])
    out["rns_interleave=%d"%il]={"launches":n,"avg_ms":dur/max(n,1)/1e6, **{k:v/max(n,1) for k,v in per.items()}}
print(json.dumps(out,indent=1))
json.dump(out,open("gpurun_out/r2r_summary.json","w"),indent=1)
PY
exit 0
