#!/bin/bash
# in-situ SQ counters of the residue GEMM (long launches of the emulated candidate solve), final kernel, 14 moduli
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r2u -o c -- python3 $R/tools/emul_once.py > $O/r2u.log 2>&1 || { tail -5 $O/r2u.log; exit 1; }
cd $R
python - <<'PY'
import csv,glob,collections,json
per=collections.defaultdict(float); n=0; dur=0; wgs=0
for f in glob.glob("gpurun_out/r2u/*counter_collection.csv"):
    for x in csv.DictReader(open(f)):
        if "rns_gemm256" in x["Kernel_Name"] and int(x["Grid_Size"])//512 >= 20000:
            per[x["Counter_Name"]]+=float(x["Counter_Value"])
            if x["Counter_Name"]=="GRBM_GUI_ACTIVE": n+=1; dur+=int(x["End_Timestamp"])-int(x["Start_Timestamp"]); wgs+=int(x["Grid_Size"])//512
clock=per["GRBM_GUI_ACTIVE"]/8/dur
out={"launches":n,"avg_workgroups":wgs/max(n,1),"avg_duration_ms":dur/max(n,1)/1e6,"effective_clock_GHz":clock,
     "SQ_VALU_MFMA_BUSY_CYCLES":per["SQ_VALU_MFMA_BUSY_CYCLES"]/n,"matrix_pipe_utilisation_in_cycles":per["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(clock*dur),
     "SQ_WAVE_CYCLES_quad":per["SQ_WAVE_CYCLES"]/n,"SQ_WAIT_INST_ANY_quad":per["SQ_WAIT_INST_ANY"]/n,"SQ_BUSY_CYCLES":per["SQ_BUSY_CYCLES"]/n}
print(json.dumps(out,indent=1)); json.dump(out,open("gpurun_out/r2u_summary.json","w"),indent=1)
PY
exit 0
