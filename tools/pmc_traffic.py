"""Average FETCH_SIZE / WRITE_SIZE per launch of the big GEMM launches (>= 1024 workgroups) from rocprofv3 --pmc runs.

usage: pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json>
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950
FETCH_SIZE counts half of the bytes of 16-byte-per-lane streaming loads (the operand panels), so it is doubled;
the C tiles are read 8 bytes per lane (uncalibrated width) -- the doubled figure is therefore an upper bound on reads.
"""
import sys, csv, glob, json


def collect(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "gemm_nt_kernel" not in r["Kernel_Name"]:
            continue
        wgs = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        if wgs < 1024 or ", 64," in r["Kernel_Name"]:
            continue
        per.setdefault(r["Dispatch_Id"], [r["Kernel_Name"], wgs, 0.0])[2] += float(r["Counter_Value"])
    return per


fe = collect(sys.argv[1], "FETCH_SIZE")
wr = collect(sys.argv[2], "WRITE_SIZE")
n = len(fe)
fetch_kb = sum(v[2] for v in fe.values()) / max(n, 1)
write_kb = sum(v[2] for v in wr.values()) / max(len(wr), 1)
tiles = sum(v[1] for v in fe.values()) / max(n, 1)
out = {"launches": n, "avg_tiles_per_launch": tiles,
       "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "read_bytes_per_launch_corrected": 2.0 * fetch_kb * 1024, "write_bytes_per_launch": write_kb * 1024,
       "traffic_bytes_per_launch": 2.0 * fetch_kb * 1024 + write_kb * 1024,
       "algorithmic_C_bytes_per_launch": tiles * 128 * 128 * 8 * 2,
       "note": "gemm_nt_kernel launches with >= 1024 workgroups of bench.py; FETCH_SIZE doubled (gfx950, 16 B/lane loads)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
