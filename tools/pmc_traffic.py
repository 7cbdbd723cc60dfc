"""HBM-side traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes; both counters are in KB).

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>
                      [--kernel SUBSTRING] [--min-wgs N] [--f16 F] [--f8 F] [--mode gemm|stream]

gfx950 corrections, calibrated on this box with tools/micro/fetch_calib (known byte counts, same access patterns):
  --f16  bytes per counted FETCH_SIZE byte for 16-byte-per-lane streaming loads (the guide's value: 2.0)
  --f8   the same for the GEMM's C-tile prologue (buffer_load_b64: 4 rows x 128 B per wave instruction)
mode gemm (C -= A B^T): every C tile is read once and written once, so the C read bytes are known exactly
  (= WRITE_SIZE bytes); what FETCH_SIZE counted for them is C / f8, the rest of the counter is operand-panel staging
  (16-byte loads) and is scaled by f16:   read = C + (FETCH_raw - C / f8) * f16.
mode stream: read = FETCH_raw * f16 (kbuild / cross_k: tiny reads, the traffic is the write).
"""
import argparse
import csv
import glob
import hashlib
import json
import math
import os

ap = argparse.ArgumentParser()
ap.add_argument("fetch_dir"); ap.add_argument("write_dir"); ap.add_argument("out")
ap.add_argument("--kernel", default="gemm_nt_kernel<1, 128, 4, false, 128>")
ap.add_argument("--min-wgs", type=int, default=1)
ap.add_argument("--f16", type=float, default=2.0)
ap.add_argument("--f8", type=float, default=2.0)
ap.add_argument("--mode", default="gemm")
a = ap.parse_args()


def collect(d, counter):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or a.kernel not in r["Kernel_Name"]:
                continue
            wgs = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            if wgs < a.min_wgs:
                continue
            per.setdefault((f, r["Dispatch_Id"]), [wgs, 0.0])[1] += float(r["Counter_Value"])
    return per


fe, wr = collect(a.fetch_dir, "FETCH_SIZE"), collect(a.write_dir, "WRITE_SIZE")
n = max(len(fe), 1)
fetch_raw = sum(v[1] for v in fe.values()) / n * 1024.0
write_b = sum(v[1] for v in wr.values()) / max(len(wr), 1) * 1024.0
wgs = sum(v[0] for v in fe.values()) / n
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "gaussian_process_optimization_amd", "csrc", "gemm.hip")
out = {"kernel": a.kernel, "launches": len(fe), "avg_workgroups_per_launch": wgs,
       "FETCH_SIZE_bytes_per_launch_raw": fetch_raw, "WRITE_SIZE_bytes_per_launch": write_b,
       "correction_16B_loads": a.f16, "correction_8B_C_tile_loads": a.f8,
       "gemm_hip_sha256_16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]}
if a.mode == "gemm":
    c_read = write_b
    operand = max(0.0, fetch_raw - c_read / a.f8) * a.f16
    tiles = wgs                                  # one workgroup per 128 x 128 output tile
    alg_c = tiles * 128 * 128 * 8 * 2
    alg_op = 2.0 * math.sqrt(tiles) * 128 * 768 * 8   # >= 2 sqrt(T) distinct 128 x 768 operand panels behind T tiles
    out.update({"C_read_bytes_per_launch": c_read, "operand_read_bytes_per_launch": operand,
                "read_bytes_per_launch_corrected": c_read + operand,
                "traffic_bytes_per_launch": c_read + operand + write_b,
                "algorithmic_bytes_per_launch": alg_c + alg_op,
                "algorithmic_C_bytes_per_launch": alg_c, "algorithmic_operand_bytes_lower_bound": alg_op,
                "traffic_over_algorithmic": (c_read + operand + write_b) / (alg_c + alg_op)})
else:
    out.update({"read_bytes_per_launch_corrected": fetch_raw * a.f16,
                "traffic_bytes_per_launch": fetch_raw * a.f16 + write_b})
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out))
