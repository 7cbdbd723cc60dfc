"""Every launch of the LAST call in a kernel trace, in start order (test tooling).
usage: trace_list.py <rocprof dir> <name of the kernel that starts the call, e.g. cross_k> [min us = 0]"""
import sys, csv, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
mark = [r for r in rows if sys.argv[2] in r["Kernel_Name"]]
t0 = int(mark[-1]["Start_Timestamp"])
minus = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
last = sorted([r for r in rows if int(r["Start_Timestamp"]) >= t0], key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e3
E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e3
tot = {}
for r in last:
    k = r["Kernel_Name"][:40]; tot[k] = tot.get(k, 0.0) + E(r) - S(r)
    if E(r) - S(r) >= minus:
        print("q%-2s %9.1f %9.1f %8.1f wg %6d x %-4s %s" % (r["Queue_Id"], S(r), E(r), E(r) - S(r), (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), r["Workgroup_Size_X"], r["Kernel_Name"][:60]))
print("end %.1f us" % max(E(r) for r in last))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print("  %10.1f us  %s" % (v, k))
