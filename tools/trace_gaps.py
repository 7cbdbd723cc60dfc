"""Where the chip is not running a long launch during the last fused step of a kernel trace (test tooling).
usage: trace_gaps.py <rocprof dir> [min workgroups of a 'long' launch = 1000] [min gap us = 40]"""
import sys, csv, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
minwg = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mingap = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
kb = [r for r in rows if "kbuild" in r["Kernel_Name"]]
t0 = int(kb[-1]["Start_Timestamp"])
last = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e3
E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e3
wg = lambda r: (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"])) * max(1, int(r["Grid_Size_Z"]))
end = max(E(r) for r in last)
big = sorted([(S(r), E(r)) for r in last if wg(r) >= minwg])
cov, cur_s, cur_e, gaps = 0.0, None, None, []
for s, e in big:
    if cur_e is None: cur_s, cur_e = s, e; gaps.append((0.0, s)); continue
    if s > cur_e:
        cov += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
cov += cur_e - cur_s; gaps.append((cur_e, end))
print("step %.2f ms; a launch of >= %d workgroups is running for %.2f ms; not running for %.2f ms" % (end / 1e3, minwg, cov / 1e3, (end - cov) / 1e3))
for a, b in gaps:
    if b - a >= mingap:
        inside = [r for r in last if S(r) < b and E(r) > a]
        names = {}
        for r in inside:
            k = r["Kernel_Name"][:28]; names[k] = names.get(k, 0) + (min(E(r), b) - max(S(r), a))
        top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
        print("  gap %8.2f -> %8.2f ms (%6.0f us): %s" % (a / 1e3, b / 1e3, b - a, ", ".join("%s %.0f" % (k, v) for k, v in top)))
