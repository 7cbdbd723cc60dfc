"""Fused gp_fit_predict timeline from a kernel trace: when the factorisation's chain ends, per-queue busy time (test tooling)."""
import sys, csv, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
kb = [r for r in rows if "kbuild" in r["Kernel_Name"]]
t0 = int(kb[-1]["Start_Timestamp"])
last = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e6
E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e6
pot = [r for r in last if "potrf" in r["Kernel_Name"]]
print("potrf launches %d; first %.2f ms, last ends %.2f ms; end of everything %.2f ms" % (len(pot), S(pot[0]), E(pot[-1]), max(E(r) for r in last)))
for k in range(0, len(pot), 12):
    print("  potrf %3d starts %.2f ms" % (k, S(pot[k])))
for q in sorted(set(r["Queue_Id"] for r in last)):
    rs = [r for r in last if r["Queue_Id"] == q]
    print("queue %s: %4d kernels, busy %.2f ms, first %.2f last %.2f" % (q, len(rs), sum(E(r) - S(r) for r in rs), S(rs[0]), E(rs[-1])))
big = [r for r in last if E(r) - S(r) > 0.25]
for r in big:
    print("  q=%s %7.2f -> %7.2f (%6.2f ms) wg %7d %s" % (r["Queue_Id"], S(r), E(r), E(r) - S(r), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Kernel_Name"][:40]))
