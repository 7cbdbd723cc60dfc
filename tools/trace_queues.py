"""Per-queue busy time and a coarse utilisation timeline of the last pass in a kernel trace (test tooling)."""
import sys, csv, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
kb = [r for r in rows if "kbuild" in r["Kernel_Name"]]
t0 = int(kb[-1]["Start_Timestamp"])
last = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
tend = max(int(r["End_Timestamp"]) for r in last)
print("span %.2f ms, %d kernels" % ((tend - t0) / 1e6, len(last)))
qs = sorted(set(r["Queue_Id"] for r in last))
for q in qs:
    rs = [r for r in last if r["Queue_Id"] == q]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    names = sorted(set(r["Kernel_Name"][:18] for r in rs))
    print("queue %s: %4d kernels busy %.2f ms  first %.2f last %.2f  %s" % (q, len(rs), busy, (int(rs[0]["Start_Timestamp"]) - t0) / 1e6, (int(rs[-1]["End_Timestamp"]) - t0) / 1e6, names[:5]))
# timeline: for each 2 ms bucket, list which big gemm launches (>0.3 ms) are active per queue
big = [r for r in last if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 300000]
for r in big:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6; e = (int(r["End_Timestamp"]) - t0) / 1e6
    print("  q=%s %7.2f -> %7.2f (%6.2f ms) grid %8s %s" % (r["Queue_Id"], s, e, e - s, r["Grid_Size_X"], r["Kernel_Name"][:24]))
