"""Summarise a rocprofv3 kernel-trace CSV: per-kernel totals and the timeline of the last fit (test tooling)."""
import sys, csv, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
print("columns:", list(rows[0].keys()))
ks = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"][:40]
    ks[n][0] += 1; ks[n][1] += d
for n, (c, t) in sorted(ks.items(), key=lambda x: -x[1][1]):
    print("%-42s %6d launches %10.1f us total %8.1f us avg" % (n, c, t, t / c))
# timeline of the second half (last fit): find potrf launches
pot = [r for r in rows if "potrf" in r["Kernel_Name"]]
half = pot[len(pot)//2:]
t0 = int(half[0]["Start_Timestamp"])
last = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
print("last fit: %d kernels, span %.2f ms" % (len(last), (max(int(r["End_Timestamp"]) for r in last) - t0) / 1e6))
qcol = "Queue_Id" if "Queue_Id" in rows[0] else None
for r in last[:90]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f %8.1f q=%s grid=%s %s" % (s, e, e - s, r.get(qcol, "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"][:30]))
