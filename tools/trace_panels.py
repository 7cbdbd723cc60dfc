"""Panel-level timeline of the last factorisation in a kernel trace: chain queue per panel vs bulk launches (test tooling)."""
import sys, csv, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
kb = [r for r in rows if "kbuild" in r["Kernel_Name"]]
t0 = int(kb[-1]["Start_Timestamp"])
last = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 6
S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e3
E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e3
pot = [r for r in last if "potrf" in r["Kernel_Name"]]
chain_q = pot[0]["Queue_Id"]
print("chain queue", chain_q, "potrf launches", len(pot))
chain = [r for r in last if r["Queue_Id"] == chain_q]
# split chain kernels into panels by potrf count
panels, cur, npot = [], [], 0
for r in chain:
    if "potrf" in r["Kernel_Name"]:
        if npot and npot % W == 0:
            panels.append(cur); cur = []
        npot += 2 if "pair" in r["Kernel_Name"] else 1     # potrf_pair_kernel factors two tile columns
    cur.append(r)
panels.append(cur)
others = [r for r in last if r["Queue_Id"] != chain_q and E(r) - S(r) > 100]
print("panel  chain_start  chain_end  chain_ms  busy_ms | other-queue launches > 0.1 ms overlapping")
for j, p in enumerate(panels):
    s, e = S(p[0]), E(p[-1])
    busy = sum(E(r) - S(r) for r in p)
    ov = [(r["Queue_Id"], round(S(r) / 1e3, 2), round(E(r) / 1e3, 2), r["Grid_Size_X"]) for r in others if S(r) < e and E(r) > s]
    print("%3d %10.2f %10.2f %8.2f %8.2f | %s" % (j, s / 1e3, e / 1e3, (e - s) / 1e3, busy / 1e3, ov))
print("end of fit %.2f ms" % (max(E(r) for r in last) / 1e3))
if len(sys.argv) > 3:
    for j in [int(x) for x in sys.argv[3].split(",")]:
        print("--- chain kernels of panel", j)
        prev = None
        allq = len(sys.argv) > 4 and sys.argv[4] == "all"   # every queue's launches inside the panel's window
        lo, hi = S(panels[j][0]), E(panels[j][-1])
        sel = sorted([r for r in last if S(r) < hi and E(r) > lo], key=S) if allq else panels[j]
        for r in sel:
            if allq: print("  q%-3s" % r["Queue_Id"], end="")
            gap = 0.0 if prev is None else S(r) - prev
            prev = E(r)
            print("  %9.1f %9.1f dur %7.1f gap %6.1f wg %6d x %s %s" % (S(r), E(r), E(r) - S(r), gap,
                  int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Workgroup_Size_X"], r["Kernel_Name"][:34]))
