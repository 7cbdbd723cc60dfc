#!/bin/bash
# kernel timeline of one-location gradient calls at N = 512 / 2048 (launch-latency bound)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5rows512
mkdir -p $out
for n in 512 2048; do
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/prof$n -o rows -- python3 $GRAFT_REPO_ROOT/tools/gpbench.py rows_trace $n 100 > $GRAFT_REPO_ROOT/$out/prof$n.log 2>&1) || exit 1
done
python3 - <<'PY'
import sqlite3, glob
for n in (512, 2048):
    f = glob.glob("gpurun_out/r5rows512/prof%d/**/*.db" % n, recursive=True)[0]
    c = sqlite3.connect(f)
    ks = c.execute("select name,start,end from kernels where name like '%rows_%' order by start").fetchall()
    # gradient calls: forward, backward, finish triples
    trip = [(ks[i], ks[i+1], ks[i+2]) for i in range(len(ks)-2) if 'forward' in ks[i][0] and 'backward' in ks[i+1][0] and 'finish' in ks[i+2][0]]
    trip = trip[5:95]
    import statistics as st
    d = lambda k: (k[2]-k[1])/1e3
    print("N=%d gradient calls: forward %.1f us, backward %.1f, finish %.1f; gaps f->b %.1f, b->f %.1f; span %.1f; period %.1f" % (
        n, st.median(d(t[0]) for t in trip), st.median(d(t[1]) for t in trip), st.median(d(t[2]) for t in trip),
        st.median((t[1][1]-t[0][2])/1e3 for t in trip), st.median((t[2][1]-t[1][2])/1e3 for t in trip),
        st.median((t[2][2]-t[0][1])/1e3 for t in trip),
        st.median((trip[i+1][0][1]-trip[i][0][1])/1e3 for i in range(len(trip)-1))))
PY
