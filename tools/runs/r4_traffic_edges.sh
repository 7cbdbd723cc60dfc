#!/bin/bash
# HBM-side traffic of the dominant GEMM symbol against the super-tile edge: FETCH_SIZE / WRITE_SIZE passes of the bench command per edge
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4u}; mkdir -p $out; export TMPDIR=/tmp
for st in 4 16 12; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/$out/${c}_$st" -o pmc -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-emulated-line --option supertile=$st > "$GRAFT_REPO_ROOT/$out/${c}_$st.log" 2>&1 ); echo "$st $c rc=$?"
  done
  python3 tools/pmc_traffic.py $out/FETCH_SIZE_$st $out/WRITE_SIZE_$st $out/gemm_traffic_$st.json --min-wgs 1400 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('supertile', $st, 'traffic GB/launch', d['traffic_bytes_per_launch']/1e9, 'x algorithmic', d['traffic_over_algorithmic'])"
  find $out -name "*counter_collection.csv" -delete
done
