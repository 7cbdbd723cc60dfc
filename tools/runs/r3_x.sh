#!/bin/bash
# HBM-side bytes of the covariance builders (round-3 kernels): FETCH_SIZE and WRITE_SIZE in separate --pmc passes
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3x}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/$out/$c" -o pmc -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-emulated-line > "$GRAFT_REPO_ROOT/$out/$c.log" 2>&1 )
  echo "$c rc=$?"
done
python3 tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE $out/kbuild.json --kernel kbuild_kernel --mode stream; echo "kbuild rc=$?"
python3 tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE $out/cross_k.json --kernel cross_k_kernel --mode stream; echo "cross_k rc=$?"
python3 tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE $out/gemm.json --min-wgs 1400; echo "gemm rc=$?"
cat $out/kbuild.json $out/cross_k.json | head -60
find $out -name "*counter_collection.csv" -size +30M -delete
exit 0
