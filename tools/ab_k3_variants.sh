#!/bin/bash
# on the GPU box: the K3 variants of gpurun_tmp/variants (tools/build_variant.sh) against the product library — bit-identity of the
# distances, kernel time of the chip-filling launch (tools/bench_mlp.py, interleaved repetitions: the chip is power-limited and drifts),
# per-layer cycle tables of the instrumented builds (k3ts*.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
V=$R/gpurun_tmp/variants
libs=""
for f in $V/*.so; do n=$(basename $f .so); case $n in *ts) ;; *) libs="$libs $n";; esac; done
echo "# bit identity against the product library"
timeout 300 python3 tools/k3_dump.py /tmp/k3_base.pt 2>/dev/null | tail -1
for n in $libs; do RA_LIB_PATH=$V/$n.so timeout 300 python3 tools/k3_dump.py /tmp/k3_$n.pt 2>/dev/null | tail -1; python3 tools/k3_dump.py cmp /tmp/k3_base.pt /tmp/k3_$n.pt; done
echo "# kernel time, 5.12 M points (tools/bench_mlp.py), ${REPS:-3} interleaved repetitions"
for rep in $(seq ${REPS:-3}); do
  echo "base: $(timeout 300 python3 tools/bench_mlp.py 2>/dev/null | tail -1)"
  for n in $libs; do echo "$n: $(RA_LIB_PATH=$V/$n.so timeout 300 python3 tools/bench_mlp.py 2>/dev/null | tail -1)"; done
done
for f in $V/*ts.so; do
  [ -f "$f" ] || continue
  echo "# RA_LIB_PATH=$(basename $f) tools/k3_timestamps.py (5.1 M points)"
  RA_LIB_PATH=$f timeout 300 python3 tools/k3_timestamps.py 2>/dev/null
done
