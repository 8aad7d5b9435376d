#!/bin/bash
# on the GPU box: same-box A/B of K3CC (csrc/ra_k3cc.hpp: four waves per 16-point tile of the compensated distance query) on a testing
# build of ra_k3c_f16.hip (tools/build_variant.sh k3c_testing ra_k3c_f16.hip "-DRA_TESTING"): RA_K3C_COOP_MAX=0 restores the 2-wave tiles
# of K3C for launches of at most 8 Ki points.  -> profiles/r04_k3cc_ab.txt
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms', round(d['ms_per_step'],3), 'seq', round(d.get('ms_per_step_sequential') or 0,3), 'frac', round(d['roofline']['frac'],4))"; }
export RA_LIB_PATH=gpurun_tmp/variants/k3c_testing.so
for coop in 0 8192; do
  export RA_K3C_COOP_MAX=$coop
  echo "== RA_K3C_COOP_MAX=$coop: kernel time per launch (tools/bench_k3c.py)"
  RA_K3C_SIZES=300,1200,2400,4096,5000,8192 python tools/bench_k3c.py 2>&1 | grep "precision 2"
  echo "== RA_K3C_COOP_MAX=$coop: frames (bench.py; ms in flight, sequential)"
  for r in 0 4; do python bench.py --emulate-world 8 --emulate-rank $r --no-cpu-baseline 2>/dev/null | p "relight 512 N=8 rank $r"; done
  python bench.py --mode sphere_tracing --size 256 --no-cpu-baseline 2>/dev/null | p "sphere tracing 256"
  python bench.py --mode sphere_tracing --emulate-world 8 --no-cpu-baseline 2>/dev/null | p "sphere tracing 512 N=8 rank 0"
  python bench.py --mode novel_light --ground --emulate-world 8 --no-cpu-baseline 2>/dev/null | p "config 5 N=8 rank 0"
done
