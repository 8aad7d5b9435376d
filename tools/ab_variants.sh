#!/bin/bash
# on the GPU box: per-kernel average durations of `bench.py ARGS` for the product library and every gpurun_tmp/variants/*.so
# usage: tools/ab_variants.sh KERNEL_REGEX -- bench args
R=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
shopt -s nullglob                       # no variants built -> only the product library is timed
for lib in "" "$R"/gpurun_tmp/variants/*.so; do
  if [ -n "$lib" ] && [ ! -f "$lib" ]; then continue; fi
  n=base; [ -n "$lib" ] && n=$(basename "$lib" .so)
  d="/tmp/ab_run_$n"
  rm -rf -- "$d"
  RA_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o r -- python3 $R/bench.py "$@" --no-cpu-baseline > "$d.json" 2>/dev/null
  python3 - "$d" "$d.json" "$pat" "$n" <<'P'
import csv, glob, json, re, sys
d, j, pat, n = sys.argv[1:5]
try: ms = round(json.loads(open(j).read().strip().splitlines()[-1])['ms_per_step'], 3)
except Exception: ms = None
print(f'== {n}  ms/step {ms}')
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r['Name']):
            print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg_us {float(r['AverageNs']) / 1e3:9.1f}")
P
done
