#!/bin/bash
# Round profile collection on the GPU box (run from the repo root through gpurun):
#   kernel-trace stats of the default bench command + PMC passes (each its own run, kernel-trace only) -> gpurun_out/prof/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>&1
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc$i -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out/prof'
agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in sorted(glob.glob(out + '/pmc*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        k = 'mlp_sdf_stream_kernel' if 'mlp_sdf_stream' in k else ('hdq_coarse_kernel' if 'hdq_coarse' in k else ('mlp_full_kernel' if 'mlp_full' in k else None))
        if k is None: continue
        agg[(k, r['Counter_Name'])] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
with open(out + '/pmc_summary.csv', 'w') as f:
    f.write('kernel,counter,value,dispatches\n')
    for (k, c), v in sorted(agg.items()): f.write(f'{k},{c},{v:.0f},{cnt[(k, c)]}\n')
print(open(out + '/pmc_summary.csv').read())
P
head -12 $OUT/kt/r_kernel_stats.csv | cut -c1-160
cat $OUT/bench.json
