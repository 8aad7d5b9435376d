#!/bin/bash
# on the GPU box: where do a kernel's wave-cycles go?  ACTIVE (issuing) / WAIT_INST (issue stall: dependency, pipe busy) / WAIT_ANY
# (parked at s_waitcnt or a barrier) as fractions of SQ_WAVE_CYCLES, plus MFMA-pipe busy cycles, per kernel family.
# usage: tools/pmc_wait_split.sh -- bench args          (separate --pmc passes, kernel trace only: see MI355X_MICROARCH.md)
R=${GRAFT_REPO_ROOT:-$(pwd)}
shift
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf -- "/tmp/ws_pmc$i"
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "/tmp/ws_pmc$i" -o r -- python3 $R/bench.py "$@" --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<'P'
import csv, glob, collections
fams = ('mlp_sdf_stream', 'mlp_fwd_tape', 'mlp_bwd_heads', 'hdq_coarse')
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('/tmp/ws_pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((x for x in fams if x in r['Kernel_Name']), None)
        if k: agg[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in fams:
    w = agg.get((k, 'SQ_WAVE_CYCLES'))
    if not w: continue
    g = lambda c: agg.get((k, c), 0.0)
    print(f"{k:16s} dispatches {n[(k,'SQ_WAVE_CYCLES')]:4d}  active {g('SQ_ACTIVE_INST_ANY')/w:.3f}  wait_inst {g('SQ_WAIT_INST_ANY')/w:.3f}  wait_any(parked) {g('SQ_WAIT_ANY')/w:.3f}"
          f"  wait_inst_lds {g('SQ_WAIT_INST_LDS')/w:.3f}  mfma_busy/busy_cycles {g('SQ_VALU_MFMA_BUSY_CYCLES')/max(g('SQ_BUSY_CYCLES'),1):.3f}  valu/mfma {g('SQ_INSTS_VALU')/max(g('SQ_INSTS_MFMA'),1):.2f}")
P
