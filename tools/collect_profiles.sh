#!/bin/bash
# Round profile collection on the GPU box (run from the repo root through gpurun):
#   kernel-trace stats of the default bench command and of the other BASELINE shapes, PMC passes of the default command (each
#   its own run, kernel-trace only), emulated per-rank times of a 2/4/8-rank job -> gpurun_out/prof/  (copy the summaries to
#   profiles/rNN_* and commit them together with any kernel change: bench.py reads the newest rNN_relight512_pmc.csv)
# NOTE bench.py takes `roofline.traffic` from the COMMITTED profiles/rNN_*_pmc.csv, i.e. from the previous collection: after a
#   change of launch granularity (e.g. cfg.volume_chunk_rays) copy the new summaries to profiles/ and collect once more, or the
#   per-launch traffic in the new bench lines belongs to the old launch size.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
RN=${RA_ROUND:-r06}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $OUT/${RN}_bench.json 2> $OUT/bench.err
: > $OUT/${RN}_bench_other_shapes.jsonl
python3 $R/bench.py --mode sphere_tracing --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/${RN}_bench_other_shapes.jsonl 2>> $OUT/bench.err
python3 $R/bench.py --mode anisdf --steps 10 --warmup 2 --no-cpu-baseline >> $OUT/${RN}_bench_other_shapes.jsonl 2>> $OUT/bench.err
python3 $R/bench.py --ground --steps 5 --warmup 2 --no-cpu-baseline >> $OUT/${RN}_bench_other_shapes.jsonl 2>> $OUT/bench.err
python3 $R/bench.py --mode novel_light --size 1024 --probes 8 --steps 5 --warmup 2 --no-cpu-baseline >> $OUT/${RN}_bench_other_shapes.jsonl 2>> $OUT/bench.err
python3 $R/bench.py --skin-noise 0 --steps 20 --warmup 3 > $OUT/${RN}_bench_skin_noise0.json 2>> $OUT/bench.err        # the smooth body, with its PSNR
python3 $R/bench.py --coverage 0.35 --steps 10 --warmup 2 > $OUT/${RN}_bench_coverage35.json 2>> $OUT/bench.err          # a frame-filling subject (camera at 0.96 m), with its PSNR
python3 $R/bench.py --trace-precision 0 --steps 20 --warmup 3 > $OUT/${RN}_bench_trace_precision0.json 2>> $OUT/bench.err  # round 3's arithmetic (plain f16 surface trace): what the compensated tier costs and buys
python3 $R/bench.py --trace-precision 2 --steps 10 --warmup 3 > $OUT/${RN}_bench_trace_precision2.json 2>> $OUT/bench.err  # every distance query compensated: the tier that meets max |err| <= 1e-2 on every pixel, and its price
python3 $R/bench.py --body split --steps 10 --warmup 3 > $OUT/${RN}_bench_body_split.json 2>> $OUT/bench.err               # the hard-case body (a part that shadows the body at distance, front key light, 12 shadow iterations)
python3 $R/bench.py --body split --weights sharp --steps 10 --warmup 3 > $OUT/${RN}_bench_body_split_weights_sharp.json 2>> $OUT/bench.err
python3 $R/bench.py --weights sharp --steps 10 --warmup 3 > $OUT/${RN}_bench_weights_sharp.json 2>> $OUT/bench.err
: > $OUT/${RN}_bench_animate.jsonl
python3 $R/bench.py --animate --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/${RN}_bench_animate.jsonl 2>> $OUT/bench.err   # N3 + N2 inside the timed region
python3 $R/bench.py --animate --frames-in-flight 1 --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/${RN}_bench_animate.jsonl 2>> $OUT/bench.err
python3 $R/bench.py --animate --mode sphere_tracing --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/${RN}_bench_animate.jsonl 2>> $OUT/bench.err
# sequential frames (one in flight) next to the default two: the whole frame and one rank's share of 2 / 4 / 8
: > $OUT/${RN}_frames_in_flight.jsonl
for d in 1 2 3 4; do python3 $R/bench.py --frames-in-flight $d --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/${RN}_frames_in_flight.jsonl 2>> $OUT/bench.err; done
for d in 1 2; do for n in 2 4 8; do python3 $R/bench.py --frames-in-flight $d --emulate-world $n --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/${RN}_frames_in_flight.jsonl 2>> $OUT/bench.err; done; done
python3 $R/bench.py --frames-in-flight 1 --mode sphere_tracing --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/${RN}_frames_in_flight.jsonl 2>> $OUT/bench.err
: > $OUT/${RN}_emulate_world.jsonl
# EVERY rank's share (the slowest rank sets the frame time of a real job): config.emulate_rank says which
for n in 2 4 8; do for r in $(seq 0 $((n-1))); do python3 $R/bench.py --emulate-world $n --emulate-rank $r --steps 20 --warmup 3 --soak 1 --no-cpu-baseline >> $OUT/${RN}_emulate_world.jsonl 2>> $OUT/bench.err; done; done
# config 5 (1024 x 1024, 8 probes) and the README command (novel light + ground) as one rank of an N-rank job
: > $OUT/${RN}_emulate_world_config5.jsonl
for n in 1 2 4 8; do for r in $(seq 0 $((n-1))); do python3 $R/bench.py --emulate-world $n --emulate-rank $r --size 1024 --mode novel_light --probes 8 --steps 10 --warmup 3 --soak 1 --no-cpu-baseline >> $OUT/${RN}_emulate_world_config5.jsonl 2>> $OUT/bench.err; done; done
for n in 1 2 4 8; do python3 $R/bench.py --emulate-world $n --size 512 --mode novel_light --ground --probes 2 --steps 10 --warmup 3 --no-cpu-baseline >> $OUT/${RN}_emulate_world_config5.jsonl 2>> $OUT/bench.err; done
$R/tools/frame_anatomy.sh w8 44 -- --emulate-world 8 --steps 10 --warmup 3 > $OUT/${RN}_emulate_world8_kernels.txt 2>> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- python3 $R/bench.py --steps 7 --warmup 2 --no-cpu-baseline --soak 0 > /dev/null 2>&1
cp $OUT/kt/r_kernel_stats.csv $OUT/${RN}_relight512_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_s -o r -- python3 $R/bench.py --mode sphere_tracing --steps 7 --warmup 2 --no-cpu-baseline --soak 0 > /dev/null 2>&1
cp $OUT/kt_s/r_kernel_stats.csv $OUT/${RN}_sphere512_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_a -o r -- python3 $R/bench.py --mode anisdf --steps 7 --warmup 2 --no-cpu-baseline --soak 0 > /dev/null 2>&1
cp $OUT/kt_a/r_kernel_stats.csv $OUT/${RN}_anisdf512_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_g -o r -- python3 $R/bench.py --ground --steps 3 --warmup 1 --no-cpu-baseline --soak 0 > /dev/null 2>&1
cp $OUT/kt_g/r_kernel_stats.csv $OUT/${RN}_relight_ground512_kernel_stats.csv
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc$i -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --soak 0 --frames-in-flight 1 > /dev/null 2>&1
done
# the same two HBM counters for the volume path (the full query's tape traffic)
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmcv$i -o r -- python3 $R/bench.py --mode anisdf --steps 2 --warmup 1 --no-cpu-baseline --soak 0 --frames-in-flight 1 > /dev/null 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out/prof'
rn = os.environ.get('RA_ROUND', 'r04')
fams = (('mlp_sdf_coop', 'mlp_sdf_coop_kernel'), ('mlp_sdf_comp', 'mlp_sdf_comp_kernel'), ('mlp_sdf_stream_kernelIDF16_Li8E', 'mlp_sdf_stream_kernel_w8'), ('mlp_sdf_stream_kernel<_Float16, 8>', 'mlp_sdf_stream_kernel_w8'), ('mlp_sdf_stream', 'mlp_sdf_stream_kernel'), ('hdq_coarse', 'hdq_coarse_kernel'), ('mlp_fwd_tape', 'mlp_fwd_tape_kernel'),
        ('mlp_bwd_heads', 'mlp_bwd_heads_kernel'))
for pat, name in (('/pmc[0-9]*/', 'relight512'), ('/pmcv[0-9]*/', 'anisdf512')):      # one summary per workload
    agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in sorted(glob.glob(out + pat + '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k = next((v for s, v in fams if s in r['Kernel_Name']), None)
            if k is None: continue
            agg[(k, r['Counter_Name'])] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    with open(out + f'/{rn}_{name}_pmc.csv', 'w') as f:
        f.write('kernel,counter,value,dispatches\n')
        for (k, c), v in sorted(agg.items()): f.write(f'{k},{c},{v:.0f},{cnt[(k, c)]}\n')
    print(open(out + f'/{rn}_{name}_pmc.csv').read())
P
head -12 $OUT/${RN}_relight512_kernel_stats.csv | cut -c1-160
cat $OUT/${RN}_bench.json
