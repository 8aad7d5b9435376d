#!/bin/bash
# A/B of cfg.trace_precision on one box: 512 x 512 relight at 3 / 1 frames in flight, sphere tracing sequential
out=gpurun_out/b_tp.jsonl; rm -f $out
for tp in 1 0; do for d in 3 1; do python bench.py --trace-precision $tp --frames-in-flight $d --no-cpu-baseline >> $out 2>gpurun_out/b_tp.err; done; done
for tp in 1 0; do python bench.py --mode sphere_tracing --no-cpu-baseline --frames-in-flight 1 --trace-precision $tp >> $out; python bench.py --mode sphere_tracing --no-cpu-baseline --frames-in-flight 3 --trace-precision $tp >> $out; done
python - <<PY
import json
for l in open("$out"):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"][:30], "tp",d["config"].get("trace_precision"),"fif",d["config"]["frames_in_flight"],"ms",round(d["ms_per_step"],3),"frac",round(d["roofline"]["frac"],4), "comp", d["config"].get("fine_queries_compensated_per_frame"))
PY
