#!/bin/bash
# animate bench + summary
python bench.py --animate --no-cpu-baseline "$@" > gpurun_out/b_anim.json 2> gpurun_out/b_anim.err; tail -3 gpurun_out/b_anim.err
python - <<PY
import json
d=json.loads(open("gpurun_out/b_anim.json").read().strip().splitlines()[-1])
print({k:round(d[k],4) for k in ("ms_per_step","ms_per_step_sequential","host_enqueue_ms_per_step","host_wait_ms_per_step","ms_per_step_frames_posed_before_the_loop","animate_over_preposed")}, d["config"]["rays_in_bbox"], d["config"]["hit_pixels_per_frame"], d["roofline"]["frac"])
PY
