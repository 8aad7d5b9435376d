"""Per-frame kernel table of one profiled bench.py run, over STEADY-STATE frames only.
usage (on the GPU box): cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 bench.py ARGS > /tmp/kt.json
                        python3 tools/kernel_split.py /tmp/kt /tmp/kt.json [rows]
The window is cut out of the kernel TRACE: from the vert_blend_kernel (ra_set_frame: one per frame) that opens the N-th last frame to the one
that opens the last frame, N = the run's timed steps.  (Rounds 2-5 divided the whole run's stats by steps + warm-up + soak: the start-up's
uploads — ~40 buffer copies per replica — and the sequential leg's frames were counted into "per frame".)"""
import collections
import csv
import glob
import json
import re
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+(\w+?_kernel)', n)
    return m.group(1) if m else n


def main():
    d, j = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    line = json.loads(open(j).read().strip().splitlines()[-1])
    rows = []
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'vert_blend_kernel' in r['Kernel_Name']]
    n = min(line['steps'], len(marks) - 1)
    if n < 1:
        raise SystemExit('kernel_split: no frame boundaries (vert_blend_kernel) in the trace: was bench.py run with --static-frame?')
    # the sequential leg (if any) follows the timed region: take the window that ends where it starts
    seq = 0 if line.get('ms_per_step_sequential') is None else max(2, min(line['steps'], 10))
    end = len(marks) - 1 - seq
    a, b = marks[end - n], marks[end]
    win = rows[a:b]
    span = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / n / 1e6
    print('ms/step', round(line['ms_per_step'], 3), '| steady-state frames in the window', n, '| window ms/frame', round(span, 3))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in win:
        k = short(r['Kernel_Name'])
        agg[k][0] += 1
        agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(v[1] for v in agg.values())
    copies = sum(v[0] for k, v in agg.items() if 'copyBuffer' in k)
    torch_k = sum(v[0] for k, v in agg.items() if k.startswith('at::') or 'at::native' in k)
    print('GPU-busy ms/frame', round(tot / n / 1e6, 3), '| kernels/frame', round(len(win) / n, 1), '| buffer copies/frame', round(copies / n, 1),
          '| host-framework (at::) kernels/frame', round(torch_k / n, 1))
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'{k[:56]:56s} calls/frame {c / n:6.1f}  avg_us {ns / c / 1e3:8.1f}  ms/frame {ns / n / 1e6:6.3f}')


if __name__ == '__main__':
    main()
