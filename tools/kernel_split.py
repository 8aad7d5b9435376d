"""Per-frame kernel table of one profiled bench.py run.
usage (on the GPU box): cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 bench.py ARGS > /tmp/kt.json
                        python3 tools/kernel_split.py /tmp/kt /tmp/kt.json [rows]"""
import csv
import glob
import json
import sys


def main():
    d, j = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    line = json.loads(open(j).read().strip().splitlines()[-1])
    frames = line['steps'] + line['warmup'] + line.get('config', {}).get('soak_frames', 0)
    print('ms/step', round(line['ms_per_step'], 3), '| frames profiled', frames)
    rows = []
    for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('GPU-busy ms/frame', round(tot / frames / 1e6, 3), '| kernels/frame', round(sum(int(r['Calls']) for r in rows) / frames, 1))
    for r in rows[:top]:
        n = r['Name']
        n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('_ZN12_GLOBAL__N_1', '')
        calls, avg, ms = int(r['Calls']) / frames, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / frames / 1e6
        print(f'{n[:56]:56s} calls/frame {calls:6.1f}  avg_us {avg:8.1f}  ms/frame {ms:6.3f}')


if __name__ == '__main__':
    main()
