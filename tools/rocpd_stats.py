#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / min / max / %."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in cur.execute(f'pragma table_info({kd})')]
    scol = [r[1] for r in cur.execute(f'pragma table_info({ks})')]
    name_col = 'kernel_name' if 'kernel_name' in scol else 'display_name'
    q = f'select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc'
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"']
    for n, c, t, mn, mx in rows:
        lines.append(f'"{n[:110]}",{c},{t},{t / c:.0f},{mn},{mx},{100 * t / tot:.2f}')
    txt = '\n'.join(lines)
    if out:
        open(out, 'w').write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
