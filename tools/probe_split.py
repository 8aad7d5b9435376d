"""kernel trace of a run with RA_COARSE_PROBE=d: even calls of hdq_coarse_kernel are the ablated probe, odd calls the real launch."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'hdq_coarse' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
print('probe ms', sum(d[0::2]) / 1e6, 'real ms', sum(d[1::2]) / 1e6, 'calls', len(d))
