import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:3]:
    print(r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, 'ms total')
