"""Average PMC counter values per kernel from rocprofv3 counter_collection.csv files.  argv: csv... --kernel substr"""
import csv, sys, collections
files = [a for a in sys.argv[1:] if a.endswith('.csv')]
pat = sys.argv[sys.argv.index('--kernel') + 1] if '--kernel' in sys.argv else ''
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    v = v[len(v) // 2:]                      # drop warm-up launches
    print(f'{k:36s} {sum(v) / len(v):16.1f}  (n={len(v)})')
