"""Print per-dispatch durations of kernels matching a substring from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2:]
for r in rows:
    n = r['Kernel_Name']
    if any(p in n for p in pat):
        print(f"{n[:60]:60s} grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us")
