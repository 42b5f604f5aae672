"""Timeline of ONE training step from a rocprofv3 kernel_trace.csv: per kernel start offset, duration, queue; idle gaps
of the union of all queues.  argv: kernel_trace.csv [step index from the end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_fused' in r['Kernel_Name']]
a, b = idx[-back - 1] + 1, idx[-back] + 1
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
end_prev = t0
busy_until = t0
idle = 0
print(f'{"start us":>9s} {"dur us":>8s} {"gap":>7s} q  kernel')
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = max(0, s - busy_until)
    idle += gap
    busy_until = max(busy_until, e)
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:7.1f} {r.get("Queue_Id", "?"):>2s} {r["Kernel_Name"][:80]}')
print(f'step span {(busy_until - t0) / 1e3:.1f} us, idle (no kernel on any queue) {idle / 1e3:.1f} us, kernels {len(step)}, '
      f'sum of durations {sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3:.1f} us')
