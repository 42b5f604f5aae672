"""Per-step statistics from a rocprofv3 kernel_trace.csv of tools/graph_probe.py: steps are delimited by the AdamW kernel; for the
eager steps and for the graph replays: span, sum of kernel durations, number of kernels, idle time on the union of queues, queues used."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_fused' in r['Kernel_Name']]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]['Start_Timestamp'])
    busy, idle = t0, 0
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        idle += max(0, s - busy); busy = max(busy, e)
    steps.append(dict(span=(busy - t0) / 1e3, ksum=sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3, n=len(seg), idle=idle / 1e3,
                      queues=len({r.get('Queue_Id') for r in seg}), gap_prev=(t0 - int(rows[a]['End_Timestamp'])) / 1e3))
def show(name, ss):
    if not ss: return
    print(f'{name:10s} n={len(ss):4d}  span {statistics.median(s["span"] for s in ss):8.1f} us  kernel-sum {statistics.median(s["ksum"] for s in ss):8.1f}  kernels {statistics.median(s["n"] for s in ss):5.0f}  '
          f'idle {statistics.median(s["idle"] for s in ss):7.1f}  queues {statistics.median(s["queues"] for s in ss):.0f}  gap-to-previous-step {statistics.median(s["gap_prev"] for s in ss):6.1f}')
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
# order in graph_probe.py: prewarm + NT eager, 3 side-stream warm-ups + capture (no kernels), 20 + NT replays, NT eager
show('last eager', steps[-k + 5:])
show('replays', steps[-2 * k + 5:-k - 5])
show('first eager', steps[-3 * k - 20:-2 * k - 30])
seg = None
