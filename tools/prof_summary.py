"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into the markdown table kept
under profiles/.  usage: prof_summary.py <kernel_stats.csv> <steps_total> [title]"""
import csv
import sys


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else 'rocprofv3 --kernel-trace --stats'
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    print(f'# {title}\n')
    print(f'{steps} steps profiled; {tot / 1e6 / steps:.3f} ms of kernel time per step, {calls / steps:.0f} launches per step.\n')
    print('| kernel | calls | us per step | avg us | % |')
    print('|---|---|---|---|---|')
    for r in rows:
        t = float(r['TotalDurationNs'])
        if t / tot < 0.002:
            continue
        print(f"| {r['Name'][:90]} | {r['Calls']} | {t / 1e3 / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * t / tot:.2f} |")


if __name__ == '__main__':
    main()
