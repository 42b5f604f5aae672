"""Interleaved A/B of the headline step under environment switches: python tools/ab_env.py VAR=a,b [VAR2=c,d ...] [--steps N] [--rounds R]
Every combination is run `rounds` times in turn (own process each: the switches are read at first use), ms/step printed per run."""
import itertools, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps, rounds, sw = 40, 2, []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == '--steps': steps = int(args.pop(0))
    elif a == '--rounds': rounds = int(args.pop(0))
    else:
        k, v = a.split('=')
        sw.append((k, v.split(',')))
combos = list(itertools.product(*[v for _, v in sw]))
res = {c: [] for c in combos}
for r in range(rounds):
    for c in combos:
        env = dict(os.environ, **{k: val for (k, _), val in zip(sw, c)})
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', str(steps), '--warmup', '5', '--headline-only', '--no-cpu-baseline'],
                             env=env, capture_output=True, text=True)
        try:
            res[c].append(json.loads(out.stdout.strip().splitlines()[-1])['ms_per_step'])
        except Exception:
            res[c].append(float('nan'))
            print(out.stdout[-500:], out.stderr[-1500:])
        print(dict(zip([k for k, _ in sw], c)), res[c], flush=True)
