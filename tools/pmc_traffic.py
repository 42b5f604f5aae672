"""Assemble profiles/round3/pmc_traffic.json from rocprofv3 --pmc counter_collection.csv files.
argv: out.json  name=kernel_substr:alg_bytes:fetch.csv:write.csv[:extra_substr_for_sum] ...
FETCH_SIZE / WRITE_SIZE are KB per dispatch; median over the dispatches after the first two (warm-up); FETCH doubled
(gfx950 under-reports wide coalesced reads by 2x: MI355X_MICROARCH.md, HBM section): bytes = (2 * FETCH + WRITE) * 1024.
A ':+substr' suffix adds the per-dispatch counters of a second kernel (the split-K reduce pass) to the launch."""
import csv, json, statistics, sys


def per_kernel(path, counter, pat):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path)) if pat in r['Kernel_Name'] and r['Counter_Name'] == counter]
    vals = vals[2:] if len(vals) > 4 else vals
    return statistics.median(vals) if vals else None


out = {'_how': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, program directly after --) on tools/run_gru_fwd.py (BWD=1: one '
               'bidirectional H = 512 layer, 2048 trials, 20 steps) and tools/run_wgrad_group.py (configs[1] layer-1 group); KB per dispatch, '
               'median; FETCH_SIZE doubled (gfx950 correction); bytes = (2 * FETCH + WRITE) * 1024; round-3 kernels'}
for spec in sys.argv[2:]:
    name, rest = spec.split('=', 1)
    parts = rest.split(':')
    pat, alg, fcsv, wcsv = parts[0], int(parts[1]), parts[2], parts[3]
    extra = parts[4][1:] if len(parts) > 4 else None
    f, w = per_kernel(fcsv, 'FETCH_SIZE', pat), per_kernel(wcsv, 'WRITE_SIZE', pat)
    if f is None or w is None:          # (kernel pattern not in the trace: no entry, no TypeError -- ADVICE r3)
        print(f'pmc_traffic: no dispatch matches {pat!r} in {fcsv} / {wcsv}', file=sys.stderr)
        continue
    if extra:
        f += per_kernel(fcsv, 'FETCH_SIZE', extra) or 0.0
        w += per_kernel(wcsv, 'WRITE_SIZE', extra) or 0.0
    hbm = int((2 * f + w) * 1024)
    out[name] = {'FETCH_SIZE_KB': f, 'WRITE_SIZE_KB': w, 'hbm_bytes_per_launch': hbm, 'algorithmic_bytes': alg, 'ratio': round(hbm / alg, 3)}
json.dump(out, open(sys.argv[1], 'w'), indent=1)
print(json.dumps(out, indent=1))
