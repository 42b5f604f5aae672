import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
for lib in sorted(f for f in os.listdir(here) if f.startswith('libxps_b')):
    env = dict(os.environ, XPS_LIB_OVERRIDE=os.path.join(here, lib))
    out = subprocess.run([sys.executable, os.path.join(here, 'bench_gemm.py')], env=env, capture_output=True, text=True).stdout
    print('==', lib); print('\n'.join(l for l in out.splitlines() if 'TF' in l and 'single' not in l))
