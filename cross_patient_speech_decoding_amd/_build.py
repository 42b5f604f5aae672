"""Build libxps.so (all HIP sources under csrc/) for gfx950 with hipcc, in-tree.

One object per source (compiled in parallel, re-used while the source and every header are older), one link:
a one-file edit of the 14 sources rebuilds in the time of that file."""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libxps.so')
OBJ = os.path.join(PKG, 'build')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def hipcc_path():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: libxps.so cannot be built')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _headers():
    return glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(PKG, '..', 'include', 'xps.h')]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def _obj_path(src, tag):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + (('.' + tag) if tag else '') + '.o')


def build(force=False, verbose=True, defines=(), out=None, jobs=None):
    """hipcc --offload-arch=gfx950: one object per source, one -shared link -> one C-ABI library.
    defines / out: diagnostic variants (e.g. defines=('XPS_CL_STAMP',), out='libxps_stamp.so') beside the product build."""
    lib_out = LIB if out is None else os.path.join(PKG, out)
    if not force and out is None and not needs_build():
        return lib_out
    os.makedirs(OBJ, exist_ok=True)
    hipcc = hipcc_path()
    tag = '_'.join(d.replace('=', '-') for d in defines)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    todo = []
    for s in sources():
        o = _obj_path(s, tag)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            todo.append((s, o))

    def compile_one(so):
        cmd = [hipcc] + FLAGS + ['-D' + d for d in defines] + ['-c', '-o', so[1], so[0]]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, todo))
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', lib_out] + [_obj_path(s, tag) for s in sources()]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_out


if __name__ == '__main__':
    defs = tuple(a[2:] for a in sys.argv[1:] if a.startswith('-D'))
    outs = [a[6:] for a in sys.argv[1:] if a.startswith('--out=')]
    build(force='--force' in sys.argv or not defs, defines=defs, out=outs[0] if outs else None)
