"""Build libxps.so (all HIP sources under csrc/) for gfx950 with hipcc, in-tree."""
import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libxps.so')


def hipcc_path():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: libxps.so cannot be built')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(PKG, '..', 'include', 'xps.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -shared: one code object, one C-ABI library."""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-o', LIB] + sources()
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force=True)
