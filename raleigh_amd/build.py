"""Builds librlhip.so (gfx950) in-tree with hipcc.

``python -m raleigh_amd.build`` or ``raleigh_amd.build.build_library()``.
hipcc cross-compiles without a GPU; the resulting .so is git-ignored but
travels with the tree to the GPU box.
"""

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIBDIR = os.path.join(PKG, 'lib')
LIBPATH = os.path.join(LIBDIR, 'librlhip.so')
SOURCES = ['context', 'gram', 'update', 'spmm', 'spmm_wide_build', 'spmm_wide_s', 'spmm_wide_d', 'spmm_wide_c',
           'spmm_wide_z', 'sptrsv', 'dense']
HOST_SOURCES = ['ldlt_host', 'shm_reduce']        # plain C++ (host only): compiled by the same driver, no offload
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-result']


HASHPATH = LIBPATH + '.srchash'


def source_hash():
    """sha256 over the compile flags and every file the library is built from (file name + bytes)."""
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC))
    files.append(os.path.join(ROOT, 'include', 'rlhip.h'))
    h = hashlib.sha256(' '.join(FLAGS + SOURCES + HOST_SOURCES).encode())
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_up_to_date():
    """The shipped .so was built from exactly these sources and flags (hash recorded by the build that
    linked it; file times say nothing on a box that received the tree as a snapshot)."""
    if not (os.path.exists(LIBPATH) and os.path.exists(HASHPATH)):
        return False
    with open(HASHPATH) as fh:
        return fh.read().strip() == source_hash()


def build_library(force=False, verbose=False):
    """Compiles every csrc/*.hip for gfx950 and links lib/librlhip.so."""
    if not force and is_up_to_date():
        return LIBPATH
    if os.path.exists(HASHPATH):
        os.remove(HASHPATH)
    hipcc = os.environ.get('HIPCC', 'hipcc')
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    inc = ['-I', os.path.join(ROOT, 'include'), '-I', CSRC]

    def compile_one(name):
        host = name in HOST_SOURCES
        src = os.path.join(CSRC, name + ('.cpp' if host else '.hip'))
        obj = os.path.join(objdir, name + '.o')
        flags = ['-x', 'c++', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'include')] + [f for f in FLAGS if not f.startswith('--offload-arch')] if host else FLAGS
        cmd = [hipcc] + flags + inc + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES) + len(HOST_SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES + HOST_SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIBPATH] + objs + ['-lrt']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s' % r.stderr[-4000:])
    with open(HASHPATH, 'w') as fh:
        fh.write(source_hash() + '\n')
    return LIBPATH


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
