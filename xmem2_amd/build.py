"""Build the gfx950 shared library (plain hipcc, no torch headers - the boundary is a C ABI).

``python -m xmem2_amd.build`` or ``__graft_entry__.build()``.  The library is built IN-TREE
(xmem2_amd/csrc/libxmem_hip.so) so that it travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
SOURCES = ['conv_mfma.hip', 'elementwise.hip', 'affinity.hip', 'affinity_filter.hip', 'consolidate.hip', 'selector.hip']
HEADERS = ['common.hpp', 'affinity_common.hpp']
LIB = os.path.join(CSRC, 'libxmem_hip.so')
ARCH = 'gfx950'


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the MI355X kernels cannot be built (no CPU fallback exists)')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(os.path.dirname(CSRC), '..', 'include', 'xmem_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def source_digest():
    """sha256 over the kernel sources + header: profiles recorded for one build are only quoted against the same build."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    with open(os.path.join(os.path.dirname(CSRC), '..', 'include', 'xmem_hip.h'), 'rb') as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(obj)
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c',
               os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print('built', LIB)
