"""Build the gfx950 shared library (plain hipcc, no torch headers - the boundary is a C ABI).

``python -m xmem2_amd.build`` or ``__graft_entry__.build()``.  The library is built IN-TREE
(xmem2_amd/csrc/libxmem_hip.so) so that it travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
SOURCES = ['conv_mfma.hip', 'gemm_stream.hip', 'elementwise.hip', 'affinity.hip', 'affinity_filter.hip', 'consolidate.hip', 'selector.hip', 'augment.hip']
HEADERS = ['common.hpp', 'affinity_common.hpp', 'gemm_stream.hpp']
LIB = os.path.join(CSRC, 'libxmem_hip.so')
ARCH = 'gfx950'
EXTRA_FLAGS = {'augment.hip': ['-ffp-contract=off']}      # per-source flags: the augmentation kernel rounds where the host libraries round
# NO PACKED-FP32 VALU INSTRUCTIONS (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any kernel of this library.  Measured on MI355X
# (round 3, tools/probes/victim_decoder_probe.py, hog_probe.py): an elementwise kernel built with them returned wrong values
# in ~1e-4 of its elements (one 32-bit half of a packed result) WHILE a kernel issuing v_mfma_f32_32x32x16_f16 ran on another
# stream - the memory-readout filter and the split-operand GEMMs do exactly that next to the side-stream key encoder.  With
# the subtarget feature turned off the same runs are bit-identical to the solo runs.  (The flag reaches the host compilation
# too, which prints an 'ignoring feature' note - filtered below.)
DEVICE_FLAGS = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)"


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
    h.update(' '.join(DEVICE_FLAGS).encode())
    for src in sorted(EXTRA_FLAGS):
        h.update((src + ':' + ' '.join(EXTRA_FLAGS[src])).encode())
    h.update(os.environ.get('XMEM_HIPCC_FLAGS', '').encode())
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
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC'] + DEVICE_FLAGS + EXTRA_FLAGS.get(src, []) + \
              os.environ.get('XMEM_HIPCC_FLAGS', '').split() + ['-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        out = '\n'.join(l for l in out.splitlines() if _NOISE not in l and l.strip())
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
