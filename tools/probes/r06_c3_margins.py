"""Round 6: where does the GPU path leave the oracle on bench.py's c3 stream?  Per frame: argmax mismatch GPU vs oracle(fp32, 1 thread),
both against the SAME oracle run in float64 (the exact answer of the reference's algorithm), and the oracle's top-2 margin at the
mismatching pixels.  Test infrastructure (imports the oracle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
torch.set_grad_enabled(False)
import clip_util as U
from oracle import cpu_ref
from xmem2_amd.network import XMem
from xmem2_amd.synth import synthetic_state_dict

clip = U.c3_bench_clip(int(os.environ.get('PROBE_STEPS', '25')))
sd = synthetic_state_dict(0, conditioning='multi_object')
t0 = time.time()
o1, p1, s1 = U.run_oracle(cpu_ref.RefNet(sd), clip, 1)
print(f'oracle fp32 1 thread: {time.time() - t0:.0f} s', flush=True)
o8, p8, _ = U.run_oracle(cpu_ref.RefNet(sd), clip, 8)
# the float64 oracle: same code, every tensor a double (Tensor.float() is redirected for the duration of the run)
t0 = time.time()
_float = torch.Tensor.float
torch.Tensor.float = lambda self, *a, **k: self.double()
torch.set_default_dtype(torch.float64)
try:
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    clip64 = U.Clip(clip.name + '_f64', clip.cfg, clip.frames.double(), clip.masks.double(), clip.labels, clip.perm_frames, clip.mask_frames,
                    clip.first_step, clip.key_batch, clip.end_flag)
    o64, p64, s64 = U._run_oracle(cpu_ref.RefNet(sd64), clip64, 32)
finally:
    torch.Tensor.float = _float
    torch.set_default_dtype(torch.float32)
print(f'oracle fp64: {time.time() - t0:.0f} s; dtype {p64[0].dtype}', flush=True)
net = XMem({'key_dim': 64, 'value_dim': 512, 'hidden_dim': 64}, None).to('cuda').eval()
net.load_weights(sd)
a, p, s = U.run_gpu(net, clip)
assert s == s1 == s64
lab = clip.labels
print('clip level:')
print('   oracle(8)  vs oracle(1):', U.fmt(U.compare(o8, o1, lab)))
print('   HIP        vs oracle(1):', U.fmt(U.compare(a, o1, lab)))
print('   oracle(1)  vs oracle f64:', U.fmt(U.compare(o1, o64, lab)))
print('   oracle(8)  vs oracle f64:', U.fmt(U.compare(o8, o64, lab)))
print('   HIP        vs oracle f64:', U.fmt(U.compare(a, o64, lab)))
print('frame: px HIP!=o1 | o1!=f64 | HIP!=f64 | o8!=o1 ; max|p-p64| HIP, o1 ; margins (oracle f64 top-2) at HIP!=f64 pixels')
for i in range(len(a)):
    t2 = torch.topk(p64[i], 2, dim=0).values
    mg = (t2[0] - t2[1]).numpy()
    d_g = a[i] != o64[i]
    e_g = float((p[i].double() - p64[i]).abs().max()); e_1 = float((p1[i].double() - p64[i]).abs().max())
    q = np.sort(mg[d_g])
    print(f'{i + 1:3d}: {int((a[i] != o1[i]).sum()):4d} | {int((o1[i] != o64[i]).sum()):4d} | {int(d_g.sum()):4d} | {int((o8[i] != o1[i]).sum()):4d} ; {e_g:.2e} {e_1:.2e} ; '
          + (f'median {np.median(q):.1e} max {q.max():.1e}' if q.size else '-'))
