"""How tight is the hint bound (tau) against the exact k-th similarity, per query, on the served-size test data?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from xmem2_amd import ops
from xmem2_amd._lib import load
import test_gpu_affinity_served_sizes as Tt
n, hw, gw, nseg = Tt.SIZES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
mk, ms, qk, qe, cuts = Tt._make(n, hw, nseg, seed=n)
segs = [(mk[a:b], ms[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
sizes = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
w0, i0, s0 = ops.affinity_topk(segs, qk, qe, 30, want_sim=True)
w, i, sv = ops.affinity_topk(segs, qk, qe, 30, want_sim=True, hint=(i0, sizes, gw))
o = [C.c_size_t(), C.c_size_t(), C.c_size_t()]
load().xmem_affinity_debug_offsets(n, hw, *[C.byref(x) for x in o])
torch.cuda.synchronize()
ws = ops.workspace(0, qk.device, 'affinity')
tau = ws[o[2].value:o[2].value + 4 * hw].view(torch.float32).cpu()
cnt = ws[o[0].value:o[0].value + 4 * hw].view(torch.int32).cpu()
kth = s0[:, 29].cpu()
gap = kth - tau
print('gap kth - tau: min %.4f median %.4f max %.4f' % (gap.min(), gap.median(), gap.max()))
big = torch.nonzero(gap > 0.5).flatten()
print('queries with gap > 0.5:', big.numel(), big[:40].tolist())
print('planted (q % 7 == 0) among them:', int((big % 7 == 0).sum()))
print('cnt: median', int(cnt.median()), 'max', int(cnt.max()), 'argmax', int(cnt.argmax()))
print('kth stats', float(kth.min()), float(kth.median()), float(kth.max()), ' top1', float(s0[:, 0].max()))
