import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xmem2_amd import ops
from xmem2_amd._lib import load
gen = torch.Generator().manual_seed(93)
n, hw, gw, k = 12000, 260, 20, 30
mk = torch.randn(n, 64, generator=gen) * 0.9
ms = torch.rand(n, generator=gen) * 3 + 1
qk = torch.randn(hw, 64, generator=gen) * 0.9
qe = torch.rand(hw, 64, generator=gen) * 0.9 + 0.05
mk[::7] *= 400.0
qk[::5] *= 300.0
segs = [(mk[:5000].cuda(), ms[:5000].cuda()), (mk[5000:].cuda(), ms[5000:].cuda())]
sizes = [5000, n - 5000]
w0, i0, s0 = ops.affinity_topk(segs, qk.cuda(), qe.cuda(), k, want_sim=True)
w, i, sv = ops.affinity_topk(segs, qk.cuda(), qe.cuda(), k, want_sim=True, hint=(i0, sizes, gw))
torch.cuda.synchronize()
lib = load()
o = [C.c_size_t(), C.c_size_t(), C.c_size_t()]
lib.xmem_affinity_debug_offsets(n, hw, *[C.byref(x) for x in o])
ws = ops._workspaces[(str(w.device), 'affinity')]
cnt = ws[o[0].value:o[0].value + 4 * hw].view(torch.int32)
nt = (hw + 127) // 128
flg = ws[o[1].value:o[1].value + 8 * nt].view(torch.int32)
tau = ws[o[2].value:o[2].value + 4 * hw].view(torch.float32)
bad = (~((sv == s0).all(1) & (i == i0).all(1))).nonzero().flatten().tolist()
print('flags', flg.tolist(), 'bad queries', bad[:40], len(bad))
for q in bad[:6]:
    d = (sv[q] != s0[q]).nonzero().flatten().tolist()
    print(q, 'q%5', q % 5, 'cnt', int(cnt[q]), 'tau', float(tau[q]), 'kth ref', float(s0[q, -1]), 'first diff at', d[:3], 'got', sv[q, d[:3]].tolist(), 'ref', s0[q, d[:3]].tolist(), 'idx got', i[q, d[:3]].tolist(), 'ref', i0[q, d[:3]].tolist())
print('cnt stats', cnt.float().mean().item(), cnt.max().item(), 'n >= lcap', int((cnt >= 2048).sum()))
