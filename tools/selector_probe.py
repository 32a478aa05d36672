"""Time xmem_cycle_dissimilarity at 480p (HW = 1620) for F candidate frames."""
import sys, time
import torch
sys.path.insert(0, '.')
from xmem2_amd import ops
F_, HW, CK = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 1620, 64
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
key = (torch.randn(F_, HW, CK, generator=g) * 0.5).to(dev)
sel = torch.rand(F_, HW, CK, generator=g).to(dev)
shr = (1 + torch.rand(F_, HW, generator=g)).to(dev)
Mexp = torch.empty(F_, HW, 2 * CK, device=dev); Qexp = torch.empty_like(Mexp); bsq = torch.empty(F_, HW, device=dev)
pres = torch.zeros(F_, dtype=torch.int32, device=dev)
for i in range(F_):
    ops.selector_prepare(key[i], sel[i], None, 30, 54, 0.5, 0.5, Mexp[i], Qexp[i], bsq[i], pres[i:i + 1])
for _ in range(2):
    out = ops.cycle_dissimilarity(Mexp, Qexp, bsq, shr, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for i in range(n):
    out = ops.cycle_dissimilarity(Mexp, Qexp, bsq, shr, i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flops = F_ * 2 * 4 * CK * HW * HW
print(f'F={F_} HW={HW}: {ms:.3f} ms per launch, {flops / ms / 1e9:.1f} TFLOP/s (fp32 MFMA), out[:4]={out[:4].tolist()}')
