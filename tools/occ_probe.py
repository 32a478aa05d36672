import os, sys
sys.path.insert(0, '/root/repo')
import torch
from xmem2_amd import ops
from xmem2_amd.ops import ConvWeights
torch.manual_seed(0)
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
# long-K 1x1 GEMMs so the steady-state loop dominates
for (H, W, Cin, Cout) in [(120, 216, 1024, 256), (60, 108, 1024, 512)]:
    x = torch.randn(1, H, W, Cin, device='cuda'); w = (torch.randn(Cout, 1, 1, Cin) * 0.05).cuda()
    cw = ConvWeights(w, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda(), 1, 0)
    fl = 2.0 * H * W * Cin * Cout
    for plan in (3, 6, 1, 4):
        us = t(lambda: ops.conv2d(x, cw, plan=(plan, 1)))
        print(f'pad={os.environ.get("XMEM_CONV_LDS_PAD", "0")} M={H*W} K={Cin} N={Cout} plan {plan}: {us:.1f} us {fl / us / 1e6:.1f} TF')
