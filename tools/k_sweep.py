"""Fixed cost vs per-K cost of the implicit-GEMM kernel on small 1x1 layers: sweep Cin for fixed M, Cout and tile plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmem2_amd import ops
from xmem2_amd.ops import ConvWeights
torch.manual_seed(0)
names = {1: '128x128/32', 2: '128x64/32', 3: '64x64/32', 4: '128x128/64', 5: '128x64/64', 6: '64x64/64'}
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (H, W, Cout) in [(60, 108, 512), (30, 54, 1024), (30, 54, 256), (120, 216, 256)]:
    for res in (False, True):
        for plan in (3, 6, 2):
            line = []
            for Cin in (32, 64, 128, 256, 512, 1024):
                x = torch.randn(1, H, W, Cin, device='cuda')
                w = (torch.randn(Cout, 1, 1, Cin) * 0.05).cuda()
                cw = ConvWeights(w, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda(), 1, 0)
                r = torch.randn(1, H, W, Cout, device='cuda') if res else None
                us = t(lambda: ops.conv2d(x, cw, res=r, relu_out=True, plan=(plan, 1)))
                line.append(f'K{Cin}={us:.1f}')
            print(f'M={H*W} N={Cout} res={int(res)} {names[plan]}: ' + ' '.join(line))
# empty-ish kernel launch floor
x = torch.zeros(64, device='cuda')
print('tiny torch kernel', t(lambda: x.add_(1.0)))
