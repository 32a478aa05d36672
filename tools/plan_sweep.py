"""Time every conv plan for a few representative layers (same box, back to back)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmem2_amd import ops
from xmem2_amd.ops import ConvWeights
torch.manual_seed(0)
LAYERS = [(1, 120, 216, 256, 256, 3), (1, 60, 108, 512, 256, 3), (1, 30, 54, 1600, 512, 3), (1, 30, 54, 512, 512, 3),
          (1, 30, 54, 256, 256, 3), (1, 30, 54, 1024, 256, 1), (1, 30, 54, 256, 1024, 1), (1, 60, 108, 128, 512, 1), (1, 120, 216, 64, 256, 1)]
names = {1: '128x128/32', 2: '128x64/32', 3: '64x64/32', 4: '128x128/64', 5: '128x64/64', 6: '64x64/64', 7: 'F128x64', 8: 'F64x64', 9: 'F64x128'}
for (B, H, W, Cin, Cout, k) in LAYERS:
    x = torch.randn(B, H, W, Cin, device='cuda')
    w = (torch.randn(Cout, k, k, Cin) * 0.05).cuda()
    cw = ConvWeights(w, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda(), 1, k // 2)
    flop = 2.0 * B * H * W * Cout * k * k * Cin
    res = []
    plans = [(t, s) for t in range(1, 7) for s in (1, 2, 4, 8)] + ([(t, 1) for t in range(7, 16)] if cw.wu is not None else [])
    for plan in plans:
        try:
            for _ in range(2):
                ops.conv2d(x, cw, plan=plan)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.conv2d(x, cw, plan=plan)
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 8
            res.append((us, plan))
        except RuntimeError as e:
            pass
    res.sort()
    print(f'M={B*H*W} Cin={Cin} Cout={Cout} k={k}:  ' + '  '.join(
        f"{('W-' + names[p[0]-6]) if p[0] > 6 else names[p[0]]}x{p[1]}={us:.0f}us({flop/us/1e6:.0f}TF)" for us, p in res[:7]))
