"""'fp32x' (split-operand) vs fp32 on representative layers: error against a fp64 convolution and time per plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from xmem2_amd import ops
from xmem2_amd.ops import ConvWeights
torch.manual_seed(0)
LAYERS = [(1, 120, 216, 256, 256, 3, 1), (1, 60, 108, 512, 512, 3, 1), (1, 30, 54, 1600, 512, 3, 1), (1, 30, 54, 512, 512, 3, 1),
          (1, 30, 54, 1024, 256, 1, 1), (1, 30, 54, 256, 1024, 1, 1), (1, 120, 216, 64, 256, 1, 1), (1, 120, 216, 256, 64, 1, 1),
          (4, 120, 216, 64, 64, 3, 1), (1, 120, 216, 128, 128, 3, 2)]
quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (B, H, W, Cin, Cout, k, st) in LAYERS:
    x = F.relu(torch.randn(B, Cin, H, W))
    w = torch.randn(Cout, Cin, k, k) * (2.0 / (Cin * k * k)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, st, k // 2) if not quick or H * W < 8000 else None
    cw = ConvWeights(w.permute(0, 2, 3, 1).contiguous().cuda(), torch.ones(Cout).cuda(), torch.zeros(Cout).cuda(), st, k // 2)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    scale = float(ref.abs().max()) if ref is not None else 1.0
    plans = [(1, 1), (2, 1), (3, 1)] + ([(7, 1), (8, 1), (9, 1), (17, 1), (18, 1), (19, 1)] if (k == 3 and st == 1) else [])
    line = []
    for plan in plans:
        res = {}
        for mode in ('fp32', 'fp32x'):
            with ops.precision(mode):
                try:
                    out = ops.conv2d(xin, cw, plan=plan)
                    us = timeit(lambda: ops.conv2d(xin, cw, plan=plan))
                    err = float((out.permute(0, 3, 1, 2).double().cpu() - ref).abs().max()) / scale if ref is not None else float('nan')
                    res[mode] = (us, err)
                except RuntimeError as e:
                    res[mode] = (float('nan'), float('nan'))
        line.append(f'plan{plan[0]:2d}: {res["fp32"][0]:6.1f}us/{res["fp32"][1]:.1e} -> {res["fp32x"][0]:6.1f}us/{res["fp32x"][1]:.1e}')
    print(f'{B}x{H}x{W} {Cin}->{Cout} k{k}s{st}:\n   ' + '\n   '.join(line), flush=True)
