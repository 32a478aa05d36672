"""Run one 3x3 layer with one plan N times (for rocprofv3 --pmc on a single kernel shape).
usage: one_conv.py H W Cin Cout tile splitk [reps]      (ONE_CONV_BATCH=n: batch size)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmem2_amd import ops
from xmem2_amd.ops import ConvWeights
H, W, Cin, Cout, tile, sk = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
torch.manual_seed(0)
x = torch.randn(int(os.environ.get('ONE_CONV_BATCH', '1')), H, W, Cin, device='cuda')
w = (torch.randn(Cout, 3, 3, Cin) * 0.05).cuda()
cw = ConvWeights(w, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda(), 1, 1)
for _ in range(reps):
    ops.conv2d(x, cw, plan=(tile, sk))
torch.cuda.synchronize()
