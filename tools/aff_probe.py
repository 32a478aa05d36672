"""Run the affinity op alone at the B32 size (N=51840, HW=1620, k=30) - for rocprofv3 PMC passes and timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmem2_amd import ops
torch.manual_seed(0)
n, hw = 51840, 1620
mk = (torch.randn(n, 64) * 0.9).cuda(); ms = (torch.rand(n) * 3 + 1).cuda()
qk = (torch.randn(hw, 64) * 0.9).cuda(); qe = (torch.rand(hw, 64) * 0.9 + 0.05).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    ops.affinity_topk([(mk, ms)], qk, qe, 30)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.affinity_topk([(mk, ms)], qk, qe, 30)
e1.record(); e1.synchronize()
print(f'affinity_topk N={n} HW={hw}: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us per call')
