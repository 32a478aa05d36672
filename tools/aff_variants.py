"""Timing experiment: where does the wide select kernel lose MFMA issue slots?  (XMEM_AFF_DEBUG_VARIANT bit 1: no filter,
bit 2: no global loads in the loop, bit 4: constant LDS operand addresses).  Results are wrong for non-zero variants."""
import os, sys, subprocess
if len(sys.argv) == 1:
    for v in (0,):
        env = dict(os.environ, XMEM_AFF_DEBUG_VARIANT=str(v))
        subprocess.run([sys.executable, os.path.abspath(__file__), str(v)], env=env)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmem2_amd import ops
torch.set_grad_enabled(False)
g = torch.Generator(device='cuda').manual_seed(1)
n, hw = 51840, 1620
mk = torch.randn(n, 64, generator=g, device='cuda') * 0.9
ms = torch.rand(n, generator=g, device='cuda') * 3 + 1
qk = torch.randn(hw, 64, generator=g, device='cuda') * 0.9
qe = torch.rand(hw, 64, generator=g, device='cuda') * 0.9 + 0.05
segs = [(mk, ms)]
w, idx, _ = ops.affinity_topk(segs, qk, qe, 30)        # valid hint from a correct run? (library reads the env once per process)
hint = (idx, [n], 54)
def run():
    ops.affinity_topk(segs, qk, qe, 30, hint=hint)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); e1.synchronize()
print(f'variant {sys.argv[1]}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per hinted affinity call (random keys)')
