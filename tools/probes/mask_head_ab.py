"""The mask head (3x3, 256 -> 1, relu_in, 1/4 resolution) one pixel per wave against four pixels of a row per wave:
   XMEM_COUT1_ROW4=0|1 python tools/probes/mask_head_ab.py [out.pt]   (prints the time; saves / compares the output bits)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xmem2_amd import ops
torch.manual_seed(0)
for (B, H, W) in ((1, 120, 216), (3, 120, 216), (1, 180, 320), (2, 37, 51)):
    x = torch.randn(B, H, W, 256, device='cuda')
    w = torch.randn(1, 256, 3, 3, device='cuda') * 0.05
    cw = ops.ConvWeights(w.permute(0, 2, 3, 1).contiguous(), torch.ones(1, device='cuda'), torch.tensor([0.3], device='cuda'), 1, 1)
    y = ops.conv2d(x, cw, relu_in=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for it in range(30):
        ev[0].record(); y = ops.conv2d(x, cw, relu_in=True); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    ts.sort()
    ref = torch.nn.functional.conv2d(torch.relu(x).permute(0, 3, 1, 2).double(), w.double(), torch.tensor([0.3], device='cuda').double(), padding=1)
    err = float((y.permute(0, 3, 1, 2).double() - ref).abs().max())
    tag = f'{B}x{H}x{W}'
    f = f'/tmp/mask_head_{tag}.pt'
    same = ''
    if os.path.exists(f):
        same = ' bit-identical to the other variant: ' + str(bool(torch.equal(torch.load(f), y.cpu())))
    else:
        torch.save(y.cpu(), f)
    print(f'XMEM_COUT1_ROW4={os.environ.get("XMEM_COUT1_ROW4", "1")} {tag}: median {ts[len(ts)//2]:.1f} us (min {ts[0]:.1f}); max |err| vs float64 {err:.2e};{same}', flush=True)
