"""Key encoder at batch 1 / 2 / 4 / 8 frames (HIP-graph replay): ms per FRAME."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
import bench
from xmem2_amd import XMem, ops
from xmem2_amd.synth import synthetic_state_dict
net = XMem(dict(bench.b32_config()), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
for B in (1, 2, 4, 8):
    img = torch.randn(B, 480, 864, 4, device='cuda')
    net.use_graphs = False
    net.encode_key_nhwc(img)                   # eager pass: autotunes unseen shapes
    net.use_graphs = True
    for _ in range(3):
        net.encode_key_nhwc(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        net.encode_key_nhwc(img)
    e1.record(); e1.synchronize()
    print(f'B={B}: {e0.elapsed_time(e1) / 20:.3f} ms per call, {e0.elapsed_time(e1) / 20 / B:.3f} ms per frame')
ops.dump_tuned_plans('gpurun_out/conv_plans_keybatch.json')
