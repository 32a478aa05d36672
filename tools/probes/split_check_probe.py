"""Every convolution of one B32 frame (eager, fp32x) against the same call in fp32: XMEM_SPLIT_CHECK=1 prints one line per call."""
import os, sys
os.environ['XMEM_SPLIT_CHECK'] = '1'
os.environ['XMEM_GUARD'] = '1'
os.environ['XMEM_HIP_GRAPHS'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict
dev = torch.device('cuda:0')
cfg = bench.b32_config()
net = XMem(dict(cfg, precision='fp32x'), None).to(dev).eval(); net.load_weights(synthetic_state_dict(0))
frames, masks, _b, _n = bench.make_clip(bench.WORKLOADS['b32'])
fr, mk = torch.from_numpy(frames).to(dev), torch.from_numpy(masks).to(dev)
core = InferenceCore(net, cfg); core.set_all_labels([1])
for j in range(2):
    core.put_to_permanent_memory(fr[j], mk[j])
core.step(fr[32], None, None)
print('--- batched key encoder (B=4) ---', file=sys.stderr)
img = torch.randn(4, 480, 864, 4, device=dev); img[..., 3] = 0
with ops.precision('fp32x'):
    net._encode_key_eager(img, True, True, False, True)
print('--- the same in fp32 ---', file=sys.stderr)
with ops.precision('fp32'):
    net._encode_key_eager(img, True, True, False, True)
