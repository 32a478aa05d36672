"""Is the fp32 decoder sensitive to a bandwidth hog on another stream?  Two fp32 runs of the same clip, one with a side-stream
copy storm (or split GEMMs on private buffers) overlapping every 4th frame's decoder."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.ops import ConvWeights
from xmem2_amd.synth import synthetic_state_dict
dev = torch.device('cuda:0')
cfg = bench.b32_config()
sd = synthetic_state_dict(0)
frames, masks, _b, _n = bench.make_clip(bench.WORKLOADS['b32'])
fr, mk = torch.from_numpy(frames).to(dev), torch.from_numpy(masks).to(dev)
side = torch.cuda.Stream()
big_a = torch.randn(4, 120, 216, 256, device=dev); big_b = torch.empty_like(big_a)
x64 = torch.randn(4, 120, 216, 64, device=dev)
w = (torch.randn(256, 1, 1, 64) * 0.1).to(dev)
cw = ConvWeights(w, torch.ones(256, device=dev), torch.zeros(256, device=dev), 1, 0)
out_hog = torch.empty(4, 120, 216, 256, device=dev)
with ops.precision('fp32x'):
    ops.conv2d(x64, cw, out=out_hog)            # builds the split operands outside the streams' critical path
torch.cuda.synchronize()
def hog(kind):
    side.wait_stream(torch.cuda.current_stream()) if False else None
    with torch.cuda.stream(side), ops.ws_scope('@hog'):
        for _ in range(12):
            if kind == 'copy':
                big_b.copy_(big_a)
            elif kind == 'split':
                with ops.precision('fp32x'):
                    ops.conv2d(x64, cw, out=out_hog)
            elif kind == 'fp32conv':
                ops.conv2d(x64, cw, out=out_hog)
outs = {}
for mode in ('none', 'split'):
    net = XMem(dict(cfg), None).to(dev).eval(); net.load_weights(sd)
    core = InferenceCore(net, cfg); core.set_all_labels([1])
    for j in range(8):
        core.put_to_permanent_memory(fr[j], mk[j])
    res = []
    for i in range(12):
        if i % 4 == 0:
            core.prefetch_keys([fr[32 + (i + j) % 32] for j in range(4)])
        if mode != 'none' and i % 4 == 3:
            hog(mode)
        res.append(core.step(fr[32 + i % 32], None, None).clone())
    torch.cuda.synchronize()
    outs[mode] = res
for mode in ('split',):
    d = [float((a - b).abs().max()) for a, b in zip(outs['none'], outs[mode])]
    print(mode, ' '.join(f'{x:.1e}' for x in d))
