"""fp32x vs fp32 through InferenceCore (graphs + batched key hints) on the B32 clip: per-frame max |dprob| and argmax flips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict
dev = torch.device('cuda:0')
cfg = bench.b32_config()
P = int(os.environ.get('PERM', '8'))
sd = synthetic_state_dict(0)
frames, masks, _b, _n = bench.make_clip(bench.WORKLOADS['b32'])
fr, mk = torch.from_numpy(frames).to(dev), torch.from_numpy(masks).to(dev)
outs = {}
for mode in ('fp32', 'fp32x'):
    net = XMem(dict(cfg, precision=mode), None).to(dev).eval(); net.load_weights(sd)
    core = InferenceCore(net, cfg); core.set_all_labels([1])
    for j in range(P):
        core.put_to_permanent_memory(fr[j], mk[j])
    res = []
    use_pf = os.environ.get('NOPF') is None
    for i in range(12):
        if use_pf and i % 4 == 0:
            core.prefetch_keys([fr[32 + (i + j) % 32] for j in range(4)])
            if os.environ.get('SYNC') in ('1', 'pf'):
                torch.cuda.synchronize()
        res.append(core.step(fr[32 + i % 32], None, None).clone())
        if os.environ.get('SYNC') in ('1', 'step'):
            torch.cuda.synchronize()
    outs[mode] = res
for i, (a, b) in enumerate(zip(outs['fp32'], outs['fp32x'])):
    flips = int((a.argmax(0) != b.argmax(0)).sum())
    print(f'frame {i}: max |dprob| {float((a - b).abs().max()):.3e} mean {float((a - b).abs().mean()):.3e} argmax flips {flips}')
