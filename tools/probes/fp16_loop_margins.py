"""Are the fp16 loop's argmax differences on the 3-object stream near-ties?  Probability error and top-2 margins of the fp32 path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_grad_enabled(False)
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict
dev = torch.device('cuda:0')
sd = synthetic_state_dict(0)
for wlk in ('c3', 'b32'):
    wl = dict(bench.WORKLOADS[wlk])
    frames, masks, base, nq = bench.make_clip(wl)
    fr, mk = torch.from_numpy(frames).to(dev), torch.from_numpy(masks).to(dev)
    probs = {}
    for prec in ('fp32', 'fp16'):
        cfg = bench.workload_config(dict(wl, mem_every=10 ** 9))
        net = XMem(dict(cfg, precision=prec), None).to(dev).eval(); net.load_weights(sd)
        core = InferenceCore(net, cfg); core.set_all_labels(list(range(1, wl['K'] + 1)))
        for j in range(min(wl['perm'], 4)):
            core.put_to_permanent_memory(fr[j], mk[j])
        probs[prec] = [core.step(fr[base + i], None, None).clone() for i in range(4)]
    for i in range(4):
        a, b = probs['fp32'][i], probs['fp16'][i]
        top2 = torch.topk(a, 2, dim=0).values
        margin = top2[0] - top2[1]
        diff = a.argmax(0) != b.argmax(0)
        dp = (a - b).abs()
        print(f'{wlk} frame {i}: mean |dp| {float(dp.mean()):.2e} max {float(dp.max()):.2e}; fp32 top-2 margin < 1e-2 on {float((margin < 1e-2).float().mean()):.3%} of the pixels, '
              f'< 5e-2 on {float((margin < 5e-2).float().mean()):.3%}; argmax differs on {float(diff.float().mean()):.3%}, of which at margin > 5e-2: {int((diff & (margin > 5e-2)).sum())} px, > 1e-1: {int((diff & (margin > 1e-1)).sum())} px')
