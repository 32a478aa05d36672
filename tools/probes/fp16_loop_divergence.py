"""Where does the fp16 loop leave the fp32 path on a multi-object stream?  Per-frame argmax mismatch between the two GPU modes on the
bench's C3 workload (3 objects, 1 permanent frame, mem_every=5) and with variations (no memory frames, memory values from the fp32
path) that separate the stages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.set_grad_enabled(False)
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict

dev = torch.device('cuda:0')
wl = dict(bench.WORKLOADS['c3'])
sd = synthetic_state_dict(0)
frames, masks, base, nq = bench.make_clip(wl)
fr, mk = torch.from_numpy(frames).to(dev), torch.from_numpy(masks).to(dev)


def run(prec, mem_every, n=30, prefetch=True):
    cfg = bench.workload_config(dict(wl, mem_every=mem_every))
    net = XMem(dict(cfg, precision=prec), None).to(dev).eval(); net.load_weights(sd)
    core = InferenceCore(net, cfg); core.set_all_labels([1, 2, 3])
    core.put_to_permanent_memory(fr[0], mk[0])
    out = []
    for i in range(n):
        if prefetch and i % 4 == 0:
            core.prefetch_keys([fr[base + (i + j) % nq] for j in range(4)])
        p = core.step(fr[base + i % nq], None, None)
        out.append(ops.argmax_u8(p).cpu().numpy())
    return out


for me, pf in ((5, True), (5, False), (10 ** 9, True), (2, True)):
    a, b = run('fp32', me, prefetch=pf), run('fp16', me, prefetch=pf)
    c = run('fp32x', me, prefetch=pf)
    print(f'mem_every={me} prefetch={pf}: fp16-vs-fp32 mismatch per frame', ' '.join(str(int((x != y).sum())) for x, y in zip(a, b)))
    print(f'                                fp32x-vs-fp32               ', ' '.join(str(int((x != y).sum())) for x, y in zip(a, c)))
    sys.stdout.flush()
