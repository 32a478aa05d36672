"""Per-launch timing of one B32 frame (every conv2d / affinity call, back-to-back repetitions) on the GPU."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench

cfg = bench.b32_config()
net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
fr = torch.from_numpy(synthetic_frames(3, 480, 854)).cuda(); mk = torch.from_numpy(synthetic_masks(3, 1, 480, 854)).cuda()
core = InferenceCore(net, cfg); core.set_all_labels([1])
core.put_to_permanent_memory(fr[0], mk[0])
core.step(fr[1], None, None)
ops.RECORD = []
core.step(fr[2], None, None)
rec, ops.RECORD = ops.RECORD, None
groups = {}
for kind, key, flop, fn, keep in rec:
    g = groups.setdefault((kind, key), [0, flop, fn, keep]); g[0] += 1
rows = []
for (kind, key), (count, flop, fn, keep) in groups.items():
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    plan = ops._load_plans().get(key) or ops._tuned_now.get(key)
    rows.append((us * count, count, us, flop / us / 1e6, kind, key, plan))
rows.sort(reverse=True)
print(f'total {sum(r[0] for r in rows)/1e3:.3f} ms over {sum(r[1] for r in rows)} launches')
for t, cnt, us, tf, kind, key, plan in rows:
    print(f'{t:8.1f}us x{cnt:2d} {us:8.1f}us {tf:6.1f} TF/s(alg) {kind:8s} {key:55s} plan={plan}')
