"""Per-launch table of one B32 frame: every conv call (shape key), its plan, duration and algorithmic TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict

dev = torch.device('cuda:0')
cfg = bench.b32_config()
net = XMem(dict(cfg), None).to(dev).eval(); net.load_weights(synthetic_state_dict(0))
frames, masks, _base, _nq = bench.make_clip(bench.WORKLOADS['b32'])
fr, mk = torch.from_numpy(frames).to(dev), torch.from_numpy(masks).to(dev)
core = InferenceCore(net, cfg); core.set_all_labels([1])
for j in range(32):
    core.put_to_permanent_memory(fr[j], mk[j])
core.step(fr[32], None, None)
ops.RECORD = []
core.step(fr[33], None, None)
records, ops.RECORD = ops.RECORD, None
rows = {}
for kind, key, flop, fn, keep in records:
    r = rows.setdefault((kind, key), [0, flop, fn])
    r[0] += 1
out = []
for (kind, key), (count, flop, fn) in rows.items():
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    out.append((us * count, us, count, kind, key, flop))
out.sort(reverse=True)
tot = sum(o[0] for o in out)
print(f'total {tot:.0f} us over {sum(o[2] for o in out)} launches')
plans = ops._load_plans() if hasattr(ops, '_load_plans') else {}
for t, us, c, kind, key, flop in out:
    plan = plans.get(key) if isinstance(plans, dict) else None
    print(f'{t:8.1f} us  {c:2d} x {us:7.1f} us  {flop / us / 1e6:6.1f} TF  {kind:8s} {key}  plan={plan}')
