"""Per-launch table of ONE batched key-encoder pass (batch B, with the decoder's skip convolutions inline) at 480p:
shape key, plan, us per launch, algorithmic TFLOP/s, bytes moved (in + out + residual) and the GB/s that implies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xmem2_amd import XMem, ops
from xmem2_amd.synth import synthetic_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
cfg = bench.b32_config()
net = XMem(dict(cfg), None).to(dev).eval(); net.load_weights(synthetic_state_dict(0))
img = torch.randn(B, 480, 864, 4, device=dev); img[..., 3] = 0
net._encode_key_eager(img, True, True, False, True)
ops.RECORD = []
net._encode_key_eager(img, True, True, False, True)
records, ops.RECORD = ops.RECORD, None
rows = {}
for kind, key, flop, fn, keep in records:
    r = rows.setdefault((kind, key), [0, flop, fn, keep])
    r[0] += 1
out = []
for (kind, key), (count, flop, fn, keep) in rows.items():
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    x, o, res = keep[0], keep[1], keep[2]
    nbytes = 4 * (x.numel() + o.numel() + (res.numel() if res is not None else 0))
    out.append((us * count, us, count, kind, key, flop, nbytes))
out.sort(reverse=True)
tot = sum(o[0] for o in out)
print(f'batch {B}: total {tot:.0f} us over {sum(o[2] for o in out)} conv launches = {tot / B:.0f} us per frame')
plans = ops._load_plans()
for t, us, c, kind, key, flop, nb in out:
    print(f'{t:8.1f} us  {c:2d} x {us:7.1f} us  {flop / us / 1e6:6.1f} TF  {nb / us / 1e3:7.0f} GB/s(act)  {kind:5s} {key}  plan={plans.get(key)}')
