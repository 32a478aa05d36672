"""Every convolution of one B32 frame (the network's own activations and weights): max |error| / max |reference| against a
float64 convolution on the host, for the fp32 kernels and for the split-operand ('fp32x') kernels with the plans each mode
ships.  Writes a table (profiles/r03_split_conv_errors.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cfg = bench.b32_config()
net = XMem(dict(cfg), None).to(dev).eval(); net.load_weights(synthetic_state_dict(0))
net.use_graphs = False
frames, masks, _b, _n = bench.make_clip(bench.WORKLOADS['b32'])
fr, mk = torch.from_numpy(frames).to(dev), torch.from_numpy(masks).to(dev)
core = InferenceCore(net, cfg); core.set_all_labels([1])
for j in range(4):
    core.put_to_permanent_memory(fr[j], mk[j])
core.step(fr[32], None, None)
ops.RECORD = []
core.step(fr[33], None, None)
recs, ops.RECORD = ops.RECORD, None
torch.cuda.synchronize()
seen, rows = set(), []
for kind, key, flop, fn, keep in recs:
    if kind != 'conv' or key in seen:
        continue
    seen.add(key)
    x, out, res, cw, ws, meta = keep
    if cw.cout == 1:
        continue                                         # the mask head is a GEMV on the fp32 VALU in both modes
    cin = meta['cin']
    xs = x[..., :cin] if x.shape[3] != cin else x
    xd = xs.double().cpu().permute(0, 3, 1, 2)
    if meta['relu_in']:
        xd = xd.relu()
    wd = cw.w[..., :cin].double().cpu().permute(0, 3, 1, 2) if cw.w.shape[3] != cin else cw.w.double().cpu().permute(0, 3, 1, 2)
    ref = F.conv2d(xd, wd, None, cw.stride, cw.pad)
    ref = ref * cw.scale.double().cpu().view(1, -1, 1, 1) + cw.shift.double().cpu().view(1, -1, 1, 1)
    if res is not None:
        rr = res[..., :cw.cout].double().cpu().permute(0, 3, 1, 2)
        ref = ref + rr
    if meta['relu_out']:
        ref = ref.relu()
    scale = float(ref.abs().max())
    errs, plans = {}, {}
    for mode in ('fp32', 'fp32x'):
        with ops.precision(mode):
            y = ops.conv2d(x, cw, res=res, relu_in=meta['relu_in'], relu_out=meta['relu_out'],
                           in_ld=meta['in_ld'] if meta['in_ld'] != x.shape[3] else None, cin=cin, res_broadcast=meta['res_broadcast'])
        errs[mode] = float((y.double().cpu().permute(0, 3, 1, 2) - ref).abs().max()) / max(scale, 1e-30)
        plans[mode] = ops._lookup_plan(key, mode == 'fp32x')
    rows.append((key, plans['fp32'], errs['fp32'], plans['fp32x'], errs['fp32x']))
print(f'{"layer (B x H x W x Cin/ld -> Cout/ld, kernel, res/relu flags)":64s} {"fp32 plan":>10s} {"fp32 err":>10s} {"fp32x plan":>11s} {"fp32x err":>10s}')
for key, p0, e0, p1, e1 in rows:
    print(f'{key:64s} {str(p0):>10s} {e0:10.2e} {str(p1):>11s} {e1:10.2e}')
print(f'worst over {len(rows)} layers: fp32 {max(r[2] for r in rows):.2e}   fp32x {max(r[4] for r in rows):.2e}   (max |error| / max |float64 reference| per layer)')
