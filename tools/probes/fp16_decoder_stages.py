"""fp16 loop vs fp32: the decoder op by op on identical inputs (readout, features and hidden state of the fp32 stream), K objects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_grad_enabled(False)
import bench
from xmem2_amd import XMem, ops
from xmem2_amd.synth import synthetic_state_dict

dev = torch.device('cuda:0')
sd = synthetic_state_dict(0)
nets = {}
for prec in ('fp32', 'fp16'):
    nets[prec] = XMem(dict(bench.b32_config(), precision=prec), None).to(dev).eval(); nets[prec].load_weights(sd); nets[prec].use_graphs = False


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max()) / max(float(a.abs().max()), 1e-9)


for K in (1, 2, 3):
    torch.manual_seed(K)
    h, w = 30, 54
    img = torch.randn(1, 480, 864, 4, device=dev); img[..., 3] = 0
    feats = {}
    for prec, net in nets.items():
        with ops.precision(prec):
            feats[prec] = net._encode_key_eager(img, True, True)
    f16, f8, f4 = feats['fp32'][3:6]
    print(f'K={K}: key-encoder features fp16 vs fp32: f16 {rel(f16, feats["fp16"][3]):.2e} f8 {rel(f8, feats["fp16"][4]):.2e} f4 {rel(f4, feats["fp16"][5]):.2e}')
    ro = torch.randn(K, h, w, 512, device=dev) * 0.5
    hid = torch.randn(K, h, w, 64, device=dev) * 0.3
    rec = {}
    for prec, net in nets.items():
        W = net._w
        with ops.precision(prec):
            act = ops.act_dtype()
            F16, F8, F4 = [t.to(act) for t in (f16, f8, f4)]            # the SAME features, rounded once for the half path
            cat16 = torch.zeros(K, h, w, 1600, dtype=act, device=dev)
            ops.copy_channels(ro, cat16, 1024)
            r = rec.setdefault(prec, {})
            if not net._shares_x('decoder.fuser', K):
                ops.copy_channels(F16, cat16, 0)
            ops.copy_channels(hid, cat16, 1536)
            r['cat16_g'] = cat16[..., 1024:].clone()
            g16 = net._fusion(cat16, 'decoder.fuser', x=F16); r['g16'] = g16
            skip8 = ops.conv2d(F8, W['decoder.up_16_8.skip_conv']); r['skip8'] = skip8
            u8 = ops.upsample2x_add(g16, skip8); r['up8'] = u8
            g8 = net._group_res(u8, 'decoder.up_16_8.out_conv'); r['g8'] = g8
            skip4 = ops.conv2d(F4, W['decoder.up_8_4.skip_conv']); r['skip4'] = skip4
            g4 = net._group_res(ops.upsample2x_add(g8, skip4), 'decoder.up_8_4.out_conv'); r['g4'] = g4
            logits = ops.conv2d(g4, W['decoder.pred'], relu_in=True, out_dtype=torch.float32); r['logits'] = logits
    for name in rec['fp32']:
        a, b = rec['fp32'][name], rec['fp16'][name]
        per_obj = [f"{rel(a[o], b[o]):.2e}" for o in range(a.shape[0])]
        print(f'   {name:8s} max err / scale per object: {per_obj}')
    sys.stdout.flush()
