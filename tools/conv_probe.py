"""Per-layer timing of every convolution of one B32 frame (encode_key + segment [+ encode_value]) on the GPU."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops
from xmem2_amd.ops import ConvWeights

calls = []
_orig = ops.conv2d


def spy(x, cw, out=None, out_ld=None, res=None, relu_in=False, relu_out=False, in_ld=None, cin=None):
    calls.append((tuple(x.shape[:3]), in_ld or x.shape[3], cw, res is not None, relu_in, relu_out))
    return _orig(x, cw, out=out, out_ld=out_ld, res=res, relu_in=relu_in, relu_out=relu_out, in_ld=in_ld, cin=cin)


def main():
    from xmem2_amd import XMem, InferenceCore
    from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
    import bench
    cfg = bench.b32_config()
    net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
    fr = torch.from_numpy(synthetic_frames(3, 480, 854)).cuda(); mk = torch.from_numpy(synthetic_masks(3, 1, 480, 854)).cuda()
    core = InferenceCore(net, cfg); core.set_all_labels([1])
    core.put_to_permanent_memory(fr[0], mk[0])
    ops.conv2d = spy
    import xmem2_amd.network as nw
    nw.ops.conv2d = spy
    core.step(fr[1], None, None)
    ops.conv2d = _orig; nw.ops.conv2d = _orig
    seen = {}
    for (bhw, ldin, cw, has_res, ri, ro) in calls:
        key = (bhw, cw.cin, cw.cout, cw.kh, cw.stride, cw.pad)
        seen.setdefault(key, [0, cw, has_res, ri, ro, ldin])[0] += 1
    rows = []
    for key, (cnt, cw, has_res, ri, ro, ldin) in seen.items():
        (B, H, W), cin, cout, k, stride, pad = key
        x = torch.randn(B, H, W, ldin, device='cuda')
        Ho = (H + 2 * pad - k) // stride + 1; Wo = (W + 2 * pad - k) // stride + 1
        res = torch.randn(B, Ho, Wo, cout, device='cuda') if has_res else None
        for _ in range(3):
            ops.conv2d(x, cw, res=res, relu_in=ri, relu_out=ro)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.conv2d(x, cw, res=res, relu_in=ri, relu_out=ro)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        flop = 2.0 * B * Ho * Wo * cout * k * k * cw.cin_true
        rows.append((us * cnt, cnt, us, flop / us / 1e6, key))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f'total conv time per frame {tot/1e3:.3f} ms over {sum(r[1] for r in rows)} launches')
    for t, cnt, us, tf, key in rows:
        (B, H, W), cin, cout, k, stride, pad = key
        M = B * ((H + 2 * pad - k) // stride + 1) * ((W + 2 * pad - k) // stride + 1)
        print(f'{t:8.1f}us x{cnt:2d} {us:8.1f}us {tf:6.1f} TF/s  M={M:6d} Cin={cin:5d} Cout={cout:5d} k={k} s={stride}  K={k*k*cin}')


if __name__ == '__main__':
    main()
