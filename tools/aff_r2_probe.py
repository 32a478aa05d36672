"""Round-2 affinity probe on the real B32 memory: per-kernel times (isolated), hint quality, overflow counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench
cfg = bench.b32_config()
net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
fr = torch.from_numpy(synthetic_frames(64, 480, 854)).cuda(); mk = torch.from_numpy(synthetic_masks(64, 1, 480, 854)).cuda()
core = InferenceCore(net, cfg); core.set_all_labels([1])
for j in range(32):
    core.put_to_permanent_memory(fr[j], mk[j])
perm = core.memory.permanent_work_mem
segs = [(None, None), (perm.key_rows(), perm.shrinkage_rows())]
sizes = [0, perm.size]
qs = []
for f in range(32, 64):
    key, shr, sel = core.encode_frame_key(fr[f])
    qs.append((key[0].permute(1, 2, 0).reshape(-1, 64).contiguous(), sel[0].permute(1, 2, 0).reshape(-1, 64).contiguous()))
HW = qs[0][0].shape[0]
def ws_views():
    ws = ops._workspaces[(str(qs[0][0].device), 'affinity')]
    a = lambda v, al=256: (v + al - 1) // al * al
    cnt_off = a(64 * HW * 88 * 8); bound_off = cnt_off + a(64 * HW * 4); tau_off = bound_off + a(64 * HW * 8 * 4)
    ovf_off = tau_off + a(HW * 4); gcand_off = ovf_off + a(((HW + 63) // 64) * 4); gcnt_off = gcand_off + a(HW * 2048 * 8)
    return (ws[tau_off:tau_off + HW * 4].view(torch.float32), ws[ovf_off:ovf_off + ((HW + 63) // 64) * 4].view(torch.int32),
            ws[gcnt_off:gcnt_off + HW * 4].view(torch.int32))
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
prev = None
NF = int(os.environ.get('PROBE_FRAMES', '32'))
for f, (qk, qe) in enumerate(qs[:NF]):
    for mode, hint in (('nohint', None), ('hint', (prev, sizes, 54) if prev is not None else None), ('hint-nonb', (prev, sizes, 0) if prev is not None else None)):
        if mode != 'nohint' and hint is None:
            continue
        w, idx, sim = ops.affinity_topk(segs, qk, qe, 30, want_sim=True, hint=hint)
        torch.cuda.synchronize()
        tau, ovf, gcnt = ws_views()
        kth = sim[:, -1]
        t = timeit(lambda: ops.affinity_topk(segs, qk, qe, 30, hint=hint), 10)
        if mode == 'hint' and f < 4:
            mkr, msr = perm.key_rows(), perm.shrinkage_rows()
            An = (mkr ** 4).sum(1).sqrt(); Bn = (mkr ** 2).sum(1).sqrt()
            C = (qe ** 2).sum(1).sqrt(); D = ((2 * qk * qe) ** 2).sum(1).sqrt()
            print(f'   rows: An mean {float(An.mean()):.2f} max {float(An.max()):.2f}  Bn mean {float(Bn.mean()):.2f} max {float(Bn.max()):.2f}  ms mean {float(msr.mean()):.2f} max {float(msr.max()):.2f}')
            print(f'   queries: C mean {float(C.mean()):.2f} max {float(C.max()):.2f}  D mean {float(D.mean()):.2f} max {float(D.max()):.2f}')
            eps = ((An * C[:128].max() + Bn * D[:128].max()) * 1.07e-3) * msr / 8
            print(f'   eps (query tile 0): mean {float(eps.mean()):.4f} max {float(eps.max()):.4f};  gcnt hist: >2048: {int((gcnt > 2048).sum())}  >256: {int((gcnt > 256).sum())} >128: {int((gcnt > 128).sum())} median {float(gcnt.float().median()):.0f}')
        if f < 6 or f % 8 == 0 or int(ovf.sum()) > 0:
            print(f'frame {f:2d} {mode:10s}: {t:6.1f} us; tau0 gap to true k-th: mean {float((kth - tau).mean()):.3f} max {float((kth - tau).max()):.3f} '
                  f'(k-th mean {float(kth.mean()):.2f}, top1-kth {float((sim[:, 0] - kth).mean()):.2f}); candidates/query mean {float(gcnt.float().mean()):.1f} '
                  f'max {int(gcnt.max())}; overflowed tiles {int(ovf.sum())}/{ovf.numel()}')
    if mode == 'nohint' or True:
        ref = ops.affinity_topk(segs, qk, qe, 30, want_sim=True)
        chk = ops.affinity_topk(segs, qk, qe, 30, want_sim=True, hint=(prev, sizes, 54) if prev is not None else None)
        assert torch.equal(ref[1], chk[1]) and torch.equal(ref[0], chk[0]), f'frame {f}: hinted result differs'
    prev = idx
print('hinted == un-hinted on all frames')
