"""Candidate statistics of the optimistic select pass on network keys: usage aff_diag.py H W perm_frames"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench
H, W, P = (int(x) for x in sys.argv[1:4])
cfg = bench.b32_config()
net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
base = 8
fr = torch.from_numpy(synthetic_frames(base + 4, H, W)).cuda(); mk = torch.from_numpy(synthetic_masks(base + 4, 1, H, W)).cuda()
core = InferenceCore(net, cfg); core.set_all_labels([1])
for j in range(P):
    sh = (3 * (j // base), 5 * (j // base))
    core.put_to_permanent_memory(torch.roll(fr[j % base], sh, (1, 2)), torch.roll(mk[j % base], sh, (1, 2)), ti=j)
perm = core.memory.permanent_work_mem
segs = [(perm.key_rows(), perm.shrinkage_rows())]
for f in (base, base + 1):
    key, shr, sel = core.encode_frame_key(fr[f])
    qk = key[0].permute(1, 2, 0).reshape(-1, 64).contiguous(); qe = sel[0].permute(1, 2, 0).reshape(-1, 64).contiguous()
    HW = qk.shape[0]
    w, idx, sim = ops.affinity_topk(segs, qk, qe, 30, want_sim=True); torch.cuda.synchronize()
    ws = ops._workspaces[(str(qk.device), 'affinity')]
    al = lambda v: (v + 255) // 256 * 256
    cnt_off = al(64 * HW * 88 * 8)
    bound_off = cnt_off + al(64 * HW * 4)
    tau_off = bound_off + al(64 * HW * 8 * 4)
    ovf_off = tau_off + al(HW * 4)
    qt = (HW + 63) // 64
    cnt = ws[cnt_off:cnt_off + 64 * HW * 4].view(torch.int32).view(64, HW).cpu()
    valid = ((cnt >= 0) & (cnt <= 88)).all(1)
    nv = int(valid.long().cumprod(0).sum())          # leading rows that look like counts
    cnt = cnt[:nv]
    tau = ws[tau_off:tau_off + HW * 4].view(torch.float32).cpu()
    ovf = ws[ovf_off:ovf_off + qt * 4].view(torch.int32).cpu()
    used = int((cnt.sum(1) > 0).sum())
    tot = cnt.sum(0).float()
    kth = sim[:, -1].cpu()
    print(f'frame {f}: N={perm.size} HW={HW} splits with data {used}; candidates/query mean {tot.mean():.0f} median {tot.median():.0f} max {tot.max():.0f}; '
          f'per (split,query) max {int(cnt.max())}; flagged query tiles {int((ovf != 0).sum())}/{qt}; '
          f'k-th value - tau0: mean {(kth - tau).mean():.4f} max {(kth - tau).max():.4f}')
