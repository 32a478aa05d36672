"""Affinity on the real B32 memory (keys from the network on the synthetic clip): candidate counts + per-kernel time."""
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
key, shr, sel = core.encode_frame_key(fr[33])
qk = key[0].permute(1, 2, 0).reshape(-1, 64).contiguous(); qe = sel[0].permute(1, 2, 0).reshape(-1, 64).contiguous()
perm = core.memory.permanent_work_mem
segs = [(perm.key_rows(), perm.shrinkage_rows())]
HW = qk.shape[0]
w, idx, sim = ops.affinity_topk(segs, qk, qe, 30, want_sim=True)
torch.cuda.synchronize()
ws = ops._workspaces[(str(qk.device), 'affinity')]
cnt_off = (64 * HW * 88 * 8 + 255) // 256 * 256
cnt = ws[cnt_off:cnt_off + 64 * HW * 4].view(torch.int32).view(64, HW)[:19].cpu()
tot = cnt.sum(0).float()
print(f'candidates per query: mean {tot.mean():.1f} max {tot.max():.0f} min {tot.min():.0f}; per (split,query) max {int(cnt.max())}')
print('sim top1 mean', float(sim[:, 0].mean()), 'k-th mean', float(sim[:, -1].mean()), 'spread', float((sim[:, 0] - sim[:, -1]).mean()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.affinity_topk(segs, qk, qe, 30)
e1.record(); e1.synchronize()
print(f'affinity on real keys: {e0.elapsed_time(e1) * 50:.1f} us per call')

# every query frame of the bench clip: worst (split, query) candidate count and time per call
perm = core.memory.permanent_work_mem
segs = [(perm.key_rows(), perm.shrinkage_rows())]
for f in range(32, 64):
    key, shr, sel = core.encode_frame_key(fr[f])
    qk = key[0].permute(1, 2, 0).reshape(-1, 64).contiguous(); qe = sel[0].permute(1, 2, 0).reshape(-1, 64).contiguous()
    ops.affinity_topk(segs, qk, qe, 30); torch.cuda.synchronize()
    ws = ops._workspaces[(str(qk.device), 'affinity')]
    cnt = ws[cnt_off:cnt_off + 64 * HW * 4].view(torch.int32).view(64, HW)[:19].cpu()
    e0.record()
    for _ in range(5):
        ops.affinity_topk(segs, qk, qe, 30)
    e1.record(); e1.synchronize()
    print(f'frame {f}: per (split,query) max {int(cnt.max())}, queries with a split > 40: {int((cnt.max(0).values > 40).sum())}, total/query mean {cnt.sum(0).float().mean():.1f} max {int(cnt.sum(0).max())}; {e0.elapsed_time(e1) * 200:.0f} us')
