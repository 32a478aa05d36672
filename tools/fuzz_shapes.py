"""Random frame geometries / object counts / schedules through InferenceCore.step against the oracle (CPU restatement).
usage: fuzz_shapes.py [n_cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_grad_enabled(False)
from oracle import cpu_ref as R
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
sd = synthetic_state_dict(0)
cfg0 = bench.b32_config()
net = XMem(dict(cfg0), None).to('cuda').eval(); net.load_weights(sd)
ref_net = R.RefNet(sd)
worst = 0.0
for case in range(n_cases):
    H, W = rng.randint(48, 260), rng.randint(48, 330)
    K = rng.choice([1, 1, 2, 3])
    T = rng.randint(4, 7)
    cfg = dict(cfg0); cfg.update(mem_every=rng.choice([1, 2, 3]), enable_long_term_count_usage=True,
                                 max_mid_term_frames=3, min_mid_term_frames=1, num_prototypes=16, top_k=rng.choice([10, 30]))
    frames = torch.from_numpy(synthetic_frames(T, H, W, seed=case)); masks = torch.from_numpy(synthetic_masks(T, K, H, W))
    core, ref = InferenceCore(net, cfg), R.RefCore(ref_net, cfg)
    labels = list(range(1, K + 1))
    for c in (core, ref):
        c.set_all_labels(labels)
    core.put_to_permanent_memory(frames[0].cuda(), masks[0].cuda()); ref.put_to_permanent_memory(frames[0], masks[0])
    use_hint = rng.random() < 0.5
    dev = [frames[t].cuda() for t in range(1, T)]
    if use_hint:
        core.prefetch_keys(dev[:2])
    mism = n = 0; perr = 0.0
    for t in range(1, T):
        try:
            p = core.step(dev[t - 1], None, None, end=(t == T - 1))
            q = ref.step(frames[t], None, None, end=(t == T - 1))
        except RuntimeError as e:                      # e.g. fewer memory elements than top_k: both sides must raise
            try:
                ref.step(frames[t], None, None, end=(t == T - 1)); raise AssertionError(f'only the HIP path raised: {e}')
            except RuntimeError:
                break
        a, b = ops.argmax_u8(p).cpu().numpy(), torch.argmax(q, 0).numpy().astype(np.uint8)
        mism += int((a != b).sum()); n += a.size
        perr = max(perr, float((p.cpu() - q).abs().mean()))
        m, rm = core.memory, ref.memory
        assert (m.temporary_work_mem.size, m.permanent_work_mem.size, m.long_mem.size) == \
               (rm.temporary_work_mem.size, rm.permanent_work_mem.size, rm.long_mem.size), (case, t)
    frac = mism / max(n, 1)
    worst = max(worst, frac)
    print(f'case {case}: {H}x{W} K={K} T={T} mem_every={cfg["mem_every"]} top_k={cfg["top_k"]} hint={use_hint}: '
          f'argmax mismatch {mism}/{n} ({frac:.1e}), mean|dp| {perr:.1e}', flush=True)
    assert frac < 2e-3 and perr < 1e-3, 'parity lost'
print('fuzz ok, worst mismatch fraction', worst)
