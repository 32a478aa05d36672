"""Long dynamic run: default-like schedule with small long-term capacity so that consolidation AND eviction fire many times;
checks finiteness / normalisation of every output, bounded memory sizes and flat HBM use.  usage: soak_probe.py [frames] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = bench.b32_config(); cfg.update(mem_every=3, enable_long_term_count_usage=True, max_mid_term_frames=6, min_mid_term_frames=3,
                                     num_prototypes=64, max_long_term_elements=int(os.environ.get('SOAK_LT_CAP', '600')))
net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
H, W = int(os.environ.get('SOAK_H', '240')), int(os.environ.get('SOAK_W', '427'))
PERM = int(os.environ.get('SOAK_PERM', '1'))      # permanent frames: >= 6 at 480p puts the readout on the large-memory (wide select) path
fr = torch.from_numpy(synthetic_frames(40, H, W)).cuda(); mk = torch.from_numpy(synthetic_masks(40, K, H, W)).cuda()
core = InferenceCore(net, cfg); core.set_all_labels(list(range(1, K + 1)))
for j in range(PERM):
    core.put_to_permanent_memory(fr[j], mk[j])
KB, bad, peak_lt, evictions, last_lt = 4, 0, 0, 0, 0
NOPF = os.environ.get('SOAK_NOPF') == '1'
frame = lambda i: fr[1 + i % 39]
if not NOPF: core.prefetch_keys([frame(j) for j in range(KB)])
mem0 = None
t0 = time.perf_counter()
for i in range(N):
    p = core.step(frame(i), None, None)
    if i % KB == 0 and not NOPF:
        core.prefetch_keys([frame(i + KB + j) for j in range(KB)])
    if i % 50 == 0:
        fin, dev = bool(torch.isfinite(p).all()), float((p.sum(0) - 1).abs().max())
        ok = fin and dev < 1e-4
        if not ok: print(f'BAD frame {i}: finite {fin} max|sum-1| {dev:.3e} min {float(p.min()):.3e} max {float(p.max()):.3e}', flush=True)
        bad += 0 if ok else 1
        lt = core.memory.long_mem.size
        peak_lt = max(peak_lt, lt); evictions += 1 if lt < last_lt else 0; last_lt = lt
        if i == 200: mem0 = torch.cuda.memory_allocated()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
m = core.memory
print(f'{N} frames K={K}: {N / dt:.1f} fps; bad outputs {bad}; long-term size now {m.long_mem.size} (peak seen {peak_lt}, '
      f'{evictions} shrink events seen); temp {m.temporary_work_mem.size}; HBM at frame 200 {mem0 / 2**20:.0f} MiB, now {torch.cuda.memory_allocated() / 2**20:.0f} MiB')
assert bad == 0 and m.long_mem.size <= int(os.environ.get('SOAK_LT_CAP', '600')) + 64
