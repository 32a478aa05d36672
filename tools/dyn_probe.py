"""B32-dyn: default schedule (mem_every=10, T_max=10, T_min=5, long-term on) with 22 permanent frames; per-frame time incl.
encode_value on memory frames and consolidation every 50 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.run_on_video import AsyncMaskFetcher
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench
cfg = bench.b32_config(); cfg.update(mem_every=10, enable_long_term_count_usage=True)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
fr = torch.from_numpy(synthetic_frames(54, 480, 854)).cuda(); mk = torch.from_numpy(synthetic_masks(54, K, 480, 854)).cuda()
core = InferenceCore(net, cfg); core.set_all_labels(list(range(1, K + 1)))
for j in range(22):
    core.put_to_permanent_memory(fr[j], mk[j])
f = AsyncMaskFetcher()
N = 230
times = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    a = time.perf_counter()
    p = core.step(fr[22 + i % 32], None, None)
    core.prefetch_key(fr[22 + (i + 1) % 32])
    f.submit(i, ops.argmax_u8(p))
    times.append(time.perf_counter() - a)
f.drain(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
m = core.memory
print(f'K={K}: {N / dt:.1f} fps over {N} frames; sizes temp {m.temporary_work_mem.size} perm {m.permanent_work_mem.size} long {m.long_mem.size}')
ts = sorted(times)
print('host per-step ms: median %.2f p90 %.2f max %.2f' % (ts[N // 2] * 1e3, ts[int(N * 0.9)] * 1e3, ts[-1] * 1e3))
slow = [(i, round(t * 1e3, 1)) for i, t in enumerate(times) if t > 5e-3]
print('steps slower than 5 ms on the host:', slow[:20])
