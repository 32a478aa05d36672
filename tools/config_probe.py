"""Throughput of the other BASELINE configs through InferenceCore.step with batched key hints.
usage: config_probe.py H W K perm_frames mem_every n_frames"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.run_on_video import AsyncMaskFetcher
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench
H, W, K, P, ME, N = (int(x) for x in sys.argv[1:7])
cfg = bench.b32_config(); cfg.update(mem_every=ME, enable_long_term_count_usage=True)
net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
base = 8
fr = torch.from_numpy(synthetic_frames(base + 16, H, W)).cuda(); mk = torch.from_numpy(synthetic_masks(base + 16, K, H, W)).cuda()
core = InferenceCore(net, cfg); core.set_all_labels(list(range(1, K + 1)))
t0 = time.perf_counter()
for j in range(P):                       # distinct memory frames: each base frame shifted by a different offset
    sh = (3 * (j // base), 5 * (j // base))
    core.put_to_permanent_memory(torch.roll(fr[j % base], sh, (1, 2)), torch.roll(mk[j % base], sh, (1, 2)), ti=j)
torch.cuda.synchronize()
print(f'preload {P} frames: {time.perf_counter() - t0:.2f} s; permanent elements {core.memory.permanent_work_mem.size}; '
      f'HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB')
f = AsyncMaskFetcher()
KB = 4
frame = lambda i: fr[base + i % 16]
def run(n, start):
    for i in range(start, start + n):
        p = core.step(frame(i), None, None)
        if i % KB == 0:
            core.prefetch_keys([frame(i + KB + j) for j in range(KB)])
        f.submit(i, ops.argmax_u8(p))
core.prefetch_keys([frame(j) for j in range(KB)])
WU = (2 * KB + 2 * ME + KB - 1) // KB * KB          # warm-up covers two memory frames (value-encoder graphs) and whole batches
run(WU, 0)
f.drain(); torch.cuda.synchronize(); t0 = time.perf_counter()
run(N, WU)
f.drain(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
m = core.memory
print(f'{H}x{W} K={K} perm_frames={P} mem_every={ME}: {N / dt:.1f} fps ({dt / N * 1e3:.2f} ms/frame) over {N} frames; '
      f'sizes temp {m.temporary_work_mem.size} perm {m.permanent_work_mem.size} long {m.long_mem.size}; '
      f'HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak')
