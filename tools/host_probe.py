"""Where does a B32 frame go: host time to enqueue a step vs GPU time (events) vs synced wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench
cfg = bench.b32_config()
net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
fr = torch.from_numpy(synthetic_frames(40, 480, 854)).cuda(); mk = torch.from_numpy(synthetic_masks(40, 1, 480, 854)).cuda()
core = InferenceCore(net, cfg); core.set_all_labels([1])
for j in range(32):
    core.put_to_permanent_memory(fr[j], mk[j])
for i in range(5):
    ops.argmax_u8(core.step(fr[32 + i % 8], None, None)).cpu()
N = 50
# (a) synced per frame (bench style)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    ops.argmax_u8(core.step(fr[32 + i % 8], None, None)).cpu()
torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / N
# (b) enqueue only, sync at the end
torch.cuda.synchronize(); t0 = time.perf_counter()
hs = []
for i in range(N):
    h0 = time.perf_counter()
    m = ops.argmax_u8(core.step(fr[32 + i % 8], None, None))
    hs.append(time.perf_counter() - h0)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / N
# (c) GPU time of one frame between events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for i in range(N):
    m = ops.argmax_u8(core.step(fr[32 + i % 8], None, None))
e1.record(); e1.synchronize()
print(f'synced per frame {ta*1e3:.3f} ms | async per frame {tb*1e3:.3f} ms (host enqueue {t_enq/N*1e3:.3f} ms, median {sorted(hs)[N//2]*1e3:.3f}) | GPU events per frame {e0.elapsed_time(e1)/N:.3f} ms')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    m = ops.argmax_u8(core.step(fr[32 + i % 8], None, None))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
