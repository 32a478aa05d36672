"""Round 6: bench.py's own parity leg (run_cpu_baseline: identical GPU-side call and allocation sequence) with the oracle replaced by a stub
(PROBE_STUB=1: no CPU arithmetic, PROBE_DELAY seconds of sleep per preload frame) or the real oracle (PROBE_STUB=0).
XMEM_BENCH_PARITY_TRACE=1 makes the leg repeat the GPU stream on fresh cores and print the per-frame differences."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_grad_enabled(False)
os.environ['XMEM_BENCH_PARITY_TRACE'] = '1'
import bench
from oracle import cpu_ref as R

delay = float(os.environ.get('PROBE_DELAY', '0.3'))
if os.environ.get('PROBE_STUB', '1') == '1':
    class StubNet:
        def __init__(self, sd): pass

    class StubCore:
        def __init__(self, net, cfg): pass
        def set_all_labels(self, l): self.k = len(l)
        def put_to_permanent_memory(self, im, mk):
            if os.environ.get('PROBE_BURN'):          # CPU arithmetic on all torch threads instead of sleeping
                a = torch.randn(2048, 2048)
                t0 = time.time()
                while time.time() - t0 < delay:
                    a = (a @ a).clamp(-1, 1)
            else:
                time.sleep(delay)
        def step(self, im, a, b):
            p = torch.zeros((self.k + 1,) + tuple(im.shape[-2:]))
            p[0] = 1.0
            return p
    R.RefNet, R.RefCore = StubNet, StubCore

args = bench.parse_args(['--no-kernel-trace', '--no-extra-modes', '--steps', '20', '--warmup', '5', '--cpu-frames', '4'])
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
res = bench.run_gpu(args, device, 0, 1)
if os.environ.get('PROBE_SYNC_EMPTY'):
    torch.cuda.synchronize(); torch.cuda.empty_cache()
cpu, parity = bench.run_cpu_baseline(res, args, device)
print('PROBE done', parity['argmax_mismatch_pixels'])
