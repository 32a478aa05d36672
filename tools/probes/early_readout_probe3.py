"""Does a bench.run_gpu with early readout leave the PROCESS in a state that makes later streams wrong?  Ground truth = a stream on a fresh
network (no early readout) BEFORE run_gpu; the same stream again after it, with and without early readout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_grad_enabled(False)
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict

args = bench.parse_args(['--no-kernel-trace', '--no-extra-modes', '--plain-steps', '0', '--steps', '40', '--scale-only'])
args.no_cpu_baseline = True
device = torch.device('cuda', 0)
wl = bench.WORKLOADS['b32']; cfg = bench.workload_config(wl)
frames, masks, base, nq = bench.make_clip(wl)
sd = synthetic_state_dict(0)
fr = torch.from_numpy(frames).to(device); mk = torch.from_numpy(masks).to(device)
KB, n_total = 4, 12
dev = [fr[base + i].clone() for i in range(n_total + 2 * KB)]


def stream(early, tag):
    net = XMem(dict(cfg), None).to(device).eval(); net.load_weights(sd)
    gpu = InferenceCore(net, cfg); gpu.early_readout = early
    gpu.set_all_labels([1])
    for j in range(wl['perm']):
        gpu.put_to_permanent_memory(fr[j], mk[j])
    gpu.prefetch_keys(dev[0:KB])
    out = []
    for i in range(n_total):
        pg = gpu.step(dev[i], None, None)
        if i % KB == 0:
            gpu.prefetch_keys(dev[i + KB:i + 2 * KB])
        out.append(ops.argmax_u8(pg).cpu())
    gpu.cancel_prefetch()
    return out


def diff(a, b):
    return [(i, int((x != y).sum())) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]


truth = stream(False, 'before')
print('early stream on a fresh network BEFORE any bench run vs truth:', diff(stream(True, 'early-before'), truth))
res = bench.run_gpu(args, device, 0, 1)
print('no-early stream AFTER bench.run_gpu vs truth:', diff(stream(False, 'after'), truth))
print('early stream AFTER bench.run_gpu vs truth:', diff(stream(True, 'early-after'), truth))
