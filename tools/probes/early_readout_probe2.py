"""bench.py's parity leg, GPU side only: after bench.run_gpu (its core still alive) a second core streams frames with the bench's hint pattern;
compared with the same stream on a fresh network without early readout.  PROBE_ARGS = extra bench flags (e.g. --scale-only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_grad_enabled(False)
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict

args = bench.parse_args(['--no-kernel-trace', '--no-extra-modes', '--plain-steps', '0', '--steps', '40'] + os.environ.get('PROBE_ARGS', '').split())
if args.scale_only:
    args.no_cpu_baseline = args.no_kernel_trace = args.no_extra_modes = True; args.plain_steps = 0
device = torch.device('cuda', 0)
res = bench.run_gpu(args, device, 0, 1)
wl, cfg = res['wl'], res['cfg']
fr, mk, KB, n_total = res['frames'], res['masks_in'], 4, 24
frame_fn = res['frame_fn']
dev = [frame_fn(i).clone() for i in range(n_total + 2 * KB)]


def stream(net, early):
    gpu = InferenceCore(net, cfg); gpu.early_readout = early
    gpu.set_all_labels([1])
    for j in range(wl['perm']):
        gpu.put_to_permanent_memory(torch.from_numpy(fr[j]).to(device), torch.from_numpy(mk[j]).to(device))
    gpu.prefetch_keys(dev[0:KB])
    out = []
    for i in range(n_total):
        pg = gpu.step(dev[i], None, None)
        if i % KB == 0:
            gpu.prefetch_keys(dev[i + KB:i + 2 * KB])
        out.append(ops.argmax_u8(pg).cpu())
    gpu.cancel_prefetch()
    return out


host = torch.from_numpy(fr)
frd = frame_fn(0).new_empty(0)                      # (keeps the device copy of the frames reachable)
base = res['base']
dev_frames = torch.stack([frame_fn(i) for i in range(32)])
before = dev_frames.cpu().clone()
assert torch.equal(before, host[base:base + 32]), 'device frames differ from the host frames BEFORE the stream'
b = stream(res['core'].network, True)
after = torch.stack([frame_fn(i) for i in range(32)]).cpu()
chg = [(i, int((after[i] != before[i]).sum())) for i in range(32) if not torch.equal(after[i], before[i])]
print(f'device copies of the input frames changed by the early-readout stream: {chg[:16]}')
chg2 = [(i, int((dev[i].cpu() != host[base + (i % 32)]).sum())) for i in range(len(dev)) if not torch.equal(dev[i].cpu(), host[base + (i % 32)])]
print(f'cloned input frames changed: {chg2[:16]}')
net2 = XMem(dict(cfg), None).to(device).eval(); net2.load_weights(res['sd'])
a = stream(net2, False)
bad = [(i, int((x != y).sum())) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]
print(f'PROBE_ARGS={os.environ.get("PROBE_ARGS", "")!r}: frames that differ (index, pixels): {bad[:16]}')
