"""Early readout under bench.py's hint pattern (the next batch is hinted right after the first frame of a batch is consumed, so readouts
are enqueued ahead ACROSS batch boundaries): masks of a stream with early readout on vs off, frame by frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_grad_enabled(False)
import bench
from xmem2_amd import InferenceCore, XMem, ops
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks

wl = dict(bench.WORKLOADS['b32']); P = int(os.environ.get('PROBE_PERM', '8')); KB = 4; steps = int(os.environ.get('PROBE_STEPS', '40'))
cfg = bench.workload_config(wl)
net = XMem(dict(cfg), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
fr = torch.from_numpy(synthetic_frames(P + 16, 480, 854)).cuda(); mk = torch.from_numpy(synthetic_masks(P + 16, 1, 480, 854)).cuda()
dev = [fr[P + (i % 16)].clone() for i in range(steps + 2 * KB)]


KEEP = []


def run(early, pattern):
    global net
    core = InferenceCore(net, cfg); core.early_readout = early
    core.set_all_labels([1])
    for j in range(P):
        core.put_to_permanent_memory(fr[j], mk[j])
    out, taken = [], 0
    if pattern == 'bench':
        core.prefetch_keys(dev[0:KB])
    for i in range(steps):
        if pattern == 'test' and i % KB == 0:
            core.prefetch_keys(dev[i:i + KB])
        had = core._early is not None
        p = core.step(dev[i], None, None)
        taken += int(had)
        if pattern == 'bench' and i % KB == 0:
            core.prefetch_keys(dev[i + KB:i + 2 * KB])
        out.append(ops.argmax_u8(p).cpu())
    core.cancel_prefetch()
    KEEP.append(core)                 # stays alive: the next core on this network gets another owner token
    return out, taken


def fresh_net():
    n = XMem(dict(cfg), None).to('cuda').eval(); n.load_weights(synthetic_state_dict(0)); return n


for pattern in ('test', 'bench'):
    # a FRESH network for the early run: its HIP-graph stages are captured while readouts are being enqueued ahead
    net = fresh_net()
    b, n = run(True, pattern)
    net = fresh_net()
    a, _ = run(False, pattern)
    bad = [(i, int((x != y).sum())) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]
    print(f'pattern {pattern}: early readouts consumed {n}/{steps}; frames that differ (index, pixels): {bad[:12]}{" ..." if len(bad) > 12 else ""}')

# bench.py's parity leg: a SECOND core on a network whose first core (early readout on) is still alive
net = fresh_net()
run(True, 'bench')
b, n = run(True, 'bench')
net = fresh_net()
a, _ = run(False, 'bench')
bad = [(i, int((x != y).sum())) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]
print(f'second core on a shared network: early readouts consumed {n}/{steps}; frames that differ: {bad[:12]}')
