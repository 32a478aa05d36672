"""Round 6: fast reproducer of the wrong stream of bench.py's parity leg without the oracle.
bench.run_gpu (early readout on) -> core 2 on the SAME network, preload with host delays between the calls (the oracle's place),
stream n frames -> core 3 the same without delays.  Core 3 was always right in round 5's traces, so core2 != core3 is the defect.
Env: PROBE_DELAY (s per preload frame, default 0.3), PROBE_DUMP=1 (compare intermediates of frame 0), PROBE_EMPTY_CACHE=1,
PROBE_PLAIN (plain steps, default 60), PROBE_SKIP (XMEM_BENCH_SKIP_PASSES)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_grad_enabled(False)
import bench
from xmem2_amd import InferenceCore, ops

delay = float(os.environ.get('PROBE_DELAY', '0.3'))
argv = ['--no-kernel-trace', '--no-extra-modes', '--steps', '20', '--warmup', '5', '--plain-steps', os.environ.get('PROBE_PLAIN', '60')]
args = bench.parse_args(argv)
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
res = bench.run_gpu(args, device, 0, 1)
wl, cfg = res['wl'], res['cfg']
fr, mk, base = res['frames'], res['masks_in'], res['base']
pnet = res['core'].network
frame_fn = res['frame_fn']
KB, n_total = 4, int(os.environ.get('PROBE_FRAMES', '12'))
dev = [frame_fn(i).clone() for i in range(n_total + 2 * KB)]
if os.environ.get('PROBE_EMPTY_CACHE'):
    torch.cuda.synchronize(); torch.cuda.empty_cache()
dumps = {}


def stream(tag, d, early=None):
    g = InferenceCore(pnet, cfg)
    if early is not None:
        g.early_readout = early
    g.set_all_labels([1])
    for j in range(wl['perm']):
        if d:
            time.sleep(d)
        g.put_to_permanent_memory(torch.from_numpy(fr[j]).to(device), torch.from_numpy(mk[j]).to(device))
    if os.environ.get('PROBE_DUMP'):
        torch.cuda.synchronize()
        m = g.memory.permanent_work_mem
        dumps[tag] = dict(keys=m.key_rows().clone(), shr=m.shrinkage_rows().clone(), val=m.value_rows(0).clone(), r16=m.rows16().clone(),
                          hidden=g.memory.get_hidden().clone())
    g.prefetch_keys(dev[0:KB])
    out = []
    for i in range(n_total):
        p = g.step(dev[i], None, None)
        if i % KB == 0:
            g.prefetch_keys(dev[i + KB:i + 2 * KB])
        out.append(ops.argmax_u8(p).cpu())
        if os.environ.get('PROBE_DUMP') and i == 0:
            torch.cuda.synchronize()
            dumps[tag]['prob0'] = p.clone()
            dumps[tag]['hint0'] = g.memory._aff_hint[0][0].clone()
    g.cancel_prefetch()
    return out, g


def diff(a, b):
    return [int((x != y).sum()) for x, y in zip(a, b)]


o2, g2 = stream('c2', delay)
o3, g3 = stream('c3', 0.0)
o4, g4 = stream('c4', delay)
print('PROBE core2(delay) vs core3(no delay):', diff(o2, o3))
print('PROBE core4(delay) vs core3(no delay):', diff(o4, o3))
if os.environ.get('PROBE_DUMP'):
    for k in dumps['c2']:
        a, b = dumps['c2'][k], dumps['c3'][k]
        if a.shape != b.shape:
            print('PROBE dump', k, 'shape differs', a.shape, b.shape); continue
        ne = int((a != b).sum())
        print(f'PROBE dump {k}: differing elements core2 vs core3: {ne} of {a.numel()}' + (f' max |d| {float((a.float() - b.float()).abs().max()):.3e}' if ne else ''))
