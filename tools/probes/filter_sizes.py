"""Pass-1 filter kernel and whole hinted call at the three served memory sizes (synthetic keys, perfect hint):
   python tools/probes/filter_sizes.py [b32 c4 c5]      (XMEM_F16_SPLITS=n overrides the split count for an A/B)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from xmem2_amd import ops
from xmem2_amd._lib import load
from test_gpu_affinity_served_sizes import _make, SIZES
names = dict(zip(['b32', 'c4', 'c5'], SIZES))
lib = load()
for nm in (sys.argv[1:] or ['b32', 'c4', 'c5']):
    n, hw, gw, nseg = names[nm]
    mk, ms, qk, qe, cuts = _make(n, hw, 1, seed=n)
    r16 = ops.affinity_rows16(mk, ms, torch.empty(n, ops.ROWS16_FLOATS, device='cuda'))
    segs = [(mk, ms, r16)]
    w0, i0, s0 = ops.affinity_topk(segs, qk, qe, 30, want_sim=True)
    hint = (i0, [n], gw)
    noise = float(os.environ.get('PROBE_NOISE', '0'))      # > 0: the hint comes from PERTURBED queries (a looser bound, longer lists - as on a moving video)
    if noise > 0:
        _, ih, _ = ops.affinity_topk(segs, qk + noise * torch.randn_like(qk), qe, 30, want_sim=True)
        hint = (ih, [n], gw)
        if os.environ.get('PROBE_HEAVY'):                   # a block of queries with a garbage hint (their lists run to the capacity)
            a, b = (int(x) for x in os.environ['PROBE_HEAVY'].split(':'))
            ih[a:b] = torch.randint(0, n, (b - a, ih.shape[1]), device=ih.device, dtype=ih.dtype)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    tf, tc = [], []
    for it in range(12):
        lib.xmem_affinity_profile_events(C.c_void_p(ev[0].cuda_event), C.c_void_p(ev[1].cuda_event))
        ev[2].record()
        w, i, s = ops.affinity_topk(segs, qk, qe, 30, want_sim=True, hint=hint)
        ev[3].record()
        torch.cuda.synchronize()
        lib.xmem_affinity_profile_events(None, None)
        if it >= 2:
            tf.append(ev[0].elapsed_time(ev[1]) * 1e3); tc.append(ev[2].elapsed_time(ev[3]) * 1e3)
    assert os.environ.get("PROBE_NOCHECK", "0") != "0" or (torch.equal(i, i0) and torch.equal(s, s0))
    flop = 2.0 * 144 * ((n + 31) // 32 * 32) * ((hw + 127) // 128 * 128)
    tf.sort(); tc.sort()
    med = tf[len(tf) // 2]
    if noise > 0:
        o = [C.c_size_t(), C.c_size_t(), C.c_size_t()]
        lib.xmem_affinity_debug_offsets(n, hw, *[C.byref(x) for x in o])
        ws = ops.workspace(0, torch.device('cuda', torch.cuda.current_device()), 'affinity')
        cnt = ws[o[0].value:o[0].value + 4 * hw].view(torch.int32).float()
        print(f'   lists: mean {float(cnt.mean()):.0f} median {float(cnt.median()):.0f} max {int(cnt.max())}', flush=True)
    print(f'{nm}: N={n} HW={hw}: filter pass 1 median {med:8.1f} us (min {tf[0]:.1f}) = {flop / med / 1e6:6.0f} TFLOP/s executed '
          f'({flop / med / 1e6 / 2500:.2f} of the fp16 MFMA peak); whole hinted call median {tc[len(tc) // 2]:8.1f} us', flush=True)
    del mk, ms, r16, segs
    torch.cuda.empty_cache()
