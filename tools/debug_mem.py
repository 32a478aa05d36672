"""Diagnostic: replay a golden memory script through the HIP path and print per-step deviations."""
import ast, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
torch.set_grad_enabled(False)
from conftest import load_golden
from test_oracle_goldens import _feed, _query
from xmem2_amd.memory_manager import MemoryManager
T = torch.from_numpy
rows = lambda t: t[0].flatten(1).t().contiguous().cuda()
tag = sys.argv[1] if len(sys.argv) > 1 else 'lt_eviction'
g = load_golden('mem_' + tag)
script = ast.literal_eval(str(g['script'])); cfg = ast.literal_eval(str(g['config']))
h, w = (int(x) for x in g['hw'])
mm = MemoryManager(cfg)
for step, op in enumerate(script):
    if op[0] in ('perm', 'temp'):
        objects, ti = op[1], (op[2] if len(op) > 2 else None)
        key, shr, val, sel = _feed(step, len(objects), (h, w))
        mm.add_memory(rows(key), shr.view(-1).cuda(), val[0].flatten(2).transpose(1, 2).contiguous().cuda(), list(objects),
                      selection=rows(sel), permanent=(op[0] == 'perm'), ti=ti, hw_shape=(h, w))
    elif op[0] == 'replace':
        key, shr, val, sel = _feed(step, op[2], (h, w))
        mm.update_permanent_memory(op[1], rows(key), shr.view(-1).cuda(), val[0].flatten(2).transpose(1, 2).contiguous().cuda(), selection=rows(sel))
    else:
        qk, qe = _query(step, (h, w))
        out = mm.match_memory(qk.cuda(), qe.cuda()).cpu()
        ref = T(g[f'readout_{step}'])
        err = (out - ref).abs()
        perq = err.amax(dim=(0, 1)).flatten()
        msg = f'step {step:2d} match: max err {float(err.max()):.2e} scale {float(ref.abs().max()):.2e} queries>1e-3: {int((perq > 1e-3).sum())}/{perq.numel()}'
        if f'tmp_use_{step}' in g.files and mm.temporary_work_mem.size > 0:
            u = mm.temporary_work_mem.use_count.cpu().flatten().numpy(); ru = g[f'tmp_use_{step}'].flatten()
            if u.shape == ru.shape:
                msg += f' | tmp use max diff {np.abs(u - ru).max():.2e} (max {ru.max():.2e})'
            else:
                msg += f' | tmp use shape {u.shape} vs {ru.shape}'
        if f'lt_use_{step}' in g.files and mm.long_mem.size > 0:
            u = mm.long_mem.use_count.cpu().flatten().numpy(); ru = g[f'lt_use_{step}'].flatten()
            msg += f' | lt use max diff {np.abs(u - ru).max():.2e}' if u.shape == ru.shape else f' | lt use shape {u.shape} vs {ru.shape}'
        print(msg)
    sizes = (mm.temporary_work_mem.size, mm.permanent_work_mem.size, mm.long_mem.size)
    if sizes != tuple(g[f'sizes_{step}']):
        print(f'step {step}: SIZES {sizes} vs {tuple(g[f"sizes_{step}"])}')
if 'lt_key' in g.files:
    lk = mm.long_mem.key.cpu(); rk = T(g['lt_key'])
    if lk.shape == rk.shape:
        print('lt keys identical fraction', float((lk == rk).all(1).float().mean()))
        # as sets
        a = set(map(tuple, np.round(lk[0].t().numpy(), 5).tolist())); b = set(map(tuple, np.round(rk[0].t().numpy(), 5).tolist()))
        print('lt key rows in common', len(a & b), 'of', len(b))
    else:
        print('lt key shape', lk.shape, rk.shape)
