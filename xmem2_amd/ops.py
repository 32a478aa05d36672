"""Tensor-level wrappers over the C-ABI kernels.

Everything here takes / returns torch CUDA tensors in the kernels' native layouts (NHWC activations
``[B,H,W,C]``, row-major memory rows ``[N,C]``) and launches on torch's current HIP stream.
torch is plumbing only: allocation, streams.  No torch compute op is used on the product path.
"""
import contextlib
import ctypes as C
import json
import math
import os
import sys
import threading

import torch

from . import _lib
from ._lib import AffinityHint, ConvDesc, KeySegment, ValueSegment, check, load, ptr, stream_ptr

_workspaces = {}
_retired = {}                  # key -> outgrown buffers that captured HIP graphs may still point into
_tls = threading.local()       # .suffix: scratch scope of the calling thread (see ws_scope)
_scope_free, _scope_next = [], [0]


def new_scope():
    """Small recycled integer naming one owner of scoped scratch (an XMem instance: its side-stream key-encoder stages).
    `release_scope` drops every workspace of the scope when the owner dies, so a process that builds one network per video
    (run_on_video, launch.py) does not keep ~100 MB of conv scratch per dead network."""
    if _scope_free:
        return _scope_free.pop()
    _scope_next[0] += 1
    return _scope_next[0]


@contextlib.contextmanager
def ws_scope(suffix):
    """Kernels launched inside take their scratch from buffers tagged `suffix` (per host thread, nestable): stages that may
    run concurrently with the main stream (the side-stream key encoder) never share scratch with it."""
    prev = getattr(_tls, 'suffix', '')
    _tls.suffix = suffix
    try:
        yield
    finally:
        _tls.suffix = prev


def release_scope(scope):
    mark = f'#{scope}#'
    for d in (_workspaces, _retired):
        for key in [k for k in d if mark in k[1]]:
            del d[key]
    _scope_free.append(scope)

# Optional live kernel timing for bench.py.  RECORD = [] makes conv2d / affinity_topk append a re-launchable
# closure for every call of one (eager) frame; time_recorded() then times each distinct launch back to back between
# two HIP events (torch events record on the current stream, which is the stream every kernel here is launched on).
RECORD = None
PROFILE = None          # kept for compatibility: any non-None value also forces the eager (non-graph) path
# EVENT_TAP = [] makes the memory-readout launches (which stay eager between the captured stages) bracket themselves with
# HIP events ON THEIR LAUNCH STREAM inside whatever region is running - bench.py's instrumented pass of the timed region.
EVENT_TAP = None


def _tap_begin():
    if EVENT_TAP is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _tap_end(kind, e0, flop):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        EVENT_TAP.append((kind, e0, e1, flop))


def eager_only():
    return RECORD is not None or PROFILE is not None


def time_recorded(records, reps=10):
    """{kind: {ms, flop, launches}} for ONE frame: sum over the frame's launches of the average duration of that
    launch (measured over `reps` back-to-back repetitions)."""
    groups = {}
    for kind, key, flop, fn, keep in records:
        g = groups.setdefault((kind, key), [0, flop, fn, keep])
        g[0] += 1
    out = {}
    for (kind, key), (count, flop, fn, keep) in groups.items():
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        d = out.setdefault(kind, dict(ms=0.0, flop=0.0, launches=0))
        d['ms'] += e0.elapsed_time(e1) / reps * count
        d['flop'] += flop * count
        d['launches'] += count
    return out


def workspace(nbytes, device, tag='default'):
    """Grow-only scratch buffer per (device, tag + scope suffix); kernels on one stream run in order so reuse is safe.
    The scope suffix (`ws_scope`, entered by XMem around its side-stream key-encoder stages, unique per network instance and
    graph slot) keeps concurrently running streams on separate scratch.
    CONTRACT: one host thread drives this module per process (as the reference's InferenceCore is single-threaded).
    The scope suffix is thread-local, but the precision mode (`ops.precision`) is a process-wide switch and un-scoped scratch
    ('conv', 'affinity', 'augment', ... without a scope) is shared per device by every caller on every stream: two host threads
    calling into ops concurrently would clobber each other's mode and scratch.  Run one process per stream of videos (launch.py)."""
    key = (str(device), tag + getattr(_tls, 'suffix', ''))      # kernels on a side stream get their own scratch
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.setdefault(key, []).append(buf)   # captured HIP graphs may still hold the old pointer: kept while the scope lives
        grow = max(int(nbytes), 2 * (buf.numel() if buf is not None else 0), 1 << 20)
        buf = torch.empty(grow, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def _req(t, name, half_ok=False):
    if not t.is_cuda:
        raise RuntimeError(f'{name}: expected a CUDA (HIP) tensor - xmem2_amd has no CPU path')
    if t.dtype != torch.float32 and not (half_ok and t.dtype == torch.float16):
        raise RuntimeError(f'{name}: expected float32' + (' or float16' if half_ok else ''))
    return t


def _h(t):
    """storage flag of the `_t` entry points: 1 = IEEE half, 0 = float"""
    return int(t.dtype == torch.float16)


# Arithmetic mode of the 3x3 / stride-1 convolutions.  'fp32' (default, the parity contract) or 'fp16': OPT-IN reduced
# precision - the Winograd-domain operands are rounded to fp16 and multiplied on the fp16 MFMA with fp32 accumulation (the
# counterpart of the reference's autocast loop, inference/run_on_video.py:76).  Set per call tree by XMem (`precision`).
_PRECISION = 'fp32'
_DRIVER = threading.RLock()          # held while a thread is inside `precision` (a network stage): see precision.__enter__
PRECISIONS = ('fp32', 'fp16', 'fp16w', 'fp32x')
# 'fp16' (opt-in): THE FP16 LOOP - the counterpart of the reference's GPU mode (torch.cuda.amp.autocast around the frame loop,
# inference/run_on_video.py:76; fp32 preload :59-66): activations are IEEE halfs in HBM, every convolution contracts half
# operands on v_mfma_f32_32x32x16_f16 in the direct form with fp32 accumulation and an fp32 epilogue, elementwise kernels
# compute in fp32 on half storage; stems, key projection output, memory (keys, values, readout weights), GRU state, logits and
# probabilities stay fp32.  'fp16w' is round 2's experiment (only the F(2x2) Winograd-domain operands in fp16), kept runnable.
# 'fp32x' (opt-in, separately reported): SPLIT-OPERAND arithmetic - every fp32 operand of every GEMM-shaped convolution is
# carried as two halfs (x = hi + lo, <= 2^-21 relative) and the four partial products run on v_mfma_f32_32x32x16_f16 with fp32
# accumulation (csrc/conv_mfma.hip, SPLIT kernels).  Same tensors, same bytes, fp32-class results; 1/4 of the fp32 MFMA cycles.


def act_dtype():
    """Storage type of the activations the network allocates in the current mode: halfs in the fp16 loop, floats otherwise."""
    return torch.float16 if _PRECISION == 'fp16' else torch.float32


class precision:
    def __init__(self, mode):
        if mode not in PRECISIONS:
            raise ValueError(f'unknown precision {mode!r} ({" | ".join(PRECISIONS)})')
        self.mode = mode

    def __enter__(self):
        global _PRECISION
        # the mode is a process-wide switch and un-scoped scratch is shared per device: ONE host thread may be inside a network stage
        # at a time (the reference's InferenceCore is single-threaded too).  A second thread entering concurrently is a
        # contract violation and fails loudly instead of silently flipping the other thread's arithmetic mode; the same thread
        # may nest, and different threads may take turns.  InferenceCore's public calls (step, prefetch_keys, put_*) hold the same
        # lock for their WHOLE duration with a blocking acquire (inference_core._on_network_device), replays and eager readout
        # included: threads that drive cores of one process are serialised call by call; only a thread that calls into `ops` /
        # XMem stages directly while another is inside one trips this check.
        if not _DRIVER.acquire(blocking=False):
            raise RuntimeError('xmem2_amd.ops: another host thread is inside a network stage - one thread drives the kernels of a process at a '
                               'time (run one process per stream of videos, xmem2_amd.launch)')
        self.prev, _PRECISION = _PRECISION, self.mode

    def __exit__(self, *exc):
        global _PRECISION
        _PRECISION = self.prev
        _DRIVER.release()


class ConvWeights:
    """Device-resident convolution parameters in kernel layout: w [Cout][KH][KW][Cin_pad], scale, shift."""
    __slots__ = ('w', 'scale', 'shift', 'cout', 'cin', 'kh', 'kw', 'stride', 'pad', 'cin_true', 'wu', 'wu_f16', 'wu4',
                 'sp_shift', 'scale_sp', 'w_sp', 'wu_sp', 'wu4_sp', 'w_h')

    def __init__(self, w, scale, shift, stride, pad, cin_true=None, winograd=True):
        self.w, self.scale, self.shift = w, scale, shift
        self.cout, self.kh, self.kw, self.cin = w.shape
        self.stride, self.pad = stride, pad
        self.cin_true = cin_true if cin_true is not None else self.cin     # un-padded Cin (algorithmic FLOPs)
        self.wu = None
        self.wu_f16 = None
        self.wu4 = None                  # F(4x4,3x3) operand, built on first use (only the large layers take that path)
        self.sp_shift = None             # 'fp32x' mode only: split operands (built on first use), see ensure_split
        self.scale_sp = self.w_sp = self.wu_sp = self.wu4_sp = None
        self.w_h = None                  # 'fp16' loop: the weights as halfs [Cout][KH][KW][Cin pad 8], built on first use (half())
        if winograd and self.kh == 3 and self.kw == 3 and stride == 1 and pad == 1 and self.cin % 32 == 0 \
                and self.cout % 4 == 0 and self.cout >= 32:
            self.wu = winograd_weights(w)
            if self.cin % 64 == 0:
                self.wu_f16 = self.wu.to(torch.float16).contiguous()      # reduced-precision mode only


    def half(self):
        """[Cout][KH][KW][Cin'] IEEE halfs, Cin' = Cin padded to a multiple of 8 with zero channels (a 16-byte operand chunk is 8
        halfs); rounded once to nearest even, as autocast casts fp32 weights."""
        if self.w_h is None:
            w = self.w
            if w.shape[3] % 8:
                w = torch.nn.functional.pad(w, (0, 8 - w.shape[3] % 8))
            self.w_h = w.to(torch.float16).contiguous()
        return self.w_h

    def ensure_split(self):
        """Split-operand forms of the weights for the 'fp32x' mode: every array the kernels may contract (direct, F(2x2), F(4x4))
        times ONE power of two 2^s per layer (so that the low halves stay in fp16's normal range), each group of four input
        channels stored as [hi x 4 | lo x 4] halfs; `scale_sp` = scale * 2^-s undoes the factor exactly in the epilogue."""
        if self.sp_shift is None:
            forms = [t for t in (self.w, self.wu, self.wu4) if t is not None]
            amax = max(float(t.abs().max()) for t in forms)
            s = 0 if not (amax > 0 and math.isfinite(amax)) else max(-8, min(24, int(math.floor(math.log2(1024.0 / amax)))))
            self.sp_shift = s
            self.scale_sp = (self.scale * (2.0 ** -s)).contiguous()
            self.w_sp = split_pack(self.w, s)
            self.wu_sp = split_pack(self.wu, s) if self.wu is not None else None
        if self.wu4 is not None and self.wu4_sp is None:
            if float(self.wu4.abs().max()) * 2.0 ** self.sp_shift > 3.0e4:      # built after the shift was chosen and larger than planned
                raise RuntimeError('split operand out of the fp16 range: build the F(4x4) operand before ensure_split()')
            self.wu4_sp = split_pack(self.wu4, self.sp_shift)


def split_pack(w, shift=0):
    """float32 [..., C] (C % 4 == 0) -> float16 [..., 2C]: per four channels [hi0..hi3 | lo0..lo3] of w * 2^shift, hi = the
    nearest half, lo = the nearest half of the remainder (|w 2^shift - hi - lo| <= 2^-22 |w 2^shift| or 3e-8 absolute)."""
    ws = w.float() * (2.0 ** shift)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    c = w.shape[-1]
    hi4 = hi.reshape(*w.shape[:-1], c // 4, 4)
    lo4 = lo.reshape(*w.shape[:-1], c // 4, 4)
    return torch.cat([hi4, lo4], -1).reshape(*w.shape[:-1], 2 * c).contiguous()


_WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


# F(4x4,3x3) with the interpolation points {0, +-3/4, +-3/2, inf} (csrc/conv_mfma.hip: why, and what the transforms look like).
# Row i of G = (1, p_i, p_i^2) / N_i with N_i = prod_{k != i} (p_i - p_k) over the finite points; the row of infinity is (0, 0, 1).
def _wino4_g(a=0.75, b=1.5):
    from fractions import Fraction as Fr
    a, b = Fr(a), Fr(b)
    na, nb = 2 * a * a * (a * a - b * b), 2 * b * b * (b * b - a * a)
    rows = [[1 / (a * a * b * b), 0, 0], [1 / na, a / na, a * a / na], [1 / na, -a / na, a * a / na],
            [1 / nb, b / nb, b * b / nb], [1 / nb, -b / nb, b * b / nb], [0, 0, 1]]
    return torch.tensor([[float(v) for v in r] for r in rows], dtype=torch.float64)       # exact rationals rounded once to fp64


_WINO4_G = _wino4_g()


def winograd4_weights(w):
    """[Cout][3][3][Cin] -> G g G^T of F(4x4,3x3) as [36][Cout][Cin] (load-time, formed in fp64 and rounded once to fp32)."""
    g = _WINO4_G.to(w.device)
    u = torch.einsum('ia,nabc,jb->ijnc', g, w.double(), g)
    return u.reshape(36, w.shape[0], w.shape[3]).to(torch.float32).contiguous()


# F(4x4,3x3) replaces F(2x2,3x3) for outputs of at least this many pixels per image (1/8 resolution of 480p and up): below
# that the 36 tile-position GEMMs are too small to fill the chip.  XMEM_WINO4=0 turns it off (F(2x2) everywhere).
WINO4_MIN_PIXELS = int(os.environ.get('XMEM_WINO4_MIN_PIXELS', '4096'))   # applies to F(2x2) entries of the plan table only
WINO4 = os.environ.get('XMEM_WINO4', '1') != '0'
# Tools knob (parity attribution, tools/parity_by_plan.py): 'direct' runs every convolution in the direct implicit-GEMM form,
# 'f2' replaces F(4x4) by F(2x2), 'direct_sk2' / 'direct_sk3' = the direct form summed in 2 / 3 slabs; None / '' = the shipped plan table.  Read at call time so that a tool can switch it.
CONV_FORM = os.environ.get('XMEM_CONV_FORM') or None


def winograd_weights(w):
    """[Cout][3][3][Cin] -> G g G^T as [16][Cout][Cin] (load-time, fp32; F(2x2,3x3) of Lavin & Gray)."""
    g = _WINO_G.to(w.device)
    u = torch.einsum('ia,nabc,jb->ijnc', g, w, g)
    return u.reshape(16, w.shape[0], w.shape[3]).contiguous()


# ---- convolution plans ---------------------------------------------------------------------------------------
# Tile shape / split-K of the implicit-GEMM kernel are chosen per layer shape.  Plans measured on an MI355X are
# shipped in conv_plans.json (deterministic: the same plan -> the same summation order); shapes not listed there
# take the library's built-in deterministic heuristic (same shape -> same tiles -> same summation order on every machine).
# XMEM_CONV_AUTOTUNE=1 opts in to timing the candidates at first use (tools/tune_convs.py does, to refresh conv_plans.json).
AUTOTUNE = os.environ.get('XMEM_CONV_AUTOTUNE', '0') == '1'
_PLAN_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_plans.json')
_PLAN_FILE_X = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_plans_fp32x.json')   # measured with the split kernels
_plans = None
_plans_x = None
_tuned_now = {}
_tuned_now_x = {}


def _read_plan_file(path):
    if os.path.exists(path):
        try:
            return {k: tuple(v) for k, v in json.load(open(path)).items()}
        except Exception:
            pass
    return {}


def _load_plans(split=False):
    """Plan table of the fp32 kernels, or (split=True) of the split-operand kernels: the GEMMs are ~4x cheaper there, so other
    tiles win; shapes the fp32x table does not list fall back to the fp32 entry."""
    global _plans, _plans_x
    if _plans is None:
        _plans = _read_plan_file(_PLAN_FILE)
    if not split:
        return _plans
    if _plans_x is None:
        _plans_x = _read_plan_file(_PLAN_FILE_X)
    return _plans_x


def _lookup_plan(key, split):
    if split:
        p = _load_plans(True).get(key) or _tuned_now_x.get(key)
        if p is not None or AUTOTUNE:                 # the tuner measures the split kernels instead of inheriting the fp32 entry
            return p
        return _load_plans().get(key) or _tuned_now.get(key)
    return _load_plans().get(key) or _tuned_now.get(key)


def dump_tuned_plans(path, split=False):
    """Write every plan known to this process (shipped + tuned now) - used to refresh conv_plans.json."""
    if _PRECISION == 'fp32x' or split:
        allp = dict(_load_plans(True))
        allp.update(_tuned_now_x)
        with open(path, 'w') as f:
            json.dump({k: list(v) for k, v in sorted(allp.items())}, f, indent=0)
        return len(allp)
    allp = dict(_load_plans())
    allp.update(_tuned_now)
    if WINO4 and (os.environ.get('XMEM_RETUNE_ALL') or '__tuned_with_f4__' in allp):
        allp['__tuned_with_f4__'] = (1, 0)           # EVERY entry was measured against the F(4x4) candidates (full retune only)
    with open(path, 'w') as f:
        json.dump({k: list(v) for k, v in sorted(allp.items())}, f, indent=0)
    return len(allp)


# Tools (tools/tune_convs.py): XMEM_RETUNE_MARGIN=0.05 re-measures every TABLED shape against its candidates and replaces the tabled plan
# only when a candidate is more than that fraction faster (12-launch timings, best of two): a table refresh after a kernel change
# without the churn of equal-within-noise entries.
RETUNE_MARGIN = float(os.environ.get('XMEM_RETUNE_MARGIN', '0') or 0)
_retuned = set()


def _tune_conv(lib, d, x_device, cw=None, incumbent=None):
    """Time the candidate (tile, split-K) plans for this descriptor; returns the fastest (or `incumbent` unless beaten by RETUNE_MARGIN)."""
    Ho = (d.H + 2 * d.pad - d.KH) // d.stride + 1
    Wo = (d.W + 2 * d.pad - d.KW) // d.stride + 1
    M, K = d.B * Ho * Wo, d.KH * d.KW * d.Cin
    best, best_t = (0, 0), None
    tiles = {1: (128, 128, 32), 2: (128, 64, 32), 3: (64, 64, 32), 4: (128, 128, 64), 5: (128, 64, 64), 6: (64, 64, 64)}
    cands = list(tiles.items())
    if d.w_winograd and d.ldout % 4 == 0 and (not d.res or d.ldres % 4 == 0):
        cands += [(t + 6, cfg) for t, cfg in tiles.items()]
        cands += [(13, (128, 64, 32)), (14, (64, 64, 32)), (15, (64, 128, 32))]
        if cw is not None and WINO4 and Ho * Wo >= 256:          # F(4x4,3x3): the same GEMM tiles over 36 positions
            if cw.wu4 is None:
                cw.wu4 = winograd4_weights(cw.w)
            d.w_winograd4 = cw.wu4.data_ptr()
            if d.arith == 1:
                cw.ensure_split()
                d.w_winograd4_split = cw.wu4_sp.data_ptr()
            cands += [(t + 16, cfg) for t, cfg in tiles.items()]
    for tile, (bm, bn, bk) in cands:
        if bn == 128 and d.Cout <= 64 and tile != 15:
            continue
        nt = -(-M // bm) * -(-d.Cout // bn)
        nk = -(-K // bk)
        for sk in ((1,) if tile > 6 else (1, 2, 3, 4, 6, 8, 12, 16)):
            if tile > 16 and nt * 36 < 128:
                continue
            if sk > 1 and (nt * sk > 2048 or nk // sk < 2):
                continue
            if sk == 1 and nt < 48 and nk >= 16:
                continue
            if incumbent is not None:
                t = _time_plan(lib, d, x_device, (tile, sk))
                if t is not None and (best_t is None or t < best_t):
                    best, best_t = (tile, sk), t
                continue
            d.plan_tile, d.plan_splitk = tile, sk
            need = lib.xmem_conv2d_workspace_bytes(C.byref(d))
            ws = workspace(need, x_device, 'conv') if need else None
            st = stream_ptr()
            for _ in range(2):
                if lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, st) != 0:
                    break
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, st)
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1)
                if best_t is None or t < best_t:
                    best, best_t = (tile, sk), t
    if incumbent is not None:
        t_inc = _time_plan(lib, d, x_device, tuple(incumbent))
        if t_inc is not None and (best_t is None or best_t > (1.0 - RETUNE_MARGIN) * t_inc):
            return tuple(incumbent)
        print(f'[retune] {tuple(incumbent)} {t_inc and round(t_inc * 1e3, 1)} us -> {best} {best_t and round(best_t * 1e3, 1)} us', file=sys.stderr)
    return best


# Streaming-GEMM variants of a tabled Winograd plan (csrc/gemm_stream.hip): plan 19 (F(4x4), 64x64 tile) -> 23 (64x64, ring 3),
# 26 (64x64, ring 4), 24 (128x64, ring 3); plan 9 (F(2x2)) -> 29, 32.  XMEM_TUNE_STREAM=1 (tools/tune_convs.py) times the tabled
# plan against them once per shape and keeps a variant only when it is at least 3 % faster (same arithmetic, same summation
# order: results are bit-identical either way).
TUNE_STREAM = os.environ.get('XMEM_TUNE_STREAM', '0') == '1'
_STREAM_VARIANTS = {19: (23, 26, 24), 9: (29, 32)}
_stream_checked = set()


def _time_plan(lib, d, dev, plan, reps=12):
    d.plan_tile, d.plan_splitk = plan
    need = lib.xmem_conv2d_workspace_bytes(C.byref(d))
    ws = workspace(need, dev, 'conv') if need else None
    st = stream_ptr()
    for _ in range(3):
        if lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, st) != 0:
            return None
    best = None
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, st)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) / reps
        best = t if best is None or t < best else best
    return best


def _tune_stream(lib, d, dev, plan):
    base = _time_plan(lib, d, dev, plan)
    best, best_t = plan, base
    if base is not None:
        for t in _STREAM_VARIANTS[plan[0]]:
            tt = _time_plan(lib, d, dev, (t, 1))
            if tt is not None and tt < 0.97 * base and tt < best_t:
                best, best_t = (t, 1), tt
    return best


_TILES = {1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (128, 128), 5: (128, 64), 6: (64, 64)}
_STREAM_TILES = ((64, 64), (128, 64), (128, 128))


def conv_executed_mfma_flops(B, Ho, Wo, cin, cout, kh, kw, stride, pad, plan_tile, winograd_ok):
    """MFMA FLOPs the library really issues for one conv2d call under `plan_tile` (what an EXECUTED roofline fraction must count,
    bench.py conv_roofline): the direct form contracts 2 * M * Cout * KH * KW * Cin with M / Cout padded to the tile and K to the
    32-deep k-tile; F(2x2) runs 16 position GEMMs over ceil(Ho/2) * ceil(Wo/2) tiles per image (1/2.25 of the direct FLOPs before
    padding), F(4x4) 36 over ceil(Ho/4) * ceil(Wo/4) (1/4).  Cout = 1 is a VALU GEMV: no MFMA work.  Mirrors make_plan() of
    csrc/conv_mfma.hip; the built-in heuristic (plan 0) is counted with the 64x64 tile."""
    if cout == 1:
        return 0.0
    t = int(plan_tile)
    up = lambda a, b: -(-a // b) * b
    form, tile = 'direct', (64, 64)
    if 23 <= t <= 40:
        grp, v = (t - 23) // 6, (t - 23) % 6
        tile = _STREAM_TILES[v % 3]
        form = ('f4', 'f2', 'direct')[grp]
    elif 17 <= t <= 22:
        form, tile = 'f4', _TILES[t - 16]
    elif t == 16:
        form, tile = 'f2', (64, 64)
    elif 13 <= t <= 15:
        form, tile = 'f2', {13: (128, 64), 14: (64, 64), 15: (64, 128)}[t]
    elif 7 <= t <= 12:
        form, tile = 'f2', _TILES[t - 6]
    elif 1 <= t <= 6:
        tile = _TILES[t]
    if form != 'direct' and not winograd_ok:
        form = 'direct'
    if form == 'direct':
        return 2.0 * up(B * Ho * Wo, tile[0]) * up(cout, tile[1]) * up(kh * kw * cin, 32)
    r, npos = (4, 36) if form == 'f4' else (2, 16)
    tiles = B * (-(-Ho // r)) * (-(-Wo // r))
    return npos * 2.0 * up(tiles, tile[0]) * up(cout, tile[1]) * up(cin, 32)


_PLAN_FILE_H = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_plans_fp16.json')   # measured with the half kernels
_plans_h = None
_tuned_now_h = {}


def _conv2d_half(lib, x, cw, out, out_ld, res, relu_in, relu_out, in_ld, cin, plan, res_broadcast, out_dtype):
    """The fp16 loop's convolution: x [B,H,W,C] halfs (pixel stride in_ld halfs), half weights, direct implicit GEMM on the fp16
    MFMA, fp32 accumulation + epilogue; output (and residual) halfs, or float32 with out_dtype=torch.float32 (key projection,
    mask head).  Input channels beyond the layer's own (a buffer padded to a multiple of 8) must be zero: the half weights are
    zero there."""
    global _plans_h
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    ldin = in_ld if in_ld is not None else x.shape[3]
    wh = cw.half()
    cin_h = wh.shape[3]                                   # the layer's Cin padded to 8
    cin = cin if cin is not None else cin_h
    if cin != cin_h and cin != cw.cin:
        raise RuntimeError(f'conv2d (half): weight expects Cin={cw.cin} (padded {cin_h}), got {cin}')
    if cin_h > ldin:
        raise RuntimeError(f'conv2d (half): the input buffer has {ldin} channels per pixel, the layer reads {cin_h} (pad to a multiple of 8)')
    if in_ld is not None:
        # a channel SLICE of a wider buffer: the kernel reads cin_h (= Cin padded to 8) halfs from the slice start whatever `cin` says,
        # and only the zero WEIGHTS mask the tail - 0 x Inf/NaN from a neighbouring slice or an uninitialised tail would be NaN.
        # So a slice must be a whole number of 8-half groups and must end inside its pixel (include/xmem_hip.h, xmem_conv_desc.in_half).
        # (A whole, zero-padded buffer - in_ld None - is not a slice: its padding channels are the caller's zeros.)
        if cin % 8 != 0:
            raise RuntimeError(f'conv2d (half): a channel slice must hold a multiple of 8 channels, got cin={cin} '
                               '(pad the slice and zero-fill the padding channels)')
        base = x._base
        if base is not None and base.dim() == 4 and base.shape[3] == ldin and base.is_contiguous():
            # offset of the slice inside its pixel, measured from the PARENT buffer (which may itself sit anywhere in a workspace)
            inpix = (x.storage_offset() - base.storage_offset()) % ldin
            if inpix + cin_h > ldin:
                raise RuntimeError(f'conv2d (half): the slice [{inpix}, +{cin_h}) crosses the pixel stride {ldin}')
    Ho = (H + 2 * cw.pad - cw.kh) // cw.stride + 1
    Wo = (W + 2 * cw.pad - cw.kw) // cw.stride + 1
    odt = out_dtype if out_dtype is not None else (out.dtype if out is not None else torch.float16)
    if out is None:
        out = torch.empty((B, Ho, Wo, cw.cout), dtype=odt, device=x.device)
        out_ld = cw.cout
    elif out_ld is None:
        out_ld = out.shape[-1]
    if out.dtype != odt:
        raise RuntimeError('conv2d (half): out buffer dtype does not match out_dtype')
    if res is not None and res.dtype != out.dtype:
        raise RuntimeError('conv2d (half): the residual must have the output storage type')
    d = ConvDesc()
    d.inp = x.data_ptr(); d.B, d.H, d.W, d.Cin, d.ldin = B, H, W, cin_h, ldin
    d.w = cw.w.data_ptr(); d.Cout, d.KH, d.KW, d.stride, d.pad = cw.cout, cw.kh, cw.kw, cw.stride, cw.pad
    d.scale = cw.scale.data_ptr(); d.shift = cw.shift.data_ptr()
    d.res = res.data_ptr() if res is not None else None
    d.ldres = res.shape[-1] if res is not None else 0
    d.res_broadcast = int(bool(res_broadcast and res is not None))
    d.out = out.data_ptr(); d.ldout = out_ld
    d.relu_in, d.relu_out = int(relu_in), int(relu_out)
    d.in_half, d.out_half, d.w_half = 1, int(out.dtype == torch.float16), wh.data_ptr()
    key = f'h{B}x{H}x{W}x{cin_h}/{ldin}->{cw.cout}/{out_ld} k{cw.kh}s{cw.stride}p{cw.pad} r{int(res is not None)}{int(relu_in)}{int(relu_out)}o{d.out_half}'
    if plan is None:
        if _plans_h is None:
            _plans_h = _read_plan_file(_PLAN_FILE_H)
        plan = _plans_h.get(key) or _tuned_now_h.get(key)
        if plan is None:
            plan = (0, 0)
            if AUTOTUNE and cw.cout > 1 and not torch.cuda.is_current_stream_capturing():
                best, best_t = (0, 0), None
                for tile in (1, 2, 3, 4):                 # 4 = 256x128, 8 waves (half kernels only)
                    for sk in (1, 2, 4, 8):
                        t = _time_plan(lib, d, x.device, (tile, sk), reps=8)
                        if t is not None and (best_t is None or t < best_t):
                            best, best_t = (tile, sk), t
                plan = best
            _tuned_now_h[key] = plan
    d.plan_tile, d.plan_splitk = plan
    need = lib.xmem_conv2d_workspace_bytes(C.byref(d))
    ws = workspace(need, x.device, 'conv') if need else None
    check(lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, stream_ptr()))
    if RECORD is not None:
        RECORD.append(('conv', key, 2.0 * B * Ho * Wo * cw.cout * cw.kh * cw.kw * cw.cin_true,
                       lambda: lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, stream_ptr()),
                       (x, out, res, cw, ws, dict(relu_in=bool(relu_in), relu_out=bool(relu_out), in_ld=ldin, cin=cin_h, out_ld=out_ld,
                                                  res_broadcast=bool(res_broadcast), plan=tuple(plan),
                                                  executed_mfma_flops=conv_executed_mfma_flops(B, Ho, Wo, cin_h, cw.cout, cw.kh, cw.kw, cw.stride,
                                                                                               cw.pad, {4: 1}.get(plan[0], plan[0]) if plan[0] <= 6 else 0, False)))))
    return out


def dump_tuned_plans_half(path):
    allp = dict(_plans_h or _read_plan_file(_PLAN_FILE_H))
    allp.update(_tuned_now_h)
    with open(path, 'w') as f:
        json.dump({k: list(v) for k, v in sorted(allp.items())}, f, indent=0)
    return len(allp)


def conv2d(x, cw, out=None, out_ld=None, res=None, relu_in=False, relu_out=False, in_ld=None, cin=None, plan=None,
           res_broadcast=False, out_dtype=None):
    """x [B,H,W,C] NHWC (or any buffer whose pixel stride is `in_ld`) -> out [B,Ho,Wo,Cout]."""
    lib = load()
    _req(x, 'conv2d input', half_ok=True)
    if x.dtype == torch.float16:
        return _conv2d_half(lib, x, cw, out, out_ld, res, relu_in, relu_out, in_ld, cin, plan, res_broadcast, out_dtype)
    if out_dtype is not None and out_dtype != torch.float32:
        raise RuntimeError('conv2d: a float32 input gives a float32 output (the fp16 loop converts at the max-pool after the stems)')
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    ldin = in_ld if in_ld is not None else x.shape[3]
    cin = cin if cin is not None else cw.cin
    if cin != cw.cin:
        raise RuntimeError(f'conv2d: weight expects Cin={cw.cin}, got {cin}')
    Ho = (H + 2 * cw.pad - cw.kh) // cw.stride + 1
    Wo = (W + 2 * cw.pad - cw.kw) // cw.stride + 1
    guard = None
    if out is None:
        if _GUARD and not torch.cuda.is_current_stream_capturing():      # tools: sentinel zones around the output (XMEM_GUARD=1)
            n, G = B * Ho * Wo * cw.cout, 65536
            flat = torch.full((n + 2 * G,), 12345.0, dtype=torch.float32, device=x.device)
            out = flat[G:G + n].view(B, Ho, Wo, cw.cout)
            guard = (flat, G, n)
        else:
            out = torch.empty((B, Ho, Wo, cw.cout), dtype=torch.float32, device=x.device)
        out_ld = cw.cout
    elif out_ld is None:
        out_ld = out.shape[-1]
    d = ConvDesc()
    d.inp = x.data_ptr(); d.B, d.H, d.W, d.Cin, d.ldin = B, H, W, cin, ldin
    d.w = cw.w.data_ptr(); d.Cout, d.KH, d.KW, d.stride, d.pad = cw.cout, cw.kh, cw.kw, cw.stride, cw.pad
    d.scale = cw.scale.data_ptr(); d.shift = cw.shift.data_ptr()
    d.res = res.data_ptr() if res is not None else None
    d.ldres = res.shape[-1] if res is not None else 0
    d.res_broadcast = int(bool(res_broadcast and res is not None))   # res [1,Ho,Wo,C] added to every batch element
    d.out = out.data_ptr(); d.ldout = out_ld
    d.relu_in, d.relu_out = int(relu_in), int(relu_out)
    explicit = plan is not None                  # a caller-given plan is taken literally (tests, the tuner)
    d.w_winograd = cw.wu.data_ptr() if cw.wu is not None else None
    d.w_winograd_f16 = None
    if _PRECISION == 'fp16w' and cw.wu_f16 is not None and plan is None and out_ld % 4 == 0 and (res is None or res.shape[-1] % 4 == 0):
        d.w_winograd_f16 = cw.wu_f16.data_ptr()
        plan = (16, 1)                       # the library falls back to the fp32 Winograd tile if its own conditions fail
    key = f'{B}x{H}x{W}x{cin}/{ldin}->{cw.cout}/{out_ld} k{cw.kh}s{cw.stride}p{cw.pad} r{int(res is not None)}{int(relu_in)}{int(relu_out)}'
    d.arith = 0
    d.w_split = d.w_winograd_split = d.w_winograd4_split = None
    split = _PRECISION == 'fp32x' and cw.cout > 1
    if split:
        # split-operand arithmetic for every GEMM-shaped path (the Cout = 1 mask head is a GEMV on the fp32 VALU)
        if cw.sp_shift is None or (cw.wu4 is not None and cw.wu4_sp is None):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('conv2d: split operands must be built before graph capture (run the stage eagerly once)')
            if cw.sp_shift is None and cw.wu is not None and cw.wu4 is None and WINO4:
                cw.wu4 = winograd4_weights(cw.w)          # so that ONE power of two covers every form of this layer
            cw.ensure_split()
        d.arith = 1
        d.w_split = cw.w_sp.data_ptr()
        d.w_winograd_split = cw.wu_sp.data_ptr() if cw.wu_sp is not None else None
        d.w_winograd4_split = cw.wu4_sp.data_ptr() if cw.wu4_sp is not None else None    # taken only by an F(4x4) plan
        d.scale = cw.scale_sp.data_ptr()
    if plan is None and CONV_FORM == 'direct' and _PRECISION == 'fp32':
        plan, explicit = (0, 0), True            # the library's deterministic direct-form heuristic
    elif plan is None and CONV_FORM in ('direct_sk2', 'direct_sk3') and _PRECISION == 'fp32':
        # the direct form with every contraction cut into 2 / 3 slabs that are summed afterwards: the SAME products in another fp32
        # summation order (tests/parity_by_plan.py: how much of a parity margin is the order of additions alone)
        plan, explicit = (3, int(CONV_FORM[-1])), True
    if plan is None:
        plan = _lookup_plan(key, split)
        if plan is not None and RETUNE_MARGIN > 0 and AUTOTUNE and not split and _PRECISION == 'fp32' and cw.cout > 1 and key not in _retuned \
                and not torch.cuda.is_current_stream_capturing():
            _retuned.add(key)                     # tools: the tabled plan against its candidates, replaced only when clearly beaten
            if 17 <= plan[0] <= 28 and cw.wu4 is None and cw.wu is not None:
                cw.wu4 = winograd4_weights(cw.w)
            if cw.wu4 is not None:
                d.w_winograd4 = cw.wu4.data_ptr()
            new_plan = _tune_conv(lib, d, x.device, cw, incumbent=tuple(plan))
            if tuple(new_plan) != tuple(plan):
                plan = tuple(new_plan)
                _tuned_now[key] = plan
    if plan is None:
        plan = (0, 0)
        if AUTOTUNE and cw.cout > 1 and not torch.cuda.is_current_stream_capturing():
            plan = _tune_conv(lib, d, x.device, cw)
        elif cw.wu is not None and out_ld % 4 == 0 and (res is None or res.shape[-1] % 4 == 0):
            # a shape the shipped table does not know (another resolution / object count): the 3x3 stride-1 layers still take
            # Winograd with the 64x64 GEMM tile - F(4x4) from 1/8 resolution of 480p up, F(2x2) below - instead of the direct
            # form (deterministic: same shape -> same plan on every machine)
            plan = (19, 1) if (WINO4 and CONV_FORM != 'f2' and B * Ho * Wo >= WINO4_MIN_PIXELS) else (9, 1)
        (_tuned_now_x if split else _tuned_now)[key] = plan
    d.w_winograd4 = None
    if not explicit and 7 <= plan[0] <= 12 and WINO4 and CONV_FORM != 'f2' and cw.wu is not None and Ho * Wo >= WINO4_MIN_PIXELS and _PRECISION == 'fp32' \
            and '__tuned_with_f4__' not in _load_plans():      # a table tuned against F(4x4) already says which layers take it
        if cw.wu4 is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('conv2d: the F(4x4) operand must be built before graph capture (run the stage eagerly once)')
            cw.wu4 = winograd4_weights(cw.w)
        d.w_winograd4 = cw.wu4.data_ptr()
        plan = (plan[0] + 10, plan[1])
    elif (17 <= plan[0] <= 22 or 23 <= plan[0] <= 28) and (not WINO4 or CONV_FORM == 'f2') and not explicit:
        plan = (plan[0] - 10 if plan[0] <= 22 else plan[0] + 6, plan[1])     # XMEM_WINO4=0: the same GEMM tile under F(2x2)
    elif 17 <= plan[0] <= 28:                        # F(4x4): classic tiles 17..22, streaming GEMM 23..28
        if cw.wu4 is None and cw.wu is not None:
            cw.wu4 = winograd4_weights(cw.w)
        d.w_winograd4 = cw.wu4.data_ptr() if cw.wu4 is not None else None
    if split and cw.wu4 is not None and cw.wu4_sp is None:          # the F(4x4) operand was built just above
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('conv2d: split operands must be built before graph capture (run the stage eagerly once)')
        cw.ensure_split()
        d.w_winograd4_split = cw.wu4_sp.data_ptr()
    if TUNE_STREAM and not explicit and plan[0] in _STREAM_VARIANTS and d.arith == 0 and key not in _stream_checked \
            and not torch.cuda.is_current_stream_capturing():
        _stream_checked.add(key)
        plan = _tune_stream(lib, d, x.device, tuple(plan))
        _tuned_now[key] = plan
    d.plan_tile, d.plan_splitk = plan
    if _SPLIT_CHECK and d.arith == 1 and not torch.cuda.is_current_stream_capturing():
        # tools: the same call in fp32 first, then compare (XMEM_SPLIT_CHECK=1; synchronises)
        d.arith, sc = 0, d.scale
        d.scale = cw.scale.data_ptr()
        need = lib.xmem_conv2d_workspace_bytes(C.byref(d))
        ws = workspace(need, x.device, 'conv') if need else None
        check(lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, stream_ptr()))
        want = out.clone()
        d.arith, d.scale = 1, sc
    need = lib.xmem_conv2d_workspace_bytes(C.byref(d))
    ws = workspace(need + (1 << 20 if guard else 0), x.device, 'conv') if need else None
    if guard and ws is not None:
        ws[need:need + (1 << 20)].fill_(0x5a)
    check(lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, stream_ptr()))
    if guard:
        flat, G, n = guard
        bad_lo, bad_hi = int((flat[:G] != 12345.0).sum()), int((flat[G + n:] != 12345.0).sum())
        bad_ws = int((ws[need:need + (1 << 20)] != 0x5a).sum()) if ws is not None else 0
        if bad_lo or bad_hi or bad_ws:
            print(f'[guard] {key} plan={tuple(plan)} arith={d.arith}: {bad_lo} floats written BEFORE the output, {bad_hi} AFTER it, '
                  f'{bad_ws} bytes past the workspace', file=sys.stderr)
    if _SPLIT_CHECK and d.arith == 1 and not torch.cuda.is_current_stream_capturing():
        err = float((out - want).abs().max()) / max(float(want.abs().max()), 1e-30)
        print(f'[split check] {key} plan={tuple(plan)} in_ld={ldin} max |fp32x - fp32| / max|fp32| = {err:.2e}' +
              ('   <-- MISMATCH' if not err < 1e-3 else ''), file=sys.stderr)
    if RECORD is not None:
        RECORD.append(('conv', key, 2.0 * B * Ho * Wo * cw.cout * cw.kh * cw.kw * cw.cin_true,
                       lambda: lib.xmem_conv2d_nhwc(C.byref(d), ptr(ws), need, stream_ptr()),
                       (x, out, res, cw, ws, dict(relu_in=bool(relu_in), relu_out=bool(relu_out), in_ld=ldin, cin=cin, out_ld=out_ld,
                                                  res_broadcast=bool(res_broadcast), plan=tuple(plan),
                                                  executed_mfma_flops=conv_executed_mfma_flops(
                                                      B, Ho, Wo, cin, cw.cout, cw.kh, cw.kw, cw.stride, cw.pad, plan[0],
                                                      bool(d.w_winograd) and out_ld % 4 == 0 and (res is None or res.shape[-1] % 4 == 0))))))
    return out


_HIPRT = None


def masked_stream(device, n_cus, first=0):
    """A HIP stream whose kernels may only run on `n_cus` compute units (hipExtStreamCreateWithCUMask), wrapped for torch.
    The mask's bits are dealt round-robin over the 8 XCDs by the driver, so the low n_cus bits are n_cus / 8 CUs of every XCD.
    Measurement knob for the frame pipeline's side stream (XMEM_SIDE_CUS); the default stream of a core is unmasked."""
    global _HIPRT
    if _HIPRT is None:
        _HIPRT = C.CDLL('libamdhip64.so')
    words = (first + n_cus + 31) // 32
    mask = (C.c_uint32 * words)()
    for b in range(first, first + n_cus):
        mask[b // 32] |= (1 << (b % 32))
    st = C.c_void_p()
    with torch.cuda.device(device):
        rc = _HIPRT.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask failed with status {rc}')
    return torch.cuda.ExternalStream(st.value, device=device)


def side_stream(device):
    """The stream `InferenceCore.prefetch_keys` runs the batched key encoder on."""
    n = int(os.environ.get('XMEM_SIDE_CUS', '0') or 0)
    return masked_stream(device, n) if n > 0 else torch.cuda.Stream(device=device)


def readout_stream(device):
    """The stream the early readout (the NEXT hinted frame's select + readout) runs on, under the current frame's decoder.
    XMEM_READOUT_PRIORITY (tools: A/B) = the stream's priority: -1 high (its short chain of kernels is dispatched ahead of the decoder's
    as CUs free up), 0 normal."""
    pr = int(os.environ.get('XMEM_READOUT_PRIORITY', '0') or 0)
    return torch.cuda.Stream(device=device, priority=pr)


def trace_marker(tag=0):
    """Empty kernel `xmem_trace_marker_kernel` on the current stream: cuts a rocprofv3 kernel trace to a region."""
    check(load().xmem_trace_marker(int(tag), stream_ptr()))


# The elementwise wrappers take float32 or (the fp16 loop) float16 activations: the `_t` entry points carry the storage type of
# every tensor; outputs follow the input's type unless `out_dtype` says otherwise.
def maxpool3x3s2(x, out_dtype=None):
    B, H, W, Cc = x.shape
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=out_dtype or x.dtype, device=x.device)
    check(load().xmem_maxpool3x3s2_t(ptr(x), _h(x), ptr(out), _h(out), B, H, W, Cc, stream_ptr()))
    return out


def upsample2x_add(g, skip):
    B, h, w, Cc = g.shape
    if skip.dtype != g.dtype:
        raise RuntimeError('upsample2x_add: g and skip must share a storage type')
    out = torch.empty((B, 2 * h, 2 * w, Cc), dtype=g.dtype, device=g.device)
    check(load().xmem_upsample2x_add_t(ptr(g), ptr(skip), ptr(out), _h(g), B, h, w, Cc, stream_ptr()))
    return out


def area_downsample(x, r, out=None, out_ld=None, out_off=0, c=None, in_ld=None):
    B, H, W = x.shape[:3]
    c = c if c is not None else x.shape[3]
    in_ld = in_ld if in_ld is not None else x.shape[3]
    if out is None:
        out = torch.empty((B, H // r, W // r, c), dtype=x.dtype, device=x.device)
        out_ld = c
    check(load().xmem_area_downsample_t(ptr(x), _h(x), in_ld, C.c_void_p(out.data_ptr() + out.element_size() * out_off), _h(out), out_ld,
                                        B, H, W, c, r, stream_ptr()))
    return out


def copy_channels(src, dst, dst_off, c=None, src_off=0):
    """dst[b, p, dst_off:dst_off+c] = src[b % srcB, p, src_off:src_off+c]; src/dst are [B,H,W,C] buffers (float32 or float16 each:
    the copy converts)."""
    B = dst.shape[0]
    P = dst.shape[1] * dst.shape[2]
    c = c if c is not None else src.shape[3]
    check(load().xmem_copy_channels_t(C.c_void_p(src.data_ptr() + src.element_size() * src_off), _h(src), src.shape[3], src.shape[0],
                                      C.c_void_p(dst.data_ptr() + dst.element_size() * dst_off), _h(dst), dst.shape[3], B, P, c, stream_ptr()))
    return dst


def hidden_update_gather(g16, g8, g4, logits, out):
    """out[..., :c16+c8+c4+1] = [g16 | area2(g8) | area4(g4) | area4(logits)] in one launch (float32 NHWC; HiddenUpdater's input,
    model/modules.py:49-57); the same bits as copy_channels + three area_downsample calls."""
    K, h, w, c16 = g16.shape
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in (g16, g8, g4, logits, out)):
        raise RuntimeError('hidden_update_gather: contiguous float32 tensors only')
    if tuple(g8.shape[:3]) != (K, 2 * h, 2 * w) or tuple(g4.shape[:3]) != (K, 4 * h, 4 * w) or tuple(logits.shape) != (K, 4 * h, 4 * w, 1) \
            or tuple(out.shape[:3]) != (K, h, w):
        raise RuntimeError('hidden_update_gather: shapes do not belong to one decoder pass')
    check(load().xmem_hidden_update_gather(ptr(g16), c16, ptr(g8), g8.shape[3], ptr(g4), g4.shape[3], ptr(logits), ptr(out), out.shape[3],
                                           K, h, w, stream_ptr()))
    return out


def cbam_residual(g, p):
    """out = g + CBAM(g); p = dict(w1,b1,w2,b2,sw,sb) device tensors (float32); g float32 or float16."""
    lib = load()
    B, H, W, Cc = g.shape
    out = torch.empty_like(g)
    need = lib.xmem_cbam_workspace_bytes(B, H * W, Cc)
    ws = workspace(need, g.device, 'cbam')
    check(lib.xmem_cbam_residual_t(ptr(g), ptr(out), _h(g), B, H, W, Cc, ptr(p['w1']), ptr(p['b1']), ptr(p['w2']), ptr(p['b2']),
                                   ptr(p['sw']), ptr(p['sb']), ptr(ws), need, stream_ptr()))
    return out


def gru_gate(values, h, out=None):
    """out may be h itself (in-place update of the hidden state).  The state is float32; `values` float32 or float16."""
    B, H, W, Ch = h.shape
    if out is None:
        out = torch.empty_like(h)
    check(load().xmem_gru_gate_t(ptr(values), _h(values), ptr(h), ptr(out), B, H * W, Ch, stream_ptr()))
    return out


def pack_image(img, Hp, Wp, lh, lw, out=None):
    """img [3,H,W] -> [1,Hp,Wp,4] zero padded NHWC (into `out` when given: one frame of a batched buffer)."""
    _req(img, 'image')
    if not img.is_contiguous():
        img = img.contiguous()
    if out is None:
        out = torch.empty((1, Hp, Wp, 4), dtype=torch.float32, device=img.device)
    check(load().xmem_pack_image(ptr(img), ptr(out), img.shape[1], img.shape[2], Hp, Wp, lh, lw, stream_ptr()))
    return out


IM_MEAN = (0.485, 0.456, 0.406)      # dataset/range_transform.py:5-8
IM_STD = (0.229, 0.224, 0.225)


def pack_image_u8(img, Hp, Wp, lh, lw, mean=IM_MEAN, std=IM_STD, out=None):
    """decoded frame uint8 [H,W,3] -> normalised, zero padded [1,Hp,Wp,4] (ToTensor + Normalize + pad in one kernel)."""
    if not img.is_cuda or img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise RuntimeError('pack_image_u8: expected a CUDA (HIP) uint8 tensor of shape [H, W, 3]')
    if not img.is_contiguous():
        img = img.contiguous()
    if out is None:
        out = torch.empty((1, Hp, Wp, 4), dtype=torch.float32, device=img.device)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(load().xmem_pack_image_u8(ptr(img), ptr(out), img.shape[0], img.shape[1], Hp, Wp, lh, lw, m3, s3, stream_ptr()))
    return out


def pack_value_input(image4, masks):
    """image4 [1,Hp,Wp,4], masks [K,Hp,Wp] -> [K,Hp,Wp,8]."""
    K, Hp, Wp = masks.shape
    out = torch.empty((K, Hp, Wp, 8), dtype=torch.float32, device=masks.device)
    check(load().xmem_pack_value_input(ptr(image4), ptr(masks), ptr(out), K, Hp, Wp, stream_ptr()))
    return out


def key_post(proj, ck, need_s=True, need_e=True):
    """proj [1,h,w,ld] -> key [h*w,ck], shrinkage [h*w] | None, selection [h*w,ck] | None."""
    P = proj.shape[0] * proj.shape[1] * proj.shape[2]
    key = torch.empty((P, ck), dtype=torch.float32, device=proj.device)
    shr = torch.empty((P,), dtype=torch.float32, device=proj.device) if need_s else None
    sel = torch.empty((P, ck), dtype=torch.float32, device=proj.device) if need_e else None
    check(load().xmem_key_post(ptr(proj), proj.shape[3], ptr(key), ptr(shr), ptr(sel), P, ck, stream_ptr()))
    return key, shr, sel


def logits_to_prob(logits, H, W, lh, lw, want_padded=True):
    """logits [K,h4,w4] -> prob [K+1,H,W] (cropped) and [K+1,4h4,4w4] (padded)."""
    K, h4, w4 = logits.shape
    prob = torch.empty((K + 1, H, W), dtype=torch.float32, device=logits.device)
    padded = torch.empty((K + 1, 4 * h4, 4 * w4), dtype=torch.float32, device=logits.device) if want_padded else None
    check(load().xmem_logits_to_prob(ptr(logits), ptr(prob), ptr(padded), K, h4, w4, H, W, lh, lw, stream_ptr()))
    return prob, padded


def aggregate_masks(masks):
    K, H, W = masks.shape
    if not masks.is_contiguous():
        masks = masks.contiguous()
    prob = torch.empty((K + 1, H, W), dtype=torch.float32, device=masks.device)
    check(load().xmem_aggregate_masks(ptr(_req(masks, 'masks')), ptr(prob), K, H, W, stream_ptr()))
    return prob


def merge_masks(pred_no_bg, mask, valid_bits):
    K, H, W = mask.shape
    out = torch.empty_like(mask)
    check(load().xmem_merge_masks(ptr(pred_no_bg), ptr(mask), valid_bits, ptr(out), K, H, W, stream_ptr()))
    return out


def resize_bilinear(prob, shape):
    Cc, Hi, Wi = prob.shape
    if not prob.is_contiguous():
        prob = prob.contiguous()
    out = torch.empty((Cc, int(shape[0]), int(shape[1])), dtype=torch.float32, device=prob.device)
    check(load().xmem_resize_bilinear(ptr(prob), ptr(out), Cc, Hi, Wi, int(shape[0]), int(shape[1]), stream_ptr()))
    return out


def argmax_u8(prob):
    Cc, H, W = prob.shape
    if not prob.is_contiguous():
        prob = prob.contiguous()
    out = torch.empty((H, W), dtype=torch.uint8, device=prob.device)
    check(load().xmem_argmax_u8(ptr(_req(prob, 'prob')), ptr(out), Cc, H, W, stream_ptr()))
    return out


def nhwc_to_nchw(x, c=None, off=0):
    """x [B,H,W,ld] (channels off..off+c) -> contiguous [B,c,H,W]."""
    B, H, W, ld = x.shape
    c = c if c is not None else ld
    out = torch.empty((B, c, H, W), dtype=torch.float32, device=x.device)
    check(load().xmem_nhwc_to_nchw(C.c_void_p(x.data_ptr() + 4 * off), ld, ptr(out), B, H * W, c, stream_ptr()))
    return out


def nchw_to_nhwc(x):
    B, c, H, W = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    out = torch.empty((B, H, W, c), dtype=torch.float32, device=x.device)
    check(load().xmem_nchw_to_nhwc(ptr(_req(x, 'tensor')), ptr(out), c, B, H * W, c, stream_ptr()))
    return out


# ---------------------------------------------------------------------------------------------
# memory readout
# ---------------------------------------------------------------------------------------------

_AFF_STATS = bool(os.environ.get('XMEM_AFFINITY_STATS'))
_SPLIT_CHECK = bool(os.environ.get('XMEM_SPLIT_CHECK'))
_GUARD = bool(os.environ.get('XMEM_GUARD'))


ROWS16_FLOATS = 72            # one fp16 filter operand row (144 halfs = 288 bytes) counted in floats


def affinity_rows16(key, shrinkage, out):
    """Filter operand rows of the memory elements `key` [n,Ck] / `shrinkage` [n] | None into out [n, 72] (float32-typed storage
    of 144 halfs per row).  Stores call this when elements are added or replaced (kv_memory_store.py) and hand the rows to
    affinity_topk, so the per-frame rows kernel and its N x 288 bytes of writes are gone."""
    n = key.shape[0]
    if n == 0:
        return out
    if out.shape[0] != n or out.shape[1] != ROWS16_FLOATS or not out.is_contiguous() or not key.is_contiguous():
        raise RuntimeError('affinity_rows16: expected contiguous key [n, Ck] and out [n, 72]')
    check(load().xmem_affinity_rows16(ptr(_req(key, 'key')), ptr(shrinkage), n, ptr(out), stream_ptr()))
    return out


def affinity_topk(segments, qk, qe, top_k, want_sim=False, hint=None):
    """segments: list of (key [n,Ck], shrinkage [n] | None[, rows16 [n,72] | None]).  Returns w [HW,k], idx [HW,k] (int32), sim | None.
    hint: None or (idx [HW,k'] int32 of an earlier call on the same list of stores, its segment sizes, grid width) - only
    tightens the internal lower bound of the k-th similarity (xmem_affinity_topk_hinted); results do not depend on it."""
    lib = load()
    HW, ck = qk.shape
    segs = [(sg[0], sg[1], sg[2] if len(sg) > 2 else None) if (sg[0] is not None and sg[0].shape[0] > 0) else (None, None, None)
            for sg in segments]                                            # empty stores keep their slot
    n_total = sum(k.shape[0] for k, _, _ in segs if k is not None)
    if n_total < top_k:
        raise RuntimeError(f'selected index k out of range: top_k={top_k} > {n_total} memory elements')
    arr = (KeySegment * max(len(segs), 1))()
    for i, (k, s, r16) in enumerate(segs):
        arr[i].key = k.data_ptr() if k is not None else None
        arr[i].shrinkage = s.data_ptr() if s is not None else None
        arr[i].n = k.shape[0] if k is not None else 0
        arr[i].rows16 = r16.data_ptr() if (r16 is not None and k is not None and r16.shape[0] == k.shape[0]) else None
    w = torch.empty((HW, top_k), dtype=torch.float32, device=qk.device)
    idx = torch.empty((HW, top_k), dtype=torch.int32, device=qk.device)
    sim = torch.empty((HW, top_k), dtype=torch.float32, device=qk.device) if want_sim else None
    need = lib.xmem_affinity_topk_workspace_bytes(n_total, HW, top_k)
    ws = workspace(need, qk.device, 'affinity')
    h, hp = None, None
    if hint is not None:
        h_idx, h_sizes, h_gw = hint
        if h_idx is not None and h_idx.is_cuda and tuple(h_idx.shape[:1]) == (HW,) and h_idx.dtype == torch.int32 \
                and len(h_sizes) == len(segs) <= 4:
            h = AffinityHint()
            h.idx = h_idx.data_ptr(); h.top_k = h_idx.shape[1]; h.n_seg = len(h_sizes)
            for i, n in enumerate(h_sizes):
                h.seg_n[i] = int(n)
            h.grid_w = int(h_gw or 0)
            hp = C.byref(h)
    e0 = _tap_begin()
    check(lib.xmem_affinity_topk_hinted(arr, len(segs), ptr(qk), ptr(qe), ck, HW, top_k, hp, ptr(w), ptr(idx), ptr(sim),
                                        ptr(ws), need, stream_ptr()))
    _tap_end('affinity', e0, 4.0 * ck * n_total * HW)
    if _AFF_STATS and hp is not None:                      # tools: candidate statistics of the fp16-filter path (synchronises)
        o = [C.c_size_t(), C.c_size_t(), C.c_size_t()]
        check(lib.xmem_affinity_debug_offsets(n_total, HW, *[C.byref(x) for x in o]))
        torch.cuda.synchronize()
        cnt = ws[o[0].value:o[0].value + 4 * HW].view(torch.int32).float()
        flg = ws[o[1].value:o[1].value + 4 * ((HW + 127) // 128)].view(torch.int32)
        print(f'[affinity] N={n_total} HW={HW}: candidates/query mean {float(cnt.mean()):.0f} median {float(cnt.median()):.0f} '
              f'max {int(cnt.max())}; fallback tiles {int((flg != 0).sum())}/{flg.numel()}', file=sys.stderr)
    if RECORD is not None:
        RECORD.append(('affinity', f'{n_total}x{HW}k{top_k}', 4.0 * ck * n_total * HW,
                       lambda: lib.xmem_affinity_topk_hinted(arr, len(segs), ptr(qk), ptr(qe), ck, HW, top_k, hp, ptr(w), ptr(idx),
                                                             ptr(sim), ptr(ws), need, stream_ptr()), (segs, qk, qe, w, idx, sim, ws, h, hint)))
    return w, idx, sim


def usage_update(w, idx, first, count, use_count, life_count):
    if count <= 0:
        return
    HW, k = w.shape
    fx = workspace(8 * count, w.device, 'usage_fx')
    check(load().xmem_usage_update(ptr(w), ptr(idx), HW, k, first, count, ptr(use_count), ptr(life_count), ptr(fx), stream_ptr()))


def readout_sparse(value_segments, w, idx, cv, out, out_ld, obj_stride, out_off=0):
    """value_segments[obj][seg] = tensor [n_seg, Cv] (n may be 0).  Writes out[obj][q][out_off:out_off+Cv]."""
    n_obj = len(value_segments)
    n_seg = len(value_segments[0])
    arr = (ValueSegment * (n_obj * n_seg))()
    for o, segs in enumerate(value_segments):
        for s, v in enumerate(segs):
            arr[o * n_seg + s].value = v.data_ptr() if v is not None and v.shape[0] > 0 else None
            arr[o * n_seg + s].n = v.shape[0] if v is not None else 0
    HW, k = w.shape
    e0 = _tap_begin()
    check(load().xmem_readout_sparse_t(arr, n_obj, n_seg, ptr(w), ptr(idx), HW, k, cv,
                                       C.c_void_p(out.data_ptr() + out.element_size() * out_off), _h(out), out_ld, obj_stride, stream_ptr()))
    _tap_end('readout', e0, 2.0 * cv * k * HW * n_obj)


def similarity_dense(key, shrinkage, qk, qe):
    """key [n,Ck], qk/qe [P,Ck] -> sim [P,n]."""
    n, ck = key.shape
    P = qk.shape[0]
    out = torch.empty((P, n), dtype=torch.float32, device=key.device)
    check(load().xmem_similarity_dense(ptr(key), ptr(shrinkage), n, ptr(qk), ptr(qe), P, ck, ptr(out), stream_ptr()))
    return out


def selector_prepare(key_rows, sel_rows, mask, h, w, alpha, eps, Mexp, Qexp, bsq, presence):
    """One frame of the candidate selector (frame_selection.py:156-186): key_rows/sel_rows [HW,Ck], mask [C,H,W] | None,
    outputs are views into the per-video operand arrays (Mexp/Qexp [HW,2Ck], bsq [HW], presence int32[1])."""
    hw, ck = key_rows.shape
    if hw != h * w:
        raise ValueError('key rows do not match h*w')
    if mask is not None:
        Cm, H, W = mask.shape
        mask = _req(mask, 'mask')
    else:
        Cm = H = W = 0
    check(load().xmem_selector_prepare(ptr(_req(key_rows, 'key')), ptr(_req(sel_rows, 'selection')), ptr(mask), Cm, H, W,
                                       h, w, ck, float(alpha), float(1 - alpha), float(eps),
                                       ptr(Mexp), ptr(Qexp), ptr(bsq), ptr(presence), stream_ptr()))


def cycle_dissimilarity(Mexp, Qexp, bsq, shrinkage, chosen, valid=None):
    """Score every frame against frame `chosen` (frame_selection.py:218-226).  Mexp/Qexp [F,HW,2Ck]; bsq/shrinkage [F,HW];
    valid uint8 [F] | None.  Returns float64 [F]."""
    F, HW, k2 = Mexp.shape
    out = torch.empty((F,), dtype=torch.float64, device=Mexp.device)
    nbytes = load().xmem_cycle_dissimilarity_workspace_bytes(F, HW)
    ws = workspace(nbytes, Mexp.device, 'selector')
    check(load().xmem_cycle_dissimilarity(ptr(_req(Mexp, 'Mexp')), ptr(_req(Qexp, 'Qexp')), ptr(_req(bsq, 'bsq')),
                                          ptr(_req(shrinkage, 'shrinkage')), F, HW, k2 // 2, int(chosen), ptr(valid),
                                          ptr(out), ptr(ws), nbytes, stream_ptr()))
    return out


def usage_ratio(use, life):
    out = torch.empty_like(use)
    check(load().xmem_usage_ratio(ptr(use), ptr(life), ptr(out), use.numel(), stream_ptr()))
    return out


def topk_1d(values, k, largest=True):
    idx = torch.empty((k,), dtype=torch.int32, device=values.device)
    val = torch.empty((k,), dtype=torch.float32, device=values.device)
    check(load().xmem_topk_1d(ptr(values), values.numel(), k, int(largest), ptr(idx), ptr(val), stream_ptr()))
    return val, idx


def gather_rows(src, index):
    """src [n,C] (or [n]) rows at int32 `index` [m] -> [m,C]."""
    cdim = src.shape[1] if src.dim() == 2 else 1
    m = index.numel()
    out = torch.empty((m, cdim) if src.dim() == 2 else (m,), dtype=torch.float32, device=src.device)
    if m > 0:
        check(load().xmem_gather_rows(ptr(src), cdim, ptr(index), m, ptr(out), stream_ptr()))
    return out


def softmax_rows_suffix(sim, count):
    P, n = sim.shape
    check(load().xmem_softmax_rows_suffix(ptr(sim), P, n, count, stream_ptr()))
    return sim


def softmax_rows_topk(sim, k):
    """In place on contiguous rows [P, n]: softmax over each row's k largest entries (no max shift, memory_util.py:45-54),
    zeros elsewhere."""
    P, n = sim.shape
    if not sim.is_contiguous():
        raise RuntimeError('softmax_rows_topk: rows must be contiguous')
    check(load().xmem_softmax_rows_topk(ptr(_req(sim, 'sim')), P, n, k, stream_ptr()))
    return sim


def weighted_rows(aff, count, V):
    """aff [P,n]; V [count,C] -> [P,C] using the last `count` columns of aff."""
    P, n = aff.shape
    cdim = V.shape[1] if V.dim() == 2 else 1
    out = torch.empty((P, cdim), dtype=torch.float32, device=aff.device)
    check(load().xmem_weighted_rows(ptr(aff), P, n, count, ptr(V), cdim, ptr(out), stream_ptr()))
    return out


def select_greater(usage, threshold_dev):
    n = usage.numel()
    idx = torch.empty((n,), dtype=torch.int32, device=usage.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=usage.device)
    check(load().xmem_select_greater(ptr(usage), n, ptr(threshold_dev), ptr(idx), ptr(cnt), stream_ptr()))
    return idx, cnt
