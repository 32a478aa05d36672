#!/usr/bin/env python
"""Benchmark of the per-frame memory-readout path (BASELINE.json: frames/sec at 480p, 1 object, 32 memory frames).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload b32|b32dyn|c3|c4|c5]

Workloads (`--workload`, default b32 = BASELINE configs[1], the one the metric is quoted on):
  b32     480x854, 1 object, 32 annotated frames preloaded with put_to_permanent_memory, mem_every=1e9: N stays 32*1620
  b32dyn  480x854, 1 object, 22 permanent frames + mem_every=10, T_max=10, T_min=5, long-term on: encode_value and
          consolidation inside the timed loop
  c3      480x854, 3 objects, 1 permanent frame, mem_every=5 (DAVIS-like multi-object stream, consolidation)
  c4      720x1280, 1 object, 256 permanent frames (N = 921 600) + long-term on
  c5      1080x1920, 5 objects, 512 permanent frames (N = 4 177 920)
Every timed step is step() (encode_key -> match_memory -> segment [-> encode_value -> add_memory]) -> resize/argmax ->
uint8 mask on the host (run_on_video.py:106-113 timing; the device->host copy of frame t is awaited after frame t+1 is
enqueued, every mask is on the host before the clock stops).

N > 1: one process per GPU, independent replica streams, no data-path collective; `--gpus N` without a torchrun
environment spawns the N ranks itself (xmem2_amd.launch.spawn_ranks) and refuses to run on fewer devices.  Rank 0 prints
ONE JSON line.  Kernel-level numbers come from (i) HIP events on the launch stream around the eager memory-readout calls
inside a second, instrumented pass of the timed schedule and (ii) a rocprofv3 --kernel-trace of this same command run as
a child process and cut to the timed region with marker kernels (N=1, rank 0).

Frame pipeline of `value` (optional for a caller, changes no result): the key encoder of the coming frames in batches of --key-batch on
a side stream (`prefetch_keys`), the memory readout of the next prefetched frame on a third stream under the current frame's decoder
(early readout, the default since round 6: DESIGN.md 4.7; `config.early_readout` says whether a line used it; XMEM_EARLY_READOUT=0 turns
it off), decoder + mask output on the main stream.  The timed region is self-contained: nothing of its frames is computed before the
clock starts (the warm-up's pending hints are dropped, the region hints its own first key batch) and every mask is on the host before it stops.
`value_no_prefetch` is the same workload through `step()` alone - the reference's call sequence.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from xmem2_amd.launch import shard_videos, spawn_ranks          # noqa: E402,F401  (shard_videos re-exported for callers)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0          # dense v_mfma_f32_32x32x16_f16 peak (~2.5 PF)
# What a register-only loop of independent MFMA chains sustains chip-wide on RANDOM operands (the chip clocks to its power budget:
# MI355X_MICROARCH.md "DVFS give-back"); measured by tools/probes/mfma_shadow/mfma_sustained.hip on the round-4 box, output in
# profiles/r04_mfma_probes.txt (fp16: 1494-1672, fp32: 130-145 TFLOP/s; 0.90-0.95 / 0.99 of nominal on all-zero operands).
# Reported beside the nominal peaks as the ceiling a kernel built on that instruction has on real data; `frac` stays nominal.
SUSTAINED_F16_MFMA_TFLOPS = 1600.0
SUSTAINED_FP32_MFMA_TFLOPS = 140.0
F16_K = 144                            # contraction length of the fp16 filter (128 terms + 16 augmentation terms)
CK, CV, TOPK = 64, 512, 30
CLEAR_MARGIN = 2e-2                    # the tests' gate (tests/test_gpu_e2e.py): 2x the reference's own 8-vs-1-thread probability noise
TIGHT_MARGIN = 2e-3                    # SURVEY 8(c)'s acceptance margin, reported beside it
SCHEMA = 6                             # meaning of the keys of the JSON line; see `schema_note` in the line
LONG_WINDOW = 200                      # frames of `value_long_window` (reported when --steps is shorter)
BASELINE_METRIC = 'frames/sec at 480p, 1 obj, 32 memory frames; mask IoU vs reference'

PRECISION_LABEL = {'fp32': '',
                   'fp16': ' [FP16 LOOP (half activations + half-operand convolutions, fp32 accumulate; the reference\'s autocast mode): not the headline metric]',
                   'fp16w': ' [EXPERIMENT fp16w (fp16 Winograd-F(2x2) operands only): not the headline metric]',
                   'fp32x': ' [SPLIT-OPERAND EXPERIMENT fp32x: separately reported, not the headline metric]'}
PRECISION_DTYPE = {'fp32': 'f32',
                   'fp16': 'f16 activations in HBM and f16 conv operands (direct implicit GEMM on v_mfma_f32_32x32x16_f16), f32 accumulate and epilogue; '
                           'stems, keys, memory, readout weights, GRU state, logits, probabilities f32; permanent-memory preload f32',
                   'fp16w': 'f16 Winograd-domain conv operands, f32 accumulate; everything else f32',
                   'fp32x': 'f32 carried as f16 pairs (hi + lo) in the conv GEMMs: four partial products on the fp16 MFMA, f32 accumulate; '
                            'tensors, memory readout and everything else f32'}

WORKLOADS = {
    'b32': dict(H=480, W=854, K=1, perm=32, mem_every=10 ** 9, count_usage=False, n_query=32,
                desc='B32: synthetic 480x854 clip, 1 object, 32 permanent memory frames (N=51840), mem_every=1e9'),
    'b32dyn': dict(H=480, W=854, K=1, perm=22, mem_every=10, count_usage=True, n_query=32,
                   desc='B32-dyn: 480x854, 1 object, 22 permanent frames + mem_every=10 (T_max=10, T_min=5, long-term on): '
                        'encode_value + consolidation inside the timed loop'),
    'b32motion': dict(H=480, W=854, K=1, perm=32, mem_every=10 ** 9, count_usage=False, n_query=32, motion=6, cut_every=16,
                      desc='B32-motion: as B32, but the query frames move 6 px per frame and every 16th frame is a hard scene cut '
                           '(mirrored, shifted scene): loose hint bounds, overflowing candidate lists, second filter pass'),
    'c3': dict(H=480, W=854, K=3, perm=1, mem_every=5, count_usage=True, n_query=32, conditioning='multi_object',
               desc='C3 stream: 480x854, 3 objects, 1 permanent frame, mem_every=5, long-term consolidation; multi-object conditioning of the '
                    'synthetic checkpoint (xmem2_amd.synth: without it 45 % of the pixels are ties between objects)'),
    'c4': dict(H=720, W=1280, K=1, perm=256, mem_every=10, count_usage=True, n_query=16,
               desc='C4: 720x1280, 1 object, 256 permanent frames (N=921600) + long-term on, mem_every=10'),
    'c5': dict(H=1080, W=1920, K=5, perm=512, mem_every=10 ** 9, count_usage=False, n_query=16,
               desc='C5: 1080x1920 (pads to 1088), 5 objects, 512 permanent frames (N=4177920)'),
}


# ---- multi-rank helpers (covered by tests/test_multi_gpu.py with gloo) ------------------------------------
def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _reduce(value, device, op):
    if not _active():
        return value
    if dist.get_backend() != 'nccl':
        device = torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(value, device):
    return _reduce(value, device, dist.ReduceOp.MAX)


def sum_over_ranks(value, device):
    v = _reduce(value, device, dist.ReduceOp.SUM)
    return int(round(v)) if isinstance(value, int) else v


def gather_over_ranks(value, device):
    """[value of rank 0, ..., value of rank N-1] on every rank."""
    if not _active():
        return [float(value)]
    if dist.get_backend() != 'nccl':
        device = torch.device('cpu')
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def barrier(device):
    if _active():
        dist.barrier(device_ids=[device.index] if (device.type == 'cuda' and dist.get_backend() == 'nccl') else None)


# ---- workload ------------------------------------------------------------------------------------------------
def workload_config(wl):
    return dict(mem_every=wl['mem_every'], deep_update_every=-1, enable_long_term=True,
                enable_long_term_count_usage=wl['count_usage'], hidden_dim=64, key_dim=CK, value_dim=CV, top_k=TOPK,
                max_mid_term_frames=10, min_mid_term_frames=5, num_prototypes=128, max_long_term_elements=10000)


def b32_config():
    return workload_config(WORKLOADS['b32'])


def padded(x):
    return (x + 15) // 16 * 16


def algorithmic_gflop_per_frame(wl, n_mem_elements=None):
    """F_key + F_dec(K) conv FLOPs (+ F_val on memory frames) + similarity; the readout runs sparse."""
    hp, wp, K = padded(wl['H']), padded(wl['W']), wl['K']
    hw = (hp // 16) * (wp // 16)
    n = n_mem_elements if n_mem_elements is not None else wl['perm'] * hw
    f_key = 139944 * hp * wp
    f_dec = (147456 + 416779 * K) * hp * wp
    f_val = 214281 * K * hp * wp / wl['mem_every'] if wl['mem_every'] < 10 ** 8 else 0.0
    f_sim = 4 * CK * n * hw
    f_ro = 2 * CV * TOPK * hw * K
    return dict(key=f_key / 1e9, decoder=f_dec / 1e9, value_amortised=f_val / 1e9, similarity=f_sim / 1e9,
                readout_sparse=f_ro / 1e9, conv=(f_key + f_dec + f_val) / 1e9,
                total=(f_key + f_dec + f_val + f_sim + f_ro) / 1e9, memory_elements=n, hw=hw)


def make_clip(wl):
    """(frames [T,3,H,W], masks [T,K,H,W], perm(j) -> (frame, mask), n_query).  More than 32 permanent frames are 8 base
    frames shifted by distinct offsets (distinct keys without generating hundreds of 1080p fields on the host)."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    H, W, K, P, nq = wl['H'], wl['W'], wl['K'], wl['perm'], wl['n_query']
    base = P if P <= 32 else 8
    frames = synthetic_frames(base + nq, H, W)
    masks = synthetic_masks(base + nq, K, H, W)
    return frames, masks, base, nq


def run_gpu(args, device, rank, world):
    from xmem2_amd import InferenceCore, XMem, ops
    from xmem2_amd.run_on_video import AsyncMaskFetcher
    from xmem2_amd.synth import synthetic_state_dict
    wl = WORKLOADS[args.workload]
    cfg = workload_config(wl)
    sd = synthetic_state_dict(0, conditioning=wl.get('conditioning'))
    net = XMem(dict(cfg, precision=args.precision), None).to(device).eval()
    net.load_weights(sd)
    frames, masks, base, n_query = make_clip(wl)
    fr = torch.from_numpy(frames).to(device)
    mk = torch.from_numpy(masks).to(device)
    core = InferenceCore(net, cfg)
    core.set_all_labels(list(range(1, wl['K'] + 1)))
    t0 = time.perf_counter()
    for j in range(wl['perm']):
        if wl['perm'] <= 32:
            core.put_to_permanent_memory(fr[j], mk[j])
        else:
            sh = (3 * (j // base), 5 * (j // base))
            core.put_to_permanent_memory(torch.roll(fr[j % base], sh, (1, 2)), torch.roll(mk[j % base], sh, (1, 2)), ti=j)
    torch.cuda.synchronize(device)
    preload_s = time.perf_counter() - t0
    hw = (padded(wl['H']) // 16) * (padded(wl['W']) // 16)
    assert core.memory.permanent_work_mem.size == wl['perm'] * hw

    fetcher = AsyncMaskFetcher()                      # uint8 masks reach the host one frame behind the GPU (as run_on_video)
    KB = max(1, args.key_batch)
    if wl.get('motion'):
        # realistic motion for the readout's hint: frame i is the clip frame moved by `motion` px per frame (circular), and
        # every `cut_every` frames the scene changes (mirrored + shifted) - the hint of the frame before a cut is useless
        mv, ce = wl['motion'], wl['cut_every']
        cache = {}

        def frame(i):
            j = i % (4 * n_query)
            if j not in cache:
                f = torch.roll(fr[base + (j % n_query)], shifts=mv * j, dims=2)
                if (j // ce) % 2 == 1:
                    f = torch.roll(torch.flip(f, dims=(1,)), shifts=137, dims=2)
                cache[j] = f.contiguous()
            return cache[j]
    else:
        frame = lambda i: fr[base + (i % n_query)]

    def hint(first):                                 # batched key encoder of frames [first, first+KB) on the side stream
        if not args.no_prefetch:
            # the default, safe form (the side stream waits for this stream's tail before it reads device inputs): measured free,
            # 611-612 frames/s either way (profiles/r06_early_readout_root_cause.txt); XMEM_BENCH_UNSAFE_HINTS=1 is the A/B knob
            core.prefetch_keys([frame(first + j) for j in range(KB)],
                               inputs_complete=bool(os.environ.get('XMEM_BENCH_UNSAFE_HINTS')) and not wl.get('motion'))

    host_t = {'step': 0.0, 'hint': 0.0, 'fetch': 0.0, 'n': 0} if os.environ.get('XMEM_BENCH_HOST_TIMES') else None

    def one_step(i, stop=None, origin=0):
        # `stop`: one past the last frame of the stream being driven; `origin`: its first frame (the phase of the key batches).  A finite
        # stream does not hint a batch that starts at or past its end (run_on_video does not either); a batch that straddles the end is
        # hinted whole (no new graph variant inside a timed region - the frames past the end are extra work INSIDE it).
        if host_t is None:
            prob = core.step(frame(i), None, None)
            if (i - origin) % KB == 0 and (stop is None or i + KB < stop):   # first frame of its batch consumed: hint the next batch under it
                hint(i + KB)
            return [m for _, m in fetcher.submit(i, ops.argmax_u8(prob))]
        t0 = time.perf_counter()
        prob = core.step(frame(i), None, None)
        t1 = time.perf_counter()
        if (i - origin) % KB == 0 and (stop is None or i + KB < stop):
            hint(i + KB)
        t2 = time.perf_counter()
        am = ops.argmax_u8(prob)
        t2b = time.perf_counter()
        out = [m for _, m in fetcher.submit(i, am)]
        t3 = time.perf_counter()
        host_t['step'] += t1 - t0; host_t['hint'] += t2 - t1; host_t['fetch'] += t3 - t2; host_t['n'] += 1
        host_t.setdefault('by_phase', {}).setdefault((i - origin) % KB, []).append((t1 - t0, t2 - t1, t2b - t2, t3 - t2b))
        return out

    # setup (untimed, like the preload): enough frames to capture every HIP graph variant the stream will replay
    # (two key batches; for dynamic memories two memory frames -> value-encoder graphs, deep-update variants)
    me = wl['mem_every'] if wl['mem_every'] < 10 ** 8 else 0
    setup = (2 * KB + 2 * me + KB - 1) // KB * KB
    hint(-setup)
    for i in range(-setup, 0):
        one_step(i)
    for i in range(args.warmup):
        one_step(i)
    # Nothing of the timed frames is computed before the clock starts (schema 6): the warm-up's hints reach into them (its last batches were
    # hinted under its last frames, and the last warm-up step enqueued the first timed frame's memory readout ahead) - all of that is dropped
    # here, and the timed region starts like a stream of its own: its first key batch is hinted INSIDE it, un-overlapped, its first frame
    # reads the memory inside its own step().  Until round 6 the first timed batch's key pass ran during the warm-up: with the closing
    # hint past the end removed (round 6, step 12) the region then held one key pass (of K / key_batch) less than its K frames need.
    # (The dropped readout leaves only its top-k as the next call's HINT, which changes no result - xmem_affinity_topk_hinted.)
    core.cancel_prefetch()
    fetcher.drain()
    # ---- timed region: exactly `steps` frames, barrier + device sync on both sides --------------------------
    barrier(device); torch.cuda.synchronize(device)
    if args.traced_child:
        ops.trace_marker(1)
    t0 = time.perf_counter()
    out_masks = []
    first = args.warmup
    hint(first)
    for i in range(args.steps):
        out_masks += one_step(first + i, stop=first + args.steps, origin=first)
    out_masks += [m for _, m in fetcher.drain()]      # every mask of the timed steps is on the host before the clock stops
    assert len(out_masks) == args.steps
    if args.traced_child:
        ops.trace_marker(2)
    torch.cuda.synchronize(device); barrier(device)
    elapsed = time.perf_counter() - t0
    if host_t is not None:
        n = max(host_t['n'], 1)
        print(f"[host ms per frame over {n} calls incl. setup] step {1e3 * host_t['step'] / n:.3f}  hint {1e3 * host_t['hint'] / n:.3f} "
              f"(per batch {1e3 * host_t['hint'] / n * KB:.3f})  argmax+fetch {1e3 * host_t['fetch'] / n:.3f}", file=sys.stderr)
        for ph, v in sorted(host_t.get('by_phase', {}).items()):
            v = v[len(v) // 2:]                       # steady state
            med = [sorted(x[j] for x in v)[len(v) // 2] * 1e3 for j in range(4)]
            print(f"   frame i%KB={ph}: median host ms  step {med[0]:.3f}  hint {med[1]:.3f}  argmax launch {med[2]:.3f}  submit/wait {med[3]:.3f}", file=sys.stderr)
    m = core.memory
    n_elems = m.temporary_work_mem.size + m.permanent_work_mem.size + m.long_mem.size
    # ---- the same measurement over a LONG window (N = 1, when `steps` is short): a self-contained region starts with its first key batch
    # un-overlapped (3.3 ms on an idle chip before the first decoder can run: 10 % of a 20-frame region, 1 % of a 200-frame one); the
    # reference's metric is the rate of a whole video (run_on_video.py:106-113,143), so the line also carries it at LONG_WINDOW frames -
    # measured exactly like `value` (nothing pending before, own first hint inside, every mask on the host before the clock stops)
    long_window = None
    if rank == 0 and world == 1 and not args.traced_child and not args.scale_only and args.steps < LONG_WINDOW \
            and not os.environ.get('XMEM_BENCH_SKIP_PASSES'):
        first = args.warmup + args.steps
        core.cancel_prefetch(); fetcher.drain()
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        hint(first)
        got = 0
        for i in range(LONG_WINDOW):
            got += len(one_step(first + i, stop=first + LONG_WINDOW, origin=first))
        got += len(fetcher.drain())
        torch.cuda.synchronize(device)
        long_window = dict(value=LONG_WINDOW / (time.perf_counter() - t1), steps=LONG_WINDOW)
        assert got == LONG_WINDOW
    # ---- instrumented pass (rank 0): the SAME schedule again (graphs, two streams, batched hints) with HIP events on
    # the launch stream around the memory-readout calls, which are eager launches between the captured stages
    taps, inst_frames, inst_elapsed = {}, 0, None
    skip_extra = bool(os.environ.get('XMEM_BENCH_SKIP_PASSES'))            # debugging aid: none of the instrumented passes
    if rank == 0 and not args.traced_child and not args.scale_only and not skip_extra:
        inst_frames = min(args.steps, 100)
        start = args.warmup + args.steps + (LONG_WINDOW if long_window else 0)
        core.cancel_prefetch()                                        # (frames a straddling last batch hinted past the end of the timed stream)
        hint(start)                                                   # the timed stream ended without a hint past its end: resume them
        ops.EVENT_TAP = []
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for i in range(inst_frames):
            one_step(start + i, origin=start)
        fetcher.drain()
        torch.cuda.synchronize(device)
        inst_elapsed = time.perf_counter() - t1
        events, ops.EVENT_TAP = ops.EVENT_TAP, None
        for kind, e0, e1, flop in events:
            d = taps.setdefault(kind, dict(ms=0.0, flop=0.0, calls=0))
            d['ms'] += e0.elapsed_time(e1); d['flop'] += flop; d['calls'] += 1
            d.setdefault('each_ms', []).append(e0.elapsed_time(e1))
    # ---- the dominant kernel itself (pass-1 fp16 filter) bracketed by HIP events inside the call, and the candidate statistics
    # of the readout (list lengths, query tiles that needed the second pass): a few more frames of the same schedule, with a
    # device sync per frame (not part of any reported rate)
    filt, cand = None, None
    if rank == 0 and not args.traced_child and not args.scale_only and not skip_extra:
        import ctypes as C
        from xmem2_amd._lib import load
        lib = load()
        fe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 24))]
        for a, b in fe:
            a.record(); b.record()                                   # materialises the hipEvent_t handles
        torch.cuda.synchronize(device)
        ms, lens, flagged, tiles = [], [], 0, 0
        start = args.warmup + args.steps + inst_frames
        start = (start + KB - 1) // KB * KB
        core.cancel_prefetch()
        hint(start)
        offs = [C.c_size_t(), C.c_size_t(), C.c_size_t()]
        hw = (padded(wl['H']) // 16) * (padded(wl['W']) // 16)
        for j, (a, b) in enumerate(fe):
            lib.xmem_affinity_profile_events(C.c_void_p(a.cuda_event), C.c_void_p(b.cuda_event))
            one_step(start + j)
            lib.xmem_affinity_profile_events(None, None)
            torch.cuda.synchronize(device)
            try:
                ms.append(a.elapsed_time(b))
            except Exception:
                pass
            mm = core.memory
            n_now = mm.temporary_work_mem.size + mm.permanent_work_mem.size + mm.long_mem.size
            if lib.xmem_affinity_debug_offsets(n_now, hw, *[C.byref(x) for x in offs]) == 0:
                # the scratch the LAST select of this frame ran on: with early readout (the default) that is the readout of the NEXT
                # hinted frame, enqueued on the readout stream under its own scratch scope
                early_scope = f'@early#{core._uid}#' if (getattr(core, 'early_readout', False) and core._early is not None) else ''
                with ops.ws_scope(early_scope):
                    ws = ops.workspace(0, device, 'affinity')
                nt = (hw + 127) // 128
                cnt = ws[offs[0].value:offs[0].value + 4 * hw].view(torch.int32).float()
                fl = ws[offs[1].value:offs[1].value + 4 * nt].view(torch.int32)
                lens.append((float(cnt.mean()), float(cnt.median()), float(cnt.max())))
                flagged += int((fl != 0).sum()); tiles += nt
        fetcher.drain()
        if ms:
            filt = dict(avg_ms=float(np.mean(ms)), median_ms=float(np.median(ms)), launches=len(ms))
        if lens:
            cand = dict(frames=len(lens), candidates_per_query_mean=float(np.mean([x[0] for x in lens])),
                        candidates_per_query_median=float(np.median([x[1] for x in lens])),
                        longest_list=float(max(x[2] for x in lens)),
                        query_tiles_needing_second_pass=flagged / max(tiles, 1),
                        note='final list length per query after the readout (after the second pass where it ran); '
                             'fraction of 128-query tiles whose lists overflowed in pass 1')
    # ---- what the convolutions of one frame EXECUTE on the matrix pipe: KB frames run eagerly with ops.RECORD on (every conv2d call notes
    # its shape, plan, algorithmic FLOPs and the MFMA FLOPs that plan issues: direct form padded to its tile, F(2x2) 16 / F(4x4) 36
    # position GEMMs).  Recording runs everything eagerly and un-hinted (`prefetch_keys` is a pass-through while ops.eager_only()): every
    # recorded frame carries its own key encoder, its select and its decoder - nothing may be pending from the passes above, or the
    # recorded frames would consume keys encoded un-recorded.  Not part of any reported rate
    conv_survey = None
    if rank == 0 and not args.traced_child and not args.scale_only and not skip_extra:
        start = ((args.warmup + args.steps + inst_frames + 400) // KB + 1) * KB
        core.cancel_prefetch()
        ops.RECORD = []
        try:
            for j in range(KB):
                one_step(start + j)
            fetcher.drain(); torch.cuda.synchronize(device)
        finally:
            recs, ops.RECORD = ops.RECORD, None
        forms = {}
        for kind, key, flop, fn, keep in recs:
            if kind != 'conv':
                continue
            info = keep[5]
            t = info['plan'][0]
            form = 'gemv_cout1' if ' ->1/' in key.replace('->', ' ->') else \
                   ('F(4x4)' if 17 <= t <= 28 else ('F(2x2)' if (7 <= t <= 16 or 29 <= t <= 34) else 'direct'))
            f = forms.setdefault(form, dict(launches=0, algorithmic_gflop=0.0, executed_mfma_gflop=0.0))
            f['launches'] += 1; f['algorithmic_gflop'] += flop / 1e9; f['executed_mfma_gflop'] += info['executed_mfma_flops'] / 1e9
        conv_survey = dict(frames=KB, forms={k: {kk: vv / KB for kk, vv in v.items()} for k, v in forms.items()},
                           algorithmic_gflop_per_frame=sum(v['algorithmic_gflop'] for v in forms.values()) / KB,
                           executed_mfma_gflop_per_frame=sum(v['executed_mfma_gflop'] for v in forms.values()) / KB)
    # ---- the select ALONE on the chip, on the stream's own data: the last recorded xmem_affinity_topk_hinted call (this stream's memory,
    # query and previous-frame hint) re-issued on an otherwise idle device, its pass-1 filter bracketed by the library's HIP events.
    # In the timed schedule that call runs on the readout stream under the previous frame's decoder and beside the batched key encoder:
    # its in-stream durations (roofline.achieved / frac, per the contract) carry the co-scheduling; these figures carry only the kernel.
    select_alone = None
    if rank == 0 and not args.traced_child and not args.scale_only and not skip_extra and conv_survey is not None:
        import ctypes as C
        from xmem2_amd._lib import load
        lib = load()
        aff = [r for r in recs if r[0] == 'affinity' and r[4][-1] is not None]
        if aff:
            fn = aff[-1][3]
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for e in evs:
                e.record()
            torch.cuda.synchronize(device)
            tf, tc = [], []
            for it in range(14):
                lib.xmem_affinity_profile_events(C.c_void_p(evs[0].cuda_event), C.c_void_p(evs[1].cuda_event))
                evs[2].record()
                rc = fn()
                evs[3].record()
                torch.cuda.synchronize(device)
                lib.xmem_affinity_profile_events(None, None)
                if rc == 0 and it >= 2:
                    tf.append(evs[0].elapsed_time(evs[1]) * 1e3); tc.append(evs[2].elapsed_time(evs[3]) * 1e3)
            if tf:
                select_alone = dict(filter_kernel_us_median=float(np.median(tf)), filter_kernel_us_min=float(min(tf)),
                                    call_us_median=float(np.median(tc)), calls=len(tf), key=aff[-1][1])
    core.cancel_prefetch()
    # ---- the reference surface's rate: step() on one frame at a time, no prefetch_keys (inference/run_on_video.py:98-113)
    plain = None
    if rank == 0 and not args.traced_child and not args.no_prefetch and args.plain_steps > 0:
        saved, args.no_prefetch = args.no_prefetch, True
        try:
            n_plain = args.plain_steps
            s0 = args.warmup + args.steps + inst_frames + 64
            for i in range(2 * KB + (2 * me if me else 0)):          # captures the un-hinted graph variants
                one_step(s0 + i)
            fetcher.drain(); torch.cuda.synchronize(device)
            t2 = time.perf_counter()
            for i in range(n_plain):
                one_step(s0 + 64 + i)
            fetcher.drain(); torch.cuda.synchronize(device)
            plain = dict(value=n_plain / (time.perf_counter() - t2), steps=n_plain)
        finally:
            args.no_prefetch = saved
    return dict(elapsed=elapsed, preload_s=preload_s, taps=taps, inst_frames=inst_frames, inst_elapsed=inst_elapsed,
                masks=out_masks, core=core, frames=frames, masks_in=masks, sd=sd, n_query=n_query, base=base,
                n_elems=n_elems, wl=wl, cfg=cfg, filter_events=filt, candidates=cand, plain=plain, frame_fn=frame,
                conv_survey=conv_survey, select_alone=select_alone, long_window=long_window)


# ---- rocprofv3 kernel trace of the timed region (child process) ------------------------------------------------
FAMILIES = (('conv', ('conv_', 'wino_', 'wino4_', 'gemm_stream_')),
            ('affinity', ('affinity_',)),
            ('readout', ('readout_sparse_kernel',)),
            ('usage', ('usage_',)),
            ('consolidation', ('topk_1d', 'gather_rows', 'similarity_dense', 'softmax_rows', 'weighted_rows', 'select_greater',
                               'usage_ratio', 'consolid')),
            ('cbam', ('cbam_',)))


def family_of(name):
    n = name[5:] if name.startswith('void ') else name
    n = n.replace('(anonymous namespace)::', '')
    if n.startswith('_Z'):                               # a name the tracer could not demangle (_Float16 arguments)
        n = n[2:].lstrip('0123456789')
    for fam, prefixes in FAMILIES:
        if n.startswith(prefixes):
            return fam
    if n.startswith(('at::', '__amd_rocclr', 'rocclr', 'Cijk_')):
        return 'torch_runtime'
    return 'elementwise'


def parse_kernel_trace(path):
    """Rows of a rocprofv3 kernel-trace CSV between the two `xmem_trace_marker_kernel` launches -> per-kernel and
    per-family table (launches, total ns) + wall ns of the window + union-busy ns."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                         r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].strip()))
    rows.sort()
    marks = [r for r in rows if r[2].startswith('xmem_trace_marker_kernel')]
    if len(marks) < 2:
        raise RuntimeError('trace markers not found in the kernel trace')
    t0, t1 = marks[0][1], marks[-1][0]
    sel = [r for r in rows if t0 <= r[0] < t1 and not r[2].startswith('xmem_trace_marker_kernel')]
    kern, fam, each = {}, {}, {}
    busy, last = 0, t0
    for s, e, name in sel:
        k = kern.setdefault(name, [0, 0]); k[0] += 1; k[1] += e - s
        each.setdefault(name, []).append(e - s)
        g = fam.setdefault(family_of(name), [0, 0]); g[0] += 1; g[1] += e - s
        a = max(s, last)
        if e > a:
            busy += e - a; last = e
    return dict(window_ns=t1 - t0, busy_ns=busy, kernels=kern, families=fam, launches=len(sel),
                median_ns={k: float(np.median(v)) for k, v in each.items()})


def run_traced_child(args):
    """rocprofv3 --kernel-trace around a child copy of this command (fewer steps), cut to its timed region."""
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    steps = min(args.steps, args.trace_steps)
    tmp = tempfile.mkdtemp(prefix='xmem_trace_', dir=os.environ.get('TMPDIR', '/tmp'))
    cmd = [exe, '--kernel-trace', '--output-format', 'csv', '-d', tmp, '--', sys.executable, os.path.abspath(__file__),
           '--steps', str(steps), '--warmup', str(args.warmup), '--workload', args.workload, '--key-batch', str(args.key_batch),
           '--traced-child', '--no-cpu-baseline', '--precision', args.precision]
    if args.no_prefetch:
        cmd.append('--no-prefetch')
    env = dict(os.environ, TMPDIR=os.environ.get('TMPDIR', '/tmp'))
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    try:
        p = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=args.trace_timeout)
        files = glob.glob(os.path.join(tmp, '**', '*kernel_trace.csv'), recursive=True)
        if p.returncode != 0 or not files:
            return None, f'traced child failed (rc {p.returncode}): {p.stderr[-400:]}'
        tr = parse_kernel_trace(max(files, key=os.path.getsize))
        tr['steps'] = steps
        if args.keep_trace:
            os.makedirs(args.keep_trace, exist_ok=True)
            shutil.copy(max(files, key=os.path.getsize), os.path.join(args.keep_trace, f'{args.workload}_kernel_trace.csv'))
        return tr, None
    except Exception as e:                                     # the headline number never depends on the tracer
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_mode_child(args, mode):
    """frames/s of the same workload in an opt-in precision mode (bench.py --precision <mode> as a child process, no CPU baseline /
    trace / plain pass), or {'error': ...}."""
    steps = min(args.steps, 120)
    cmd = [sys.executable, os.path.abspath(__file__), '--precision', mode, '--steps', str(steps), '--warmup', str(args.warmup),
           '--workload', args.workload, '--key-batch', str(args.key_batch), '--no-cpu-baseline', '--no-kernel-trace', '--plain-steps', '0',
           '--no-extra-modes']
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        j = json.loads(p.stdout.strip().splitlines()[-1])
        out = dict(value=j['value'], unit='frames/s', steps=steps, dtype=j['dtype'],
                   note='opt-in mode, separately reported, not the headline: ' + PRECISION_LABEL[mode].strip())
        if j.get('value_long_window'):
            out['value_long_window'] = j['value_long_window']          # the same mode over LONG_WINDOW frames (see the line's own key)
        return out
    except Exception as e:
        return dict(error=f'{type(e).__name__}: {e}')


def committed_pmc(workload, precision='fp32'):
    """HBM-side bytes per frame from the committed PMC passes (profiles/r*_pmc_per_frame*.json), quoted ONLY when they
    were recorded for exactly this build of the kernels (source digest match) - otherwise None, never a stale number."""
    from xmem2_amd.build import source_digest
    dig = source_digest()
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*pmc_per_frame*.json'))):
        try:
            with open(path) as f:
                j = json.load(f)
        except Exception:
            continue
        if j.get('source_digest') == dig and j.get('workload', 'b32') == workload and j.get('precision', 'fp32') == precision:
            best = (path, j)
    if best is None:
        return None
    path, j = best
    fam = {k: v['read_bytes'] + v['write_bytes'] for k, v in j['families'].items()}
    return dict(file=os.path.relpath(path, ROOT), families=fam, mfma_busy=j.get('mfma_busy'))


# ---- CPU baseline -----------------------------------------------------------------------------------------------
def _physical_cores():
    """Physical cores of the host (distinct (package, core id) pairs of /proc/cpuinfo), or None."""
    try:
        seen, phys, core = set(), None, None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


def run_cpu_baseline(res, args, device):
    """The oracle (CPU restatement, bit-equal to the imported reference in the build container) on the same workload on
    this box's host cores.  Protocol (bounded to keep the default run within minutes): thread sweep
    {1, 8, 32, all} with 1 warm-up + 3 timed frames each, then 3 warm-up + `cpu_frames` (>= 20) timed frames at the best
    thread count; medians.  Parity: a FRESH GPU stream (same preload, same frame order) is compared frame by frame."""
    from oracle import cpu_ref as R
    from xmem2_amd import InferenceCore, ops
    wl, cfg = res['wl'], res['cfg']
    ref = R.RefCore(R.RefNet(res['sd']), cfg)
    labels = list(range(1, wl['K'] + 1))
    ref.set_all_labels(labels)
    fr, mk, base = res['frames'], res['masks_in'], res['base']
    pnet = res['core'].network
    if os.environ.get('XMEM_BENCH_PARITY_FRESH_NET'):                      # debugging aid
        from xmem2_amd import XMem
        pnet = XMem(dict(cfg, precision=args.precision), None).to(device).eval()
        pnet.load_weights(res['sd'])
    gpu = InferenceCore(pnet, cfg)
    gpu.set_all_labels(labels)
    all_threads = torch.get_num_threads()
    torch.set_num_threads(min(all_threads, 32))
    for j in range(wl['perm']):
        ref.put_to_permanent_memory(torch.from_numpy(fr[j]), torch.from_numpy(mk[j]))
        gpu.put_to_permanent_memory(torch.from_numpy(fr[j]).to(device), torch.from_numpy(mk[j]).to(device))
    sweep = sorted({t for t in (1, 8, 32, all_threads) if t <= all_threads})
    plan = [(t, 1, 3) for t in sweep]
    n_total = sum(w + k for _, w, k in plan) + 3 + args.cpu_frames
    KB = max(1, args.key_batch)
    frame_fn = res['frame_fn']                                   # the workload's query frames (device tensors), as the timed region
    cpu_frame = lambda i: frame_fn(i).cpu()
    # GPU stream first, driven exactly like the timed region (batched key-encoder hints when enabled)
    dev = [frame_fn(i).clone() for i in range(n_total + 2 * KB)]
    hint = (lambda a: gpu.prefetch_keys(dev[a:a + KB])) if not args.no_prefetch else (lambda a: None)
    hint(0)
    gpu_out = []
    for i in range(n_total):
        pg = gpu.step(dev[i], None, None)
        if i % KB == 0:
            hint(i + KB)
        gpu_out.append((ops.argmax_u8(pg).cpu().numpy(), pg[:, 4::8, 4::8].cpu()))
    gpu.cancel_prefetch()
    if os.environ.get('XMEM_BENCH_PARITY_TRACE'):
        # debugging aid: the same GPU stream again (fresh cores on the same network, preload without CPU work in between) with the early
        # readout on and off, each compared with the stream above
        def again(early):
            g2 = InferenceCore(pnet, cfg); g2.early_readout = early
            g2.set_all_labels(labels)
            for j in range(wl['perm']):
                g2.put_to_permanent_memory(torch.from_numpy(fr[j]).to(device), torch.from_numpy(mk[j]).to(device))
            g2.prefetch_keys(dev[0:KB]) if not args.no_prefetch else None
            o = []
            for i in range(n_total):
                q = g2.step(dev[i], None, None)
                if i % KB == 0 and not args.no_prefetch:
                    g2.prefetch_keys(dev[i + KB:i + 2 * KB])
                o.append(ops.argmax_u8(q).cpu().numpy())
            g2.cancel_prefetch()
            return o
        for early in (True, False):
            o = again(early)
            print(f'[parity] GPU stream again, early_readout={early}: pixels that differ from the first GPU stream per frame: '
                  + ' '.join(str(int((a != b[0]).sum())) for a, b in zip(o, gpu_out)), file=sys.stderr)
    if os.environ.get('XMEM_BENCH_PARITY_TRACE'):                         # debugging aid: were the device-resident inputs touched?
        torch.cuda.synchronize(device)
        host = torch.from_numpy(fr)
        bad = []
        for i in range(min(32, n_total)):
            d = frame_fn(i).cpu()
            h = host[base + (i % res['n_query'])]
            if not torch.equal(d, h):
                bad.append((i, int((d != h).sum())))
        print(f'[parity] device input frames that differ from their host originals after the GPU stream: {bad[:12]}', file=sys.stderr)
        bad = [(i, int((dev[i].cpu() != host[base + (i % res["n_query"])]).sum())) for i in range(len(dev))
               if not torch.equal(dev[i].cpu(), host[base + (i % res['n_query'])])]
        print(f'[parity] cloned input frames that differ: {bad[:12]}', file=sys.stderr)
    del dev
    ious, mism, perr, clear_mism, near_tie, tight_mism = [], 0, 0.0, 0, 0, 0
    cpu_masks, cpu_probs, cpu_margins = [], [], []
    pos = [0]

    def cpu_frames(n):
        ts = []
        for _ in range(n):
            i = pos[0]; pos[0] += 1
            t0 = time.perf_counter()
            p = ref.step(cpu_frame(i), None, None)
            m = R.post_process(p)
            ts.append(time.perf_counter() - t0)
            g, pg = gpu_out[i]
            cpu_masks.append(np.asarray(m))
            cpu_probs.append(p[:, 4::8, 4::8].clone())
            ious.append(R.compute_array_iou(g, m))
            nonlocal mism, perr, clear_mism, near_tie, tight_mism
            mism += int((g != m).sum())
            perr = max(perr, float((pg - p[:, 4::8, 4::8]).abs().max()))
            top2 = torch.topk(p, 2, dim=0).values                       # the CPU path's own top-2 margin per pixel
            margin = (top2[0] - top2[1]).numpy()
            cpu_margins.append(margin.astype(np.float32))
            diff = (g != np.asarray(m))
            clear_mism += int((diff & (margin > CLEAR_MARGIN)).sum())
            tight_mism += int((diff & (margin > TIGHT_MARGIN)).sum())
            near_tie += int((margin < 1e-2).sum())
        return ts

    sweep_fps = {}
    for t, w, k in plan:
        torch.set_num_threads(t)
        cpu_frames(w)
        sweep_fps[t] = 1.0 / float(np.median(cpu_frames(k)))
    best_t = max(sweep_fps, key=sweep_fps.get)
    torch.set_num_threads(best_t)
    cpu_frames(3)
    ts = cpu_frames(args.cpu_frames)
    floor = None
    if wl['K'] == 1:
        # the reference path's own thread-count noise on exactly these frames: the same oracle again at ONE
        # other fixed thread count (8, or 32 when the run above settled on 8) - the figure the GPU-vs-CPU numbers are to be read against
        t2 = 8 if best_t != 8 else min(32, all_threads)
        torch.set_num_threads(t2)
        ref2 = R.RefCore(R.RefNet(res['sd']), cfg)
        ref2.set_all_labels(labels)
        for j in range(wl['perm']):
            ref2.put_to_permanent_memory(torch.from_numpy(fr[j]), torch.from_numpy(mk[j]))
        two, perr2 = [], 0.0
        for i in range(len(cpu_masks)):
            p2 = ref2.step(cpu_frame(i), None, None)
            two.append(np.asarray(R.post_process(p2)))
            perr2 = max(perr2, float((cpu_probs[i] - p2[:, 4::8, 4::8]).abs().max()))
        A, B = np.stack(cpu_masks), np.stack(two)
        M = np.stack(cpu_margins)
        floor = dict(frames=len(two), second_run_threads=t2,
                     argmax_mismatch_pixels_at_clear_cpu_margin=int(((A != B) & (M > CLEAR_MARGIN)).sum()),
                     argmax_mismatch_pixels_at_survey_margin=int(((A != B) & (M > TIGHT_MARGIN)).sum()),
                     first_run_threads=f'sweep {sweep} over the first {sum(w + k for _, w, k in plan)} frames, then {best_t}',
                     argmax_mismatch_pixels=int((A != B).sum()), pixels=int(A.size),
                     mask_iou_min=float(min(R.compute_array_iou(a, b) for a, b in zip(cpu_masks, two))),
                     max_abs_prob_err_ds8=perr2,
                     note='oracle vs oracle at another thread count on the same frames: the reference path\'s own noise; read '
                          'argmax_mismatch_pixels / max_abs_prob_err_ds8 of `parity` against these')
    if wl['K'] > 1:
        # multi-object streams: the reference path's OWN thread-count noise on exactly these frames next
        # to the GPU figures - the same oracle again at 1 thread (the goldens' count) vs the run above (thread sweep, then best_t)
        torch.set_num_threads(1)
        ref1 = R.RefCore(R.RefNet(res['sd']), cfg)
        ref1.set_all_labels(list(range(1, wl['K'] + 1)))
        for j in range(wl['perm']):
            ref1.put_to_permanent_memory(torch.from_numpy(fr[j]), torch.from_numpy(mk[j]))
        one = [np.asarray(R.post_process(ref1.step(cpu_frame(i), None, None))) for i in range(len(cpu_masks))]

        def clip_stats(a, b):
            A, B = np.stack(a), np.stack(b)
            return dict(iou_per_object=[float(((A == c) & (B == c)).sum() / max(((A == c) | (B == c)).sum(), 1)) for c in range(1, wl['K'] + 1)],
                        argmax_mismatch_pixels=int((A != B).sum()), pixels=int(A.size))
        floor = dict(frames=len(one),
                     oracle_sweep_threads_vs_oracle_1_thread=clip_stats(cpu_masks, one),
                     gpu_vs_oracle_1_thread=clip_stats([g for g, _ in gpu_out[:len(one)]], one),
                     gpu_vs_oracle_sweep_threads=clip_stats([g for g, _ in gpu_out[:len(one)]], cpu_masks),
                     note='clip-level IoU per object and argmax mismatch over the same frames; the reference path at two thread counts '
                          'differs from itself by the first entry (tests/test_gpu_e2e.py gates the GPU path against 1.5x that floor)')
    torch.set_num_threads(all_threads)
    if os.environ.get('XMEM_BENCH_PARITY_TRACE'):
        print('[parity] per-frame IoU vs the CPU path: ' + ' '.join(f'{v:.4f}' for v in ious), file=sys.stderr)
    fps = 1.0 / float(np.median(ts))
    return dict(value=fps, unit='frames/s', cores=best_t, threads=best_t, host_physical_cores=_physical_cores(), kind='port',
                cores_note='cores = threads = the torch thread count of the best sweep point (what the contract calls cores: the threads actually used); '
                           'host_physical_cores / host_logical_cpus describe the box',
                one_thread_fps=sweep_fps.get(1), thread_sweep_fps={str(k): v for k, v in sweep_fps.items()},
                host_logical_cpus=os.cpu_count(), frames_timed=len(ts), statistic='median',
                sample=f'{wl["desc"]}: {wl["perm"]} permanent frames preloaded (untimed); thread sweep {sweep} '
                       f'(1 warm-up + 3 timed frames each), then 3 warm-up + {len(ts)} timed frames of step()+argmax at the '
                       f'best count ({best_t} threads); oracle/cpu_ref.py; host has {os.cpu_count()} logical CPUs'), \
        dict(mask_iou_vs_cpu_min=float(min(ious)), mask_iou_vs_cpu_mean=float(np.mean(ious)), argmax_mismatch_pixels=mism,
             frames_compared=len(ious), pixels_per_frame=wl['H'] * wl['W'], max_abs_prob_err_ds8=perr,
             argmax_mismatch_pixels_at_clear_cpu_margin=clear_mism, clear_margin=CLEAR_MARGIN,
             argmax_mismatch_pixels_at_survey_margin=tight_mism, survey_margin=TIGHT_MARGIN,
             cpu_pixels_near_tie_fraction=near_tie / max(len(ious) * wl['H'] * wl['W'], 1), near_tie_margin=1e-2,
             **({'oracle_thread_noise_floor': floor} if floor is not None else {}))


def run_sampled_readout_check(res, device, n_pick=32):
    """C4 / C5: the oracle cannot materialise N x HW (13 GB / 136 GB), so the readout the stream just used - hinted fp16 filter +
    exact refine on the stream's own memory and hint - is checked on a random subset of the queries against the oracle's
    get_similarity + top-k (model/memory_util.py:7-65) over ALL N memory elements."""
    from oracle import cpu_ref as R
    from xmem2_amd import ops
    core, mm = res['core'], res['core'].memory
    stores = [st for st in (mm.long_mem if mm.enable_long_term else None, mm.temporary_work_mem, mm.permanent_work_mem)]
    segs = [((st.key_rows(), st.shrinkage_rows(), st.rows16()) if (st is not None and st.engaged() and st.size > 0) else (None, None, None))
            for st in stores]
    k, _, e = core.encode_frame_key(res['frame_fn'](5))
    h, w = k.shape[-2:]
    qk = k[0].permute(1, 2, 0).reshape(h * w, -1).contiguous()
    qe = e[0].permute(1, 2, 0).reshape(h * w, -1).contiguous()
    hint = mm._aff_hint.get(0)
    if hint is not None and len(hint[1]) != len(segs):
        hint = None
    wgt, idx, sim = ops.affinity_topk(segs, qk, qe, TOPK, want_sim=True, hint=hint)
    torch.cuda.synchronize(device)
    pick = torch.randperm(h * w, generator=torch.Generator().manual_seed(11))[:n_pick]
    mk = torch.cat([sg[0] for sg in segs if sg[0] is not None], 0).cpu()
    ms = torch.cat([sg[1] for sg in segs if sg[0] is not None], 0).cpu()
    ref = R.get_similarity(mk.t().unsqueeze(0), ms.view(1, 1, -1), qk.cpu()[pick].t().unsqueeze(0), qe.cpu()[pick].t().unsqueeze(0))
    rv, ri = torch.topk(ref[0], TOPK, dim=0)
    gv, gi = sim.cpu()[pick], idx.cpu().long()[pick]
    same = (torch.sort(gi, 1)[0] == torch.sort(ri.t(), 1)[0]).all(1).float().mean()
    # where the index sets differ the elements must be (near-)ties of the k-th similarity: the oracle's own value at every index
    # this path picked must reach the oracle's k-th value (synthetic memories of shifted copies hold exact duplicates by the thousand)
    own = torch.gather(ref[0].t(), 1, gi)                                    # [n_pick, k] oracle similarity of OUR indices
    kth = rv[-1].unsqueeze(1)
    tie_ok = ((own >= kth - 2e-5 * kth.abs().clamp(min=1.0)).all(1)).float().mean()
    return dict(kind='sampled readout check (the oracle cannot materialise N x HW at this size)', queries_sampled=int(n_pick),
                memory_elements=int(mk.shape[0]), hinted=hint is not None,
                topk_similarity_max_abs_err=float((gv - rv.t()).abs().max()),
                identical_topk_index_sets=float(same), queries_whose_picks_all_reach_the_oracle_kth_value=float(tie_ok),
                weights_sum_max_err=float((wgt.sum(1) - 1).abs().max()))


# ---- entry -------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='b32', choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-frames', type=int, default=20, help='timed frames of the CPU baseline at its best thread count')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'fp16', 'fp16w', 'fp32x'],
                    help='opt-in modes, each reported under its own metric label, never the headline: fp16 = the fp16 loop (half activations in HBM, '
                         'half-operand direct convolutions on the fp16 MFMA, fp32 accumulate; the reference\'s autocast mode, SURVEY 8f-4); fp16w = only the '
                         'F(2x2) Winograd-domain operands in fp16; fp32x = every fp32 GEMM operand carried as two halfs, four partial products on the fp16 '
                         'MFMA, fp32 accumulation (fp32-class results)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prefetch', action='store_true', help='do not pipeline the coming frames\' key encoder')
    ap.add_argument('--key-batch', type=int, default=4, help='frames per batched key-encoder hint (prefetch_keys)')
    ap.add_argument('--plain-steps', type=int, default=60, help='frames of the extra un-hinted pass that gives value_no_prefetch (0: skip)')
    ap.add_argument('--no-kernel-trace', action='store_true', help='skip the rocprofv3 child run (kernel tables of the timed region)')
    ap.add_argument('--trace-steps', type=int, default=60)
    ap.add_argument('--trace-timeout', type=int, default=420)
    ap.add_argument('--keep-trace', default=None, help='directory to keep the child\'s kernel_trace.csv in')
    ap.add_argument('--traced-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-extra-modes', action='store_true', help='skip the child runs that add value_fp32x / value_fp16_loop to the default line')
    ap.add_argument('--dist-backend', default='gloo', help='control plane of the timing barrier / scalar reductions (the data path has no '
                    'collective).  gloo (default): every rank sees only its own GPU (device isolation as xmem2_amd.launch), reductions on the '
                    'host.  nccl: RCCL control plane, every device stays visible to every rank')
    ap.add_argument('--scale-only', action='store_true', help='only the timed region and the JSON line: no traced child, no opt-in mode children, '
                    'no un-hinted pass, no instrumented passes, no CPU baseline (what a multi-GPU scaling run needs; implied on ranks != 0)')
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if os.environ.get('XMEM_POISON_EMPTY'):
        # debugging aid: torch.empty returns NaN / 0xFF-filled memory instead of whatever the allocator held (activations, arenas, scratch):
        # a read of memory nobody wrote shows up as NaN instead of depending on the allocation history.  Slower; never for a reported rate.
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 or args.gpus > 1:
        args.scale_only = True                     # a scaling run measures the timed region only, on every rank alike
    if args.scale_only:
        args.no_cpu_baseline = args.no_kernel_trace = args.no_extra_modes = True
        args.plain_steps = 0
    torch.set_grad_enabled(False)
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:                          # no torchrun environment: start the N ranks ourselves
            codes = spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
            sys.exit(max(abs(c) for c in codes))
        world, rank, local = 1, 0, 0
    else:
        world = int(os.environ['WORLD_SIZE'])
        rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if world != args.gpus:
            raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}')
    dev_index = local
    if world > 1 and args.dist_backend != 'nccl':
        # Replica streams exchange nothing on the device: every rank sees ONLY its own GPU (as the video launcher's ranks do), so
        # that no rank can allocate on or synchronise with a foreign device; the timing barrier and the scalar reductions run
        # over gloo on the host.  Must happen before the first CUDA call of this process.  (--dist-backend nccl keeps every
        # device visible for an RCCL control plane instead.)
        from xmem2_amd.launch import isolated_device_env
        iso = isolated_device_env(local, os.environ)
        os.environ.update({k: v for k, v in iso.items() if k != 'LOCAL_RANK'})
        dev_index = 0
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    if torch.cuda.device_count() <= dev_index:
        raise SystemExit(f'rank {rank}: device {dev_index} requested but only {torch.cuda.device_count()} visible')
    device = torch.device('cuda', dev_index)
    torch.cuda.set_device(device)
    if world > 1:                                   # one process per GPU: cores next to the GPU, a bounded intra-op thread pool
        from xmem2_amd.launch import pin_rank
        pin = pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)), max_threads=8, device_index=dev_index)
    else:
        pin = None
    backend = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # control plane only (timing barrier + max / sum / gather of scalars): RCCL by default
        backend = args.dist_backend
        if backend == 'nccl':
            try:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe)                       # fail here, on every rank alike, rather than mid-measurement
                torch.cuda.synchronize(device)
            except Exception as e:                           # RCCL unusable on this node: the control plane falls back to gloo
                print(f'[bench] rank {rank}: RCCL control plane failed ({type(e).__name__}: {e}); using gloo', file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                os.environ['MASTER_PORT'] = str(int(os.environ.get('MASTER_PORT', '29500')) + 1)
                backend = 'gloo'
        if backend != 'nccl':
            dist.init_process_group(backend, rank=rank, world_size=world)
    if os.environ.get('XMEM_MAIN_CUS'):            # measurement knob: "first:count" - the main stream on a CU subset (see ops.masked_stream)
        from xmem2_amd import ops as _ops
        first, count = (int(v) for v in os.environ['XMEM_MAIN_CUS'].split(':'))
        torch.cuda.set_stream(_ops.masked_stream(device, count, first))
    res = run_gpu(args, device, rank, world)
    elapsed = max_over_ranks(res['elapsed'], device)
    total_frames = sum_over_ranks(args.steps, device)
    per_rank = [args.steps / e for e in gather_over_ranks(res['elapsed'], device)]
    preload_all = gather_over_ranks(res['preload_s'], device)
    pinned_cpus = gather_over_ranks((pin or {}).get('cpus') or 0, device)
    host_threads = gather_over_ranks(torch.get_num_threads(), device)
    visible = gather_over_ranks(torch.cuda.device_count(), device)
    fps = total_frames / elapsed
    if args.traced_child:
        print(json.dumps({'traced_child': True, 'fps_under_tracer': fps, 'steps': args.steps}), flush=True)
        return
    if rank == 0:
        wl = res['wl']
        alg = algorithmic_gflop_per_frame(wl, res['n_elems'])
        nf = max(res['inst_frames'], 1)
        aff = res['taps'].get('affinity')
        ro = res['taps'].get('readout')
        aff_ms = aff['ms'] / nf if aff else None
        aff_gf = aff['flop'] / nf / 1e9 if aff else alg['similarity']
        aff_tflops = (aff_gf / aff_ms) if aff_ms else None                     # GF / ms = TF/s (whole call, fp32-equivalent)
        fe = res.get('filter_events')
        calls_pf = (aff['calls'] / nf) if aff else 1.0
        gf_per_call = aff_gf / max(calls_pf, 1e-9)
        filt_tflops = (gf_per_call / fe['avg_ms']) if fe else None            # algorithmic F_sim of one call / the filter kernel's own time
        q_pad = (alg['hw'] + 63) // 64 * 64                                     # a wave contracts two 32-query blocks or none
        exec_gf = gf_per_call * (F16_K / 128.0) * (q_pad / alg['hw'])           # what the kernel really contracts: K = 144, queries padded to 64
        line = {
            'metric': (BASELINE_METRIC if args.workload == 'b32' else f'frames/sec ({wl["desc"]})') + PRECISION_LABEL[args.precision],
            'value': fps, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': PRECISION_DTYPE[args.precision],
            'data': 'synthetic',
            'config': {'workload': wl['desc'] + '; step()+argmax+uint8 mask to host per frame, conditioned synthetic weights',
                       'workload_key': args.workload, 'replica_streams': world, 'top_k': TOPK,
                       'frame_pipelining': (not args.no_prefetch), 'key_batch': (args.key_batch if not args.no_prefetch else 0),
                       'early_readout': bool(getattr(res['core'], 'early_readout', False)) and not args.no_prefetch,
                       'parallelism': f'{world} independent streams, no collectives',
                       'control_plane': backend or 'none'},
            'per_rank_fps': per_rank,
            'per_rank_host': {'pinned_cpus': [int(v) for v in pinned_cpus], 'torch_threads': [int(v) for v in host_threads],
                              'visible_devices': [int(v) for v in visible],
                              'note': 'pinned_cpus 0 = not pinned (one rank, or no sched_setaffinity); with the gloo control plane every rank '
                                      'sees exactly one device'},
            'schema': SCHEMA,
            'schema_note': 'schema 5 (round 5): roofline.frac = F_sim / mean duration of the filter kernel in the marker-cut trace / 2.5 PF (as round 4); '
                           'roofline.call_frac = F_sim / the whole select call (all its launches, trace) / 2.5 PF (new); conv_roofline.achieved/frac = '
                           'EXECUTED MFMA FLOPs / conv family time / the peak of the pipe the mode runs on (fp32: 157.3 TF; fp16 / fp32x / fp16w modes: 2.5 PF - '
                           'round 4 divided every mode by 157.3); parity.clear_margin = 2e-2 (round 4: 5e-2) and parity.survey_margin = 2e-3.  Round 6, same schema: '
                           'config.early_readout is true (the schedule is the default again: its wrong stream was a caller-side race, DESIGN.md 4.7); the timed '
                           'stream does not hint frames past its end; cpu_baseline gains threads / host_physical_cores (cores = threads used).  Schema 6 (end of round 6): '
                           'the timed region is self-contained - the warm-up\'s pending key hints and its readout enqueued ahead are dropped before the clock '
                           'starts, the first key batch of the timed frames is hinted (un-overlapped) inside the region, key batches are phased from its first '
                           'frame; before, the first timed batch was encoded during the warm-up, so a region of K frames held K / key_batch - 1 key passes '
                           '(schema-5 lines of round 6 read ~1 % high at 200 steps and ~7 % high at 20)',
            'roofline': {'bound': 'mfma',
                         'kernel': 'affinity_filter16_kernel<false, 4|8> (pass 1 of xmem_affinity_topk_hinted): the N x HW similarity contraction '
                                   'of model/memory_util.py:7-39 on v_mfma_f32_32x32x16_f16 with augmented fp16 operands (the result is a rigorous '
                                   'UPPER estimate; one candidate bit per memory row x query, and since round 6 the kernel turns its bits into the per-query '
                                   'candidate lists itself - no bit matrix, no scan kernel).  Around it in the same call: bound from the previous '
                                   'frame\'s matches, [tighten + second filter pass over query tiles whose lists '
                                   'overflowed], exact fp32 refine of the listed candidates - outputs bit-identical to the fp32 MFMA select',
                         'note': 'achieved = SURVEY 8(d) algorithmic FLOPs of the similarity (F_sim = 4*C_k*N*HW per call) / the average duration of '
                                 'THIS kernel (since round 6 it runs on the readout stream UNDER the previous frame\'s decoder - early readout is the '
                                 'default -, so its in-stream duration includes sharing the chip and waiting for CUs; roofline.alone has the same call re-issued on an idle device); peak = the dense fp16 MFMA peak, the pipe it runs on; frac = executed fraction of that pipe in algorithmic '
                                 'FLOPs (executed_tflops counts the K = 144 operands and the queries padded to 64).  frac_fp32_equivalent is the round-2 '
                                 'yardstick: F_sim / the time of the WHOLE call (all kernels) / the fp32 MFMA peak the contraction ran on before - it '
                                 'exceeds 1 on large memories because the work is not done in fp32 any more',
                         'achieved': filt_tflops, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': (filt_tflops / PEAK_F16_MFMA_TFLOPS) if filt_tflops else None, 'traffic': None, 'source': 'hip_events',
                         'kernel_avg_us': (1e3 * fe['avg_ms']) if fe else None, 'kernel_median_us': (1e3 * fe['median_ms']) if fe else None,
                         'kernel_launches_timed': fe['launches'] if fe else None,
                         'executed_tflops': (exec_gf / fe['avg_ms']) if fe else None,
                         'algorithmic_gflop_per_call': gf_per_call,
                         'frac_fp32_equivalent': (aff_tflops / PEAK_FP32_MFMA_TFLOPS) if aff_tflops else None,
                         'call_tflops_fp32_equivalent': aff_tflops, 'peak_fp32_mfma': PEAK_FP32_MFMA_TFLOPS,
                         'algorithmic_gflop_per_frame': aff_gf, 'ms_per_frame': aff_ms,
                         'call_median_ms': float(np.median(aff['each_ms'])) if aff and aff.get('each_ms') else None,
                         'calls_per_frame': (aff['calls'] / nf) if aff else None,
                         'candidates': res.get('candidates'),
                         'measured': f'kernel: HIP events recorded by the library on the launch stream right before / after the pass-1 filter launch '
                                     f'(xmem_affinity_profile_events) over {fe["launches"] if fe else 0} frames of the timed schedule; whole call: HIP '
                                     f'events around every xmem_affinity_topk_hinted call inside an instrumented pass ({res["inst_frames"]} frames; '
                                     f'HIP graphs, two streams, batch-{args.key_batch} key hints)'},
            'readout': {'ms_per_frame': (ro['ms'] / nf) if ro else None,
                        'algorithmic_gflop_per_frame': (ro['flop'] / nf / 1e9) if ro else None,
                        'bound': 'hbm', 'algorithmic_bytes_per_frame': 4.0 * CV * wl['K'] * TOPK * alg['hw'] + 4.0 * CV * wl['K'] * alg['hw']},
            'frame_gflop': alg, 'whole_frame_tflops': alg['total'] / 1e3 * fps / world,
            'instrumented_pass_fps': (res['inst_frames'] / res['inst_elapsed']) if res['inst_elapsed'] else None,
            'preload_s_per_rank': preload_all,
            'slowest_rank_fps': min(per_rank), 'fastest_rank_fps': max(per_rank),
        }
        if res.get('long_window'):
            line['value_long_window'] = res['long_window']['value']
            line['value_long_window_note'] = (f'frames/s of a {res["long_window"]["steps"]}-frame timed region measured exactly like `value` (self-contained, same '
                                              'process, right after it): a region starts with its first key batch un-overlapped (~3.3 ms before the '
                                              f'first decoder can run), which weighs {args.steps} timed frames far more than a video')
        if res.get('plain'):
            # the reference's caller hands step() one frame at a time (inference/run_on_video.py:98-113): the same workload
            # WITHOUT the prefetch_keys extra - the drop-in rate of the unchanged call sequence
            line['value_no_prefetch'] = res['plain']['value']
            line['value_no_prefetch_note'] = (f'frames/s of {res["plain"]["steps"]} frames through step() alone (no prefetch_keys hints: key encoder, '
                                              'readout and decoder in one stream) after the timed region; `value` uses batched key hints, a streaming '
                                              'extra the reference surface does not have')
        if world == 1 and not args.no_kernel_trace:
            tr, err = run_traced_child(args)
            if tr is None:
                line['kernel_trace'] = {'error': err}
            else:
                st = tr['steps']
                fam = {k: dict(launches_per_frame=v[0] / st, us_per_frame=v[1] / st / 1e3) for k, v in tr['families'].items()}
                top = sorted(tr['kernels'].items(), key=lambda kv: -kv[1][1])[:12]
                line['kernel_trace'] = {
                    'source': f'rocprofv3 --kernel-trace of a child copy of this command ({st} timed frames), cut to the timed region by marker kernels',
                    'frames': st, 'wall_us_per_frame_under_tracer': tr['window_ns'] / st / 1e3,
                    'gpu_busy_union': tr['busy_ns'] / tr['window_ns'], 'launches_per_frame': tr['launches'] / st,
                    'families': fam,
                    'top_kernels': [dict(kernel=k, launches_per_frame=v[0] / st, avg_us=v[1] / v[0] / 1e3, us_per_frame=v[1] / st / 1e3)
                                    for k, v in top]}
                conv_us = fam.get('conv', {}).get('us_per_frame')
                aff_us = fam.get('affinity', {}).get('us_per_frame')
                if conv_us:
                    ctf = alg['conv'] / (conv_us * 1e-3)
                    cs = res.get('conv_survey')
                    etf = (cs['executed_mfma_gflop_per_frame'] / (conv_us * 1e-3)) if cs else None
                    # the pipe the mode's convolutions run on: fp32 MFMA, or the fp16 MFMA (fp32x executes four half products per fp32 product)
                    conv_peak = PEAK_FP32_MFMA_TFLOPS if args.precision == 'fp32' else PEAK_F16_MFMA_TFLOPS
                    if etf and args.precision == 'fp32x':
                        etf *= 4.0
                    line['conv_roofline'] = {'bound': 'mfma', 'kernel': 'xmem_conv2d_nhwc: conv_mfma_kernel / gemm_stream_kernel (Winograd-domain position GEMMs) + transform kernels',
                                             'achieved': etf, 'peak': conv_peak, 'unit': 'TFLOP/s',
                                             'frac': (etf / conv_peak) if etf else None,
                                             'executed_mfma_gflop_per_frame': cs['executed_mfma_gflop_per_frame'] if cs else None,
                                             'executed_by_form': cs['forms'] if cs else None,
                                             'algorithmic_tflops': ctf, 'algorithmic_speed_vs_fp32_peak': ctf / PEAK_FP32_MFMA_TFLOPS,
                                             'algorithmic_gflop_per_frame': alg['conv'],
                                             'algorithmic_gflop_per_frame_surveyed': cs['algorithmic_gflop_per_frame'] if cs else None,
                                             'us_per_frame': conv_us, 'traffic': None,
                                             'note': 'achieved / frac = MFMA FLOPs the plans of one frame EXECUTE (direct form padded to its tiles, F(2x2) 16 and F(4x4) 36 '
                                                     'position GEMMs; surveyed with ops.RECORD over key_batch eagerly run frames, each with its own key encoder) / the conv family\'s kernel time in the '
                                                     'marker-cut trace of the timed region (GEMMs AND transform kernels) / the fp32 MFMA peak.  algorithmic_* = the DIRECT '
                                                     'convolution\'s FLOPs of SURVEY 8(d) over the same time: a speed figure (it exceeds the peak because Winograd executes 1/2.25 '
                                                     'or 1/4 of those FLOPs), not a roofline fraction'}
                if aff_us:
                    line['roofline']['timed_region_trace_us_per_frame'] = aff_us
                    # the SELECT as a whole (every launch of the call: bound, filter + lists, [pass 2], refine) against the pipe its contraction runs on
                    line['roofline']['call_frac'] = (aff_gf / (aff_us * 1e-3)) / PEAK_F16_MFMA_TFLOPS
                    line['roofline']['call_tflops'] = aff_gf / (aff_us * 1e-3)
                    line['roofline']['frac_fp32_equivalent_from_trace'] = (aff_gf / (aff_us * 1e-3)) / PEAK_FP32_MFMA_TFLOPS
                    ks = {}
                    for k, v in tr['kernels'].items():
                        if family_of(k) == 'affinity' and v[0]:
                            ks[k.split('(')[0]] = dict(launches_per_frame=v[0] / st, avg_us=v[1] / v[0] / 1e3, us_per_frame=v[1] / st / 1e3,
                                                       median_us=tr['median_ns'].get(k, 0.0) / 1e3)
                    # the pass-1 filter is its own instantiation (<false, waves>): its executed fp16 FLOPs / its own time, from the trace
                    fk = next((k for k in ks if 'filter16' in k and ('<false' in k or 'ILb0' in k)), None)
                    if fk and ks[fk]['avg_us']:
                        tf = exec_gf / (ks[fk]['avg_us'] * 1e-3)
                        ks[fk].update(executed_gflop_per_launch=exec_gf, executed_tflops=tf, peak_tflops=PEAK_F16_MFMA_TFLOPS,
                                      frac_of_f16_mfma_peak=tf / PEAK_F16_MFMA_TFLOPS,
                                      algorithmic_frac_of_f16_peak=(gf_per_call / (ks[fk]['avg_us'] * 1e-3)) / PEAK_F16_MFMA_TFLOPS)
                        line['roofline']['kernel_avg_us_from_trace'] = ks[fk]['avg_us']
                        # THE roofline figure: this kernel's launches inside the timed region of the traced child (mean per the
                        # contract, median beside it: every key_batch-th launch queues behind the other stream's key encoder)
                        ev = {k: line['roofline'].get(k) for k in ('achieved', 'frac', 'kernel_avg_us', 'kernel_median_us', 'kernel_launches_timed', 'executed_tflops')}
                        mean_us, med_us = ks[fk]['avg_us'], ks[fk]['median_us']
                        line['roofline'].update(
                            achieved=gf_per_call / (mean_us * 1e-3), frac=gf_per_call / (mean_us * 1e-3) / PEAK_F16_MFMA_TFLOPS,
                            achieved_median=gf_per_call / (med_us * 1e-3) if med_us else None,
                            frac_median=(gf_per_call / (med_us * 1e-3) / PEAK_F16_MFMA_TFLOPS) if med_us else None,
                            kernel_avg_us=mean_us, kernel_median_us=med_us, kernel_launches_timed=int(round(ks[fk]['launches_per_frame'] * st)),
                            executed_tflops=exec_gf / (mean_us * 1e-3), source='timed_region_trace',
                            from_hip_events=ev,
                            measured=f'kernel: every launch of {fk} inside the timed region of the rocprofv3 kernel trace of a child copy of this command '
                                     f'({st} frames, cut by marker kernels): mean (achieved / frac) and median (achieved_median / frac_median).  from_hip_events = the '
                                     f'same kernel bracketed by HIP events the library records on the launch stream in an extra pass that synchronises after every '
                                     f'frame (secondary).  Whole call (frac_fp32_equivalent): HIP events around every xmem_affinity_topk_hinted call inside an '
                                     f'instrumented pass ({res["inst_frames"]} frames; HIP graphs, two streams, batch-{args.key_batch} key hints)')
                    line['roofline']['kernels'] = ks
        sa = res.get('select_alone')
        if sa and 'roofline' in line:
            gfc = line['roofline'].get('algorithmic_gflop_per_call') or (4.0 * 64 * res['n_elems'] * res['n_query'] / 1e9)
            sa = dict(sa)
            sa['frac'] = gfc / (sa['filter_kernel_us_median'] * 1e-3) / PEAK_F16_MFMA_TFLOPS
            sa['call_frac'] = gfc / (sa['call_us_median'] * 1e-3) / PEAK_F16_MFMA_TFLOPS
            sa['note'] = ('the SAME call on the stream\'s own data (memory, query, previous-frame hint of a late frame), re-issued alone on an idle device: '
                          'pass-1 filter between the library\'s HIP events (median of 12), F_sim / that / the fp16 MFMA peak = frac; the whole call '
                          '(5 launches) likewise = call_frac.  roofline.achieved / frac above are the in-stream figures the contract asks for: there the '
                          'call runs on the readout stream under the previous frame\'s decoder and beside the batched key encoder (GPU busy 0.96-0.97 '
                          'over three streams), and a kernel\'s duration includes waiting for CUs - the unchanged hint-bound kernel takes 19 us alone '
                          'and 25-77 us in the stream depending on what it is scheduled beside')
            line['roofline']['alone'] = sa
        pmc = committed_pmc(args.workload, args.precision)
        if pmc is not None:
            line['roofline']['traffic'] = pmc['families'].get('affinity')
            line['roofline']['traffic_source'] = pmc['file'] + ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this build; bytes per frame)'
            if pmc.get('mfma_busy'):
                line['roofline']['mfma_busy'] = pmc['mfma_busy'].get('affinity')
            if 'conv_roofline' in line:
                line['conv_roofline']['traffic'] = pmc['families'].get('conv')
                if pmc.get('mfma_busy'):
                    line['conv_roofline']['mfma_busy'] = pmc['mfma_busy'].get('conv')
        if world == 1 and not args.no_cpu_baseline:
            if args.workload in ('c4', 'c5'):
                line['cpu_baseline'] = None
                line['cpu_baseline_note'] = 'not run: the oracle materialises the N x HW affinity (13 GB / 136 GB per frame at this size)'
                line['parity'] = run_sampled_readout_check(res, device)
            else:
                cpu, parity = run_cpu_baseline(res, args, device)
                line['cpu_baseline'] = cpu
                line['parity'] = parity
                line['speedup_vs_cpu'] = fps / cpu['value']
        if world == 1 and args.workload == 'b32' and args.precision == 'fp32' and not args.no_extra_modes and not args.no_prefetch:
            # the opt-in modes on the same workload, each as its own labelled key (never the headline): child copies of this command
            for mode, label in (('fp32x', 'value_fp32x'), ('fp16', 'value_fp16_loop')):
                line[label] = run_mode_child(args, mode)
        for key, sus in (('roofline', SUSTAINED_F16_MFMA_TFLOPS),
                         ('conv_roofline', SUSTAINED_FP32_MFMA_TFLOPS if args.precision == 'fp32' else SUSTAINED_F16_MFMA_TFLOPS)):
            r = line.get(key)
            if r and r.get('achieved'):
                r['peak_sustained_random_operands'] = sus
                r['frac_of_sustained'] = r['achieved'] / sus
                if r.get('achieved_median'):
                    r['frac_of_sustained_median'] = r['achieved_median'] / sus
                r['peak_sustained_source'] = 'tools/probes/mfma_shadow/mfma_sustained.hip -> profiles/r04_mfma_probes.txt'
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
