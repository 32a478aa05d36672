#!/usr/bin/env python
"""Benchmark of the per-frame memory-readout path (BASELINE.json: frames/sec at 480p, 1 object, 32 memory frames).

Workload B32: synthetic 480x854 clip (pads to 480x864, HW=1620), 1 object, 32 annotated frames preloaded
with put_to_permanent_memory, mem_every=1e9 so N stays 32*1620 = 51840; every timed step is
encode_key -> match_memory -> segment -> resize/argmax -> uint8 mask on the host (run_on_video.py:106-113 timing;
the device->host copy of frame t is awaited after frame t+1 is enqueued, every mask is on the host before the clock stops).
One process per GPU; ranks run independent replica streams (no data-path collective); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
H, W, MEM_FRAMES, CK, CV, TOPK = 480, 854, 32, 64, 512, 30


# ---- multi-rank helpers (covered by tests/test_multi_gpu.py with gloo) ------------------------------------
def shard_videos(videos, lengths, rank, world):
    """Longest-first, dealt round-robin: independent per-GPU streams, no exchange step."""
    order = sorted(range(len(videos)), key=lambda i: -lengths[i])
    return [videos[i] for i in order[rank::world]]


def _reduce(value, device, op):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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


def barrier(device):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier(device_ids=[device.index] if (device.type == 'cuda' and dist.get_backend() == 'nccl') else None)


# ---- workload ------------------------------------------------------------------------------------------------
def b32_config():
    return dict(mem_every=10 ** 9, deep_update_every=-1, enable_long_term=True, enable_long_term_count_usage=False,
                hidden_dim=64, key_dim=CK, value_dim=CV, top_k=TOPK, max_mid_term_frames=10, min_mid_term_frames=5,
                num_prototypes=128, max_long_term_elements=10000)


def make_clip(n_query):
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    t = MEM_FRAMES + n_query
    return synthetic_frames(t, H, W), synthetic_masks(t, 1, H, W)


def algorithmic_gflop_per_frame():
    """F_key + F_dec(K=1) conv FLOPs + similarity; the readout runs in its sparse form."""
    hp, wp = 480, 864
    hw = (hp // 16) * (wp // 16)
    n = MEM_FRAMES * hw
    f_key = 139944 * hp * wp
    f_dec = (147456 + 416779) * hp * wp
    f_sim = 4 * CK * n * hw
    f_ro = 2 * CV * TOPK * hw
    return dict(key=f_key / 1e9, decoder=f_dec / 1e9, similarity=f_sim / 1e9, readout_sparse=f_ro / 1e9,
                total=(f_key + f_dec + f_sim + f_ro) / 1e9)


def run_gpu(args, device, rank, world):
    from xmem2_amd import InferenceCore, XMem, ops
    from xmem2_amd.synth import synthetic_state_dict
    cfg = b32_config()
    sd = synthetic_state_dict(0)
    net = XMem(dict(cfg), None).to(device).eval()
    net.load_weights(sd)
    n_query = 32
    frames, masks = make_clip(n_query)
    fr = torch.from_numpy(frames).to(device)
    mk = torch.from_numpy(masks).to(device)
    core = InferenceCore(net, cfg)
    core.set_all_labels([1])
    t0 = time.perf_counter()
    for j in range(MEM_FRAMES):
        core.put_to_permanent_memory(fr[j], mk[j])
    torch.cuda.synchronize(device)
    preload_s = time.perf_counter() - t0
    assert core.memory.permanent_work_mem.size == MEM_FRAMES * 1620

    from xmem2_amd.run_on_video import AsyncMaskFetcher
    fetcher = AsyncMaskFetcher()                      # uint8 masks reach the host one frame behind the GPU (as run_on_video)

    KB = max(1, args.key_batch)
    frame = lambda i: fr[MEM_FRAMES + (i % n_query)]

    def hint(first):                                 # batched key encoder of frames [first, first+KB) on the side stream
        if not args.no_prefetch:
            core.prefetch_keys([frame(first + j) for j in range(KB)])

    def one_step(i):
        prob = core.step(frame(i), None, None)
        if i % KB == 0:                              # first frame of its batch consumed: hint the next batch under it
            hint(i + KB)
        return [m for _, m in fetcher.submit(i, ops.argmax_u8(prob))]

    # setup (untimed, like the preload): two batches of frames capture every HIP graph variant the stream will replay
    hint(-2 * KB)
    for i in range(-2 * KB, 0):
        one_step(i)
    for i in range(args.warmup):
        one_step(i)
    fetcher.drain()
    # ---- timed region: exactly `steps` frames, barrier + device sync on both sides --------------------------
    barrier(device); torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out_masks = []
    for i in range(args.steps):
        out_masks += one_step(args.warmup + i)
    out_masks += [m for _, m in fetcher.drain()]      # every mask of the timed steps is on the host before the clock stops
    assert len(out_masks) == args.steps
    torch.cuda.synchronize(device); barrier(device)
    elapsed = time.perf_counter() - t0
    # ---- per-kernel durations, measured live with HIP events on the launch stream: the launches of ONE frame are
    # recorded in an eager pass (the timed region replays captured HIP graphs, which cannot carry timing events) and
    # every distinct launch is then timed back to back (10 repetitions between two events).  profiles/ holds the
    # rocprofv3 kernel trace of the timed command itself.
    prof, prof_frames = {}, 0
    if rank == 0:
        core.cancel_prefetch()                           # the surveyed frame runs its own key encoder
        ops.RECORD = []
        one_step(args.warmup + args.steps)
        fetcher.drain()
        records, ops.RECORD = ops.RECORD, None
        prof = ops.time_recorded(records, reps=10)
        prof_frames = 1
    return dict(elapsed=elapsed, preload_s=preload_s, prof=prof, prof_frames=prof_frames, masks=out_masks, core=core,
                frames=frames, masks_in=masks, sd=sd, n_query=n_query)


def run_cpu_baseline(res, args, device):
    """The oracle (CPU restatement, bit-equal to the imported reference in the build container) on the same B32
    workload, bounded sample: preload 32 frames (untimed), 1 warm-up + `cpu_frames` timed frames, all host threads.
    Parity: a FRESH GPU stream (same preload, same frame order from the same start) is compared frame by frame."""
    from oracle import cpu_ref as R
    from xmem2_amd import InferenceCore, ops
    cfg = b32_config()
    ref = R.RefCore(R.RefNet(res['sd']), cfg)
    ref.set_all_labels([1])
    fr, mk = res['frames'], res['masks_in']
    gpu = InferenceCore(res['core'].network, cfg)
    gpu.set_all_labels([1])
    for j in range(MEM_FRAMES):
        ref.put_to_permanent_memory(torch.from_numpy(fr[j]), torch.from_numpy(mk[j]))
        gpu.put_to_permanent_memory(torch.from_numpy(fr[j]).to(device), torch.from_numpy(mk[j]).to(device))
    # GPU stream first, driven exactly like the timed region (batched key-encoder hints when enabled)
    n = args.cpu_frames + 1
    KB = max(1, args.key_batch)
    idx_of = lambda i: MEM_FRAMES + (i % res['n_query'])
    dev = [torch.from_numpy(fr[idx_of(i)]).to(device) for i in range(n + 2 * KB)]
    hint = (lambda a: gpu.prefetch_keys(dev[a:a + KB])) if not args.no_prefetch else (lambda a: None)
    hint(0)
    gpu_out = []
    for i in range(n):
        pg = gpu.step(dev[i], None, None)
        if i % KB == 0:
            hint(i + KB)
        gpu_out.append((ops.argmax_u8(pg).cpu().numpy(), pg.cpu()))
    gpu.cancel_prefetch()
    ious, mism, perr, times = [], 0, 0.0, []
    for i in range(n):
        t0 = time.perf_counter()
        p = ref.step(torch.from_numpy(fr[idx_of(i)]), None, None)
        m = R.post_process(p)
        dt = time.perf_counter() - t0
        if i >= 1:
            times.append(dt)
        g, pg = gpu_out[i]
        ious.append(R.compute_array_iou(g, m))
        mism += int((g != m).sum())
        perr = max(perr, float((pg - p).abs().max()))
    fps = len(times) / sum(times)
    return dict(value=fps, unit='frames/s', cores=torch.get_num_threads(), kind='port',
                sample=f'B32: 32 permanent frames preloaded (untimed), 1 warm-up + {len(times)} timed frames of '
                       f'step()+argmax at 480x854, oracle/cpu_ref.py on {torch.get_num_threads()} threads '
                       f'(host has {os.cpu_count()} logical CPUs)'), \
        dict(mask_iou_vs_cpu_min=float(min(ious)), argmax_mismatch_pixels=mism, frames_compared=len(ious),
             pixels_per_frame=H * W, max_abs_prob_err=perr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--cpu-frames', type=int, default=8, help='timed frames of the CPU baseline leg (rank 0, N=1 only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prefetch', action='store_true', help='do not pipeline the coming frames\' key encoder')
    ap.add_argument('--key-batch', type=int, default=4, help='frames per batched key-encoder hint (prefetch_keys)')
    ap.add_argument('--dist-backend', default='nccl', help='control-plane backend for the timing barrier / max-reduce '
                    '(nccl = RCCL; the data path has no collective)')
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    device = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # control plane only (timing barrier + max / sum of two scalars): RCCL by default
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
    res = run_gpu(args, device, rank, world)
    elapsed = max_over_ranks(res['elapsed'], device)
    total_frames = sum_over_ranks(args.steps, device)
    fps = total_frames / elapsed
    if rank == 0:
        alg = algorithmic_gflop_per_frame()
        prof = res['prof']
        conv = prof.get('conv', dict(ms=0.0, flop=0.0, launches=0))
        aff = prof.get('affinity', dict(ms=0.0, flop=0.0, launches=0))
        conv_tflops = conv['flop'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else None
        aff_tflops = aff['flop'] / (aff['ms'] * 1e-3) / 1e12 if aff['ms'] > 0 else None
        # HBM-side traffic per frame of the kernel families: PMC counters need rocprofv3 around the process, so bench.py
        # reports the committed measurement of this same command (profiles/, separate --pmc passes, gfx950 correction)
        traffic = {}
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_bench_b32_pmc_per_frame.json')) as f:
                fam = json.load(f)['families']
            traffic = {k: v['read_bytes'] + v['write_bytes'] for k, v in fam.items()}
        except Exception:
            pass
        line = {
            'metric': 'frames/sec at 480p, 1 obj, 32 memory frames; mask IoU vs reference',
            'value': fps, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'B32: synthetic 480x854 clip, 1 object, 32 permanent memory frames (N=51840), '
                                   'mem_every=1e9, step()+argmax per frame, conditioned synthetic weights',
                       'replica_streams': world, 'top_k': TOPK, 'frame_pipelining': (not args.no_prefetch), 'key_batch': (args.key_batch if not args.no_prefetch else 0), 'parallelism': f'{world} independent streams, no collectives'},
            'roofline': {'bound': 'mfma', 'kernel': 'xmem_conv2d_nhwc: conv_mfma_kernel (implicit GEMM / Winograd-domain GEMM, fp32 MFMA) + transforms',
                         'achieved': conv_tflops, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': (conv_tflops / PEAK_FP32_MFMA_TFLOPS) if conv_tflops else None, 'traffic': traffic.get('conv'),
                         'traffic_unit': 'HBM-side bytes per frame of these launches (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_bench_b32_pmc_per_frame.json)',
                         'measured': 'HIP events on the launch stream: every distinct convolution launch of one frame (un-hinted, batch-1 key encoder) timed over 10 back-to-back repetitions, Winograd transform kernels included; the timed region itself replays HIP graphs on two streams with batch-4 key passes',
                         'launches_per_frame': conv['launches'] / max(res['prof_frames'], 1),
                         'kernel_ms_per_frame': conv['ms'] / max(res['prof_frames'], 1),
                         'algorithmic_gflop_per_frame': conv['flop'] / 1e9 / max(res['prof_frames'], 1)},
            'affinity_roofline': {'bound': 'mfma', 'kernel': 'affinity_topk_kernel + merge (fused similarity/top-k/softmax)',
                                  'achieved': aff_tflops, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': (aff_tflops / PEAK_FP32_MFMA_TFLOPS) if aff_tflops else None,
                                  'kernel_ms_per_frame': aff['ms'] / max(res['prof_frames'], 1),
                                  'algorithmic_gflop_per_frame': alg['similarity'], 'traffic': traffic.get('affinity')},
            'frame_gflop': alg, 'whole_frame_tflops': alg['total'] / 1e3 * fps / world,
            'preload_s_per_rank': res['preload_s'],
        }
        if world == 1 and not args.no_cpu_baseline:
            cpu, parity = run_cpu_baseline(res, args, device)
            line['cpu_baseline'] = cpu
            line['parity'] = parity
            line['speedup_vs_cpu'] = fps / cpu['value']
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
