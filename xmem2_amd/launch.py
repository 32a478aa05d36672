"""Multi-GPU launcher: one process per GPU, each an independent stream of whole videos.

The reference evaluates one ``InferenceCore`` per video (``eval.py:160-163``) and orders experiment videos by
length (``inference/run_experiments.py:418``); videos never exchange state, so the path shards by video with
**no data-path collective**.  This module is the product-side entry for that:

    python -m xmem2_amd.launch --gpus 8 --videos /data/DAVIS/JPEGImages/480p --masks /data/DAVIS/Annotations/480p \
                               --out /results --frames-with-masks 0

* ``--videos`` is a directory whose sub-directories are videos (frames inside), or a text file with one
  ``<frames dir>[,<masks dir>[,<name>]]`` per line; ``--masks`` is the parallel directory of annotation folders.
* videos are sorted longest first and dealt to the rank with the least work so far (LPT; ``shard_videos``), every
  rank then runs ``run_on_video`` on its videos in that order on its own GPU (``LOCAL_RANK``), writes
  ``<out>/<name>/masks/*.png`` and a per-rank ``<out>/_rank<r>.json``; the parent merges them into
  ``<out>/summary.json`` (per-video frames, seconds, frames/s, mean IoU if asked) - host-side merge, no RCCL.
* ``--gpus N`` fails if fewer than N devices are visible: it never silently runs on fewer.

``bench.py`` uses ``spawn_ranks`` of this module for its own ``--gpus N`` replica streams.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time


def shard_videos(videos, lengths, rank, world):
    """Longest-processing-time-first deal: videos sorted by length (descending, ties by position) are given one by one
    to the rank with the least frames so far (ties -> lowest rank).  Deterministic, disjoint, covers every video; each
    rank's list stays in longest-first order.  With equal lengths this is a plain round-robin."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f'bad rank/world {rank}/{world}')
    if len(videos) != len(lengths):
        raise ValueError('videos and lengths differ in length')
    order = sorted(range(len(videos)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        load[r] += max(int(lengths[i]), 0)
        if r == rank:
            mine.append(videos[i])
    return mine


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def isolated_device_env(r, parent_env):
    """Environment that shows rank r exactly ONE GPU (the r-th of what the parent sees), LOCAL_RANK=0.  The parent's set is read
    from HIP_VISIBLE_DEVICES, else from CUDA_VISIBLE_DEVICES (a node shared through the CUDA-style variable, which the HIP
    runtime honours when its own is absent); BOTH variables are set to the rank's id in the child so that they cannot disagree
    (a child given only HIP_VISIBLE_DEVICES=r under a parent limited by CUDA_VISIBLE_DEVICES=4,5,6,7 would land on physical GPU r,
    outside the allotted set).  A rank then cannot allocate on, or synchronise with, another rank's device by accident (torch's
    default device 0 IS its own GPU).  XMEM_DEVICE_ORDINAL keeps the node-wide ordinal for CPU pinning and logs."""
    vis = parent_env.get('HIP_VISIBLE_DEVICES')
    if vis is None or vis.strip() == '':
        vis = parent_env.get('CUDA_VISIBLE_DEVICES')
    ids = [v.strip() for v in vis.split(',') if v.strip() != ''] if vis else None
    if ids is not None and r >= len(ids):
        raise ValueError(f'rank {r} has no device: the visible set is {ids}')
    dev = ids[r] if ids else str(r)
    return dict(HIP_VISIBLE_DEVICES=dev, CUDA_VISIBLE_DEVICES=dev, LOCAL_RANK='0', XMEM_DEVICE_ORDINAL=dev)


def spawn_ranks(argv, world, extra_env=None, check_devices=True, timeout=None, isolate_devices=False, nonce=None):
    """Start `world` copies of `python <argv...>` with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT), one per GPU, and wait for them.  Returns the list of exit codes.
    Rank r uses device LOCAL_RANK = r of the visible devices (as torch.distributed.run does); with `isolate_devices` each
    rank sees only its own GPU (`isolated_device_env`; used by the video launcher, whose ranks exchange nothing on the
    device - bench.py keeps every device visible for its RCCL control plane).  Every rank gets the same
    XMEM_LAUNCH_NONCE so that results of an earlier run in the same output directory are never mistaken for this one's."""
    if world < 1:
        raise ValueError('world must be >= 1')
    if check_devices:
        n = visible_gpus()
        if n < world:
            raise SystemExit(f'--gpus {world} requested but only {n} MI355X device(s) are visible: refusing to run on fewer')
    port = free_port()
    nonce = nonce or f'{os.getpid()}-{port}-{time.time_ns()}'
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), XMEM_LAUNCH_NONCE=nonce)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if isolate_devices:
            env.update(isolated_device_env(r, os.environ))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    codes = []
    deadline = time.time() + timeout if timeout else None
    for p in procs:
        try:
            codes.append(p.wait(timeout=max(1.0, deadline - time.time()) if deadline else None))
        except subprocess.TimeoutExpired:
            for q in procs:                       # exactly the processes started here
                if q.poll() is None:
                    q.kill()
            codes.append(-9)
    return codes


# ---- host-side placement of a rank -------------------------------------------------------------------------------
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out = []
    for part in text.strip().split(','):
        part = part.strip()
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-')
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_local_cpus(pci_bus_id, sysfs='/sys/bus/pci/devices'):
    """CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function), or None when unknown."""
    if not pci_bus_id:
        return None
    for name in (pci_bus_id.lower(), pci_bus_id.upper()):
        try:
            with open(os.path.join(sysfs, name, 'local_cpulist')) as f:
                cpus = parse_cpulist(f.read())
            return cpus or None
        except (OSError, ValueError):
            continue
    return None


def rank_cpu_set(allowed, local_cpus, local_rank, local_world, share_with=None):
    """The CPUs a rank should run on: the GPU-local CPUs (intersected with what the process may use) split evenly among
    the `share_with` ranks whose GPUs sit on the same node (index `local_rank` among them), or - when the topology is
    unknown - an even slice of the allowed set.  Never empty; deterministic."""
    allowed = sorted(allowed)
    base = sorted(set(allowed) & set(local_cpus)) if local_cpus else []
    if base:
        peers, me = (share_with if share_with else (1, 0))
    else:
        base, peers, me = allowed, max(1, local_world), local_rank
    per = max(1, len(base) // max(1, peers))
    mine = base[me * per:(me + 1) * per] if me < peers else []
    return mine or base


def pin_rank(local_rank=None, local_world=None, max_threads=8, device_index=None, verbose=False):
    """One process per GPU: pin this rank to the CPU cores next to its GPU (sched_setaffinity) and cap torch's intra-op
    threads (the host side of a stream is launch-bound Python + a few decode / writer threads; 256 default OpenMP threads per
    rank times 8 ranks oversubscribe the box).  eval.py:160-163 runs one core per video; this is the N-process version.
    Returns dict(cpus, threads, numa_known).  No-op on platforms without sched_setaffinity."""
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else local_rank
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))) if local_world is None else local_world
    # index of this rank among the node's ranks (LOCAL_RANK is 0 for every rank when each sees only its own device)
    isolated = 'XMEM_DEVICE_ORDINAL' in os.environ
    slot = (int(os.environ.get('RANK', '0')) % max(1, local_world)) if isolated else local_rank
    info = dict(cpus=None, threads=None, numa_known=False)
    import torch
    if hasattr(os, 'sched_getaffinity') and local_world > 1:
        allowed = os.sched_getaffinity(0)
        local, share = None, None
        try:
            if torch.cuda.is_available():
                n = torch.cuda.device_count()
                ids = [getattr(torch.cuda.get_device_properties(i), 'pci_bus_id', None) for i in range(n)]
                fmt = lambda p, i: p if isinstance(p, str) else (
                    '%04x:%02x:%02x.0' % (getattr(torch.cuda.get_device_properties(i), 'pci_domain_id', 0), p,
                                          getattr(torch.cuda.get_device_properties(i), 'pci_device_id', 0)) if p is not None else None)
                di = device_index if device_index is not None else (local_rank if n > local_rank else 0)
                local = gpu_local_cpus(fmt(ids[di], di))
                if local and n > 1:          # ranks whose GPUs share this NUMA node split its cores
                    same = [i for i in range(n) if gpu_local_cpus(fmt(ids[i], i)) == local]
                    share = (len(same), same.index(di))
                elif local:                  # isolated devices: assume the node's GPUs are spread evenly over the NUMA nodes
                    nodes = max(1, len(allowed) // max(1, len(local)))
                    share = (max(1, local_world // nodes), (slot % max(1, local_world // nodes)))
        except Exception:
            local = None
        cpus = rank_cpu_set(allowed, local, slot, local_world, share)
        try:
            os.sched_setaffinity(0, cpus)
            info.update(cpus=len(cpus), numa_known=bool(local))
        except OSError:
            pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    threads = max(1, min(max_threads, avail))
    if local_world > 1:
        torch.set_num_threads(threads)
        info['threads'] = threads
    if verbose:
        print(f'[launch] rank {os.environ.get("RANK", "0")}: {info}', file=sys.stderr)
    return info


# ---- video lists ---------------------------------------------------------------------------------------------
_IMG_EXT = ('.jpg', '.jpeg', '.png', '.JPG', '.JPEG', '.PNG')


def count_frames(frames_dir):
    try:
        return sum(1 for f in os.listdir(frames_dir) if f.endswith(_IMG_EXT))
    except OSError:
        return 0


def read_video_list(videos, masks=None):
    """-> list of dict(name, frames, masks, length)."""
    out = []
    if os.path.isdir(videos):
        for name in sorted(os.listdir(videos)):
            fdir = os.path.join(videos, name)
            if not os.path.isdir(fdir):
                continue
            mdir = os.path.join(masks, name) if masks else None
            out.append(dict(name=name, frames=fdir, masks=mdir))
    else:
        with open(videos) as f:
            for line in f:
                line = line.strip()
                if not line or line.startswith('#'):
                    continue
                parts = [p.strip() for p in line.split(',')]
                fdir = parts[0]
                name = parts[2] if len(parts) > 2 else os.path.basename(os.path.normpath(fdir))
                mdir = parts[1] if len(parts) > 1 and parts[1] else (os.path.join(masks, name) if masks else None)
                out.append(dict(name=name, frames=fdir, masks=mdir))
    for v in out:
        v['length'] = count_frames(v['frames'])
    names = [v['name'] for v in out]
    if len(set(names)) != len(names):
        raise ValueError('video names must be unique (they name the output folders)')
    return out


def _resolve(spec):
    mod, fn = spec.split(':')
    return getattr(importlib.import_module(mod), fn)


# ---- per-rank worker -------------------------------------------------------------------------------------------
def worker(args):
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    os.makedirs(args.out, exist_ok=True)
    mine_path = os.path.join(args.out, f'_rank{rank}.json')
    if os.path.exists(mine_path):                       # a result of an EARLIER run in this directory is not this run's
        os.remove(mine_path)
    videos = read_video_list(args.videos, args.masks)
    # every video is checked before any video runs: a missing annotation folder must not cost the work done before it
    bad = [v['name'] for v in videos if not v['masks'] or not os.path.isdir(v['masks'])]
    if bad:
        raise SystemExit(f'no annotation directory for video(s) {bad[:8]}{"..." if len(bad) > 8 else ""} (pass --masks or list it)')
    mine = shard_videos(videos, [v['length'] for v in videos], rank, world)
    if args.device != 'cpu':
        import torch
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
            raise SystemExit(f'rank {rank}: device {local} is not visible')
        torch.cuda.set_device(local)
    if world > 1:
        pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)), max_threads=args.threads_per_rank)
    runner = _resolve(args.runner)
    over = json.loads(args.config) if args.config else {}
    fm = [int(x) for x in args.frames_with_masks.split(',') if x != '']
    results = []
    for v in mine:
        t0 = time.perf_counter()
        stats = runner(v['frames'], v['masks'], os.path.join(args.out, v['name']), frames_with_masks=fm,
                       compute_iou=args.compute_iou, print_progress=False, overwrite_config=dict(over))
        dt = time.perf_counter() - t0
        row = dict(name=v['name'], frames=v['length'], seconds=dt, fps=v['length'] / dt if dt > 0 else None, rank=rank)
        if args.compute_iou and stats is not None and 'iou' in getattr(stats, 'columns', ()):
            ious = [float(x) for x in stats['iou'] if x >= 0]
            row['mean_iou'] = sum(ious) / len(ious) if ious else None
        results.append(row)
    tmp = mine_path + '.tmp'
    with open(tmp, 'w') as f:
        json.dump(dict(rank=rank, world=world, nonce=run_nonce(), videos=results), f)
    os.replace(tmp, mine_path)
    return 0


def run_nonce():
    """Identifies ONE launch across its ranks: set by spawn_ranks; under torch.distributed.run every local rank has the
    same agent process as parent and the same rendezvous port."""
    return os.environ.get('XMEM_LAUNCH_NONCE') or \
        f'{os.getppid()}-{os.environ.get("MASTER_PORT", "")}-{os.environ.get("TORCHELASTIC_RUN_ID", "")}'


def rank_result(out_dir, r, nonce=None):
    """The per-rank result of THIS launch (matching nonce), or None (absent, unreadable, or left by an earlier run)."""
    p = os.path.join(out_dir, f'_rank{r}.json')
    try:
        with open(p) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return None
    if nonce is not None and j.get('nonce') != nonce:
        return None
    return j


def merge(out_dir, world, wall, nonce=None):
    videos, missing = [], []
    for r in range(world):
        j = rank_result(out_dir, r, nonce)
        if j is None:
            missing.append(r)
            continue
        videos += j['videos']
    frames = sum(v['frames'] for v in videos)
    per_rank = {}
    for v in videos:
        d = per_rank.setdefault(v['rank'], dict(frames=0, seconds=0.0, videos=0))
        d['frames'] += v['frames']; d['seconds'] += v['seconds']; d['videos'] += 1
    busy = [d['seconds'] for d in per_rank.values()]
    summary = dict(n_gpus=world, videos=sorted(videos, key=lambda v: v['name']), total_frames=frames, wall_seconds=wall,
                   aggregate_fps=frames / wall if wall > 0 else None, ranks_missing=missing,
                   per_rank={str(k): v for k, v in sorted(per_rank.items())},
                   slowest_rank_seconds=max(busy) if busy else None, fastest_rank_seconds=min(busy) if busy else None)
    with open(os.path.join(out_dir, 'summary.json'), 'w') as f:
        json.dump(summary, f, indent=1)
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser(description='Run run_on_video over many videos, sharded over the GPUs of one node.')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--videos', required=True)
    ap.add_argument('--masks', default=None)
    ap.add_argument('--out', required=True)
    ap.add_argument('--frames-with-masks', default='0', help='comma-separated frame indices whose annotation is given')
    ap.add_argument('--config', default=None, help='JSON dict merged into VIDEO_INFERENCE_CONFIG (overwrite_config)')
    ap.add_argument('--compute-iou', action='store_true')
    ap.add_argument('--runner', default='xmem2_amd.run_on_video:run_on_video', help='module:function with run_on_video\'s signature')
    ap.add_argument('--device', default='cuda', choices=['cuda', 'cpu'], help='cpu only for launcher tests with a stub runner')
    ap.add_argument('--merge-timeout', type=float, default=86400.0, help='under torchrun: how long rank 0 waits for the other ranks\' results')
    ap.add_argument('--threads-per-rank', type=int, default=8, help='torch intra-op threads per rank (ranks are pinned to the cores next to their GPU)')
    ap.add_argument('--no-isolate', action='store_true', help='keep every GPU visible in every rank (default: a rank sees only its own)')
    ap.add_argument('--as-worker', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    under_torchrun = (not args.as_worker) and int(os.environ.get('WORLD_SIZE', '1')) > 1 and 'RANK' in os.environ
    if args.as_worker or under_torchrun:
        if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
            raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE={os.environ.get("WORLD_SIZE")}')
        t0 = time.perf_counter()
        rc = worker(args)
        if under_torchrun and int(os.environ['RANK']) == 0:        # launched by torch.distributed.run: rank 0 merges
            deadline = time.time() + args.merge_timeout
            nonce = run_nonce()
            while time.time() < deadline and not all(rank_result(args.out, r, nonce) is not None for r in range(args.gpus)):
                time.sleep(0.5)
            merge(args.out, args.gpus, time.perf_counter() - t0, nonce)
        return rc
    os.makedirs(args.out, exist_ok=True)
    for r in range(args.gpus):
        p = os.path.join(args.out, f'_rank{r}.json')
        if os.path.exists(p):
            os.remove(p)
    t0 = time.perf_counter()
    child = ['-m', 'xmem2_amd.launch', '--as-worker'] + [a for a in (argv if argv is not None else sys.argv[1:])]
    nonce = f'{os.getpid()}-{time.time_ns()}'
    codes = spawn_ranks(child, args.gpus, check_devices=(args.device != 'cpu'), nonce=nonce,
                        isolate_devices=(args.device != 'cpu' and not args.no_isolate),
                        extra_env={'PYTHONPATH': os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                                                 [p for p in os.environ.get('PYTHONPATH', '').split(os.pathsep) if p])})
    wall = time.perf_counter() - t0
    summary = merge(args.out, args.gpus, wall, nonce)
    print(json.dumps({k: summary[k] for k in ('n_gpus', 'total_frames', 'wall_seconds', 'aggregate_fps', 'ranks_missing',
                                              'slowest_rank_seconds', 'fastest_rank_seconds')}))
    if any(c != 0 for c in codes) or summary['ranks_missing']:
        raise SystemExit(f'rank exit codes {codes}, missing results from ranks {summary["ranks_missing"]}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
