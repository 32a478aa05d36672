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


def spawn_ranks(argv, world, extra_env=None, check_devices=True, timeout=None):
    """Start `world` copies of `python <argv...>` with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT), one per GPU, and wait for them.  Returns the list of exit codes.
    Rank r uses device LOCAL_RANK = r of the visible devices (as torch.distributed.run does)."""
    if world < 1:
        raise ValueError('world must be >= 1')
    if check_devices:
        n = visible_gpus()
        if n < world:
            raise SystemExit(f'--gpus {world} requested but only {n} MI355X device(s) are visible: refusing to run on fewer')
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
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
    videos = read_video_list(args.videos, args.masks)
    mine = shard_videos(videos, [v['length'] for v in videos], rank, world)
    if args.device != 'cpu':
        import torch
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
            raise SystemExit(f'rank {rank}: device {local} is not visible')
        torch.cuda.set_device(local)
    runner = _resolve(args.runner)
    over = json.loads(args.config) if args.config else {}
    fm = [int(x) for x in args.frames_with_masks.split(',') if x != '']
    results = []
    for v in mine:
        if not v['masks']:
            raise SystemExit(f'video {v["name"]}: no annotation directory (pass --masks or list it)')
        t0 = time.perf_counter()
        stats = runner(v['frames'], v['masks'], os.path.join(args.out, v['name']), frames_with_masks=fm,
                       compute_iou=args.compute_iou, print_progress=False, overwrite_config=dict(over))
        dt = time.perf_counter() - t0
        row = dict(name=v['name'], frames=v['length'], seconds=dt, fps=v['length'] / dt if dt > 0 else None, rank=rank)
        if args.compute_iou and stats is not None and 'iou' in getattr(stats, 'columns', ()):
            ious = [float(x) for x in stats['iou'] if x >= 0]
            row['mean_iou'] = sum(ious) / len(ious) if ious else None
        results.append(row)
    os.makedirs(args.out, exist_ok=True)
    tmp = os.path.join(args.out, f'_rank{rank}.json.tmp')
    with open(tmp, 'w') as f:
        json.dump(dict(rank=rank, world=world, videos=results), f)
    os.replace(tmp, os.path.join(args.out, f'_rank{rank}.json'))
    return 0


def merge(out_dir, world, wall):
    videos, missing = [], []
    for r in range(world):
        p = os.path.join(out_dir, f'_rank{r}.json')
        if not os.path.exists(p):
            missing.append(r)
            continue
        with open(p) as f:
            videos += json.load(f)['videos']
    frames = sum(v['frames'] for v in videos)
    summary = dict(n_gpus=world, videos=sorted(videos, key=lambda v: v['name']), total_frames=frames, wall_seconds=wall,
                   aggregate_fps=frames / wall if wall > 0 else None, ranks_missing=missing)
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
            while time.time() < deadline and not all(os.path.exists(os.path.join(args.out, f'_rank{r}.json')) for r in range(args.gpus)):
                time.sleep(0.5)
            merge(args.out, args.gpus, time.perf_counter() - t0)
        return rc
    os.makedirs(args.out, exist_ok=True)
    for r in range(args.gpus):
        p = os.path.join(args.out, f'_rank{r}.json')
        if os.path.exists(p):
            os.remove(p)
    t0 = time.perf_counter()
    child = ['-m', 'xmem2_amd.launch', '--as-worker'] + [a for a in (argv if argv is not None else sys.argv[1:])]
    codes = spawn_ranks(child, args.gpus, check_devices=(args.device != 'cpu'),
                        extra_env={'PYTHONPATH': os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                                                 [p for p in os.environ.get('PYTHONPATH', '').split(os.pathsep) if p])})
    wall = time.perf_counter() - t0
    summary = merge(args.out, args.gpus, wall)
    print(json.dumps({k: summary[k] for k in ('n_gpus', 'total_frames', 'wall_seconds', 'aggregate_fps', 'ranks_missing')}))
    if any(c != 0 for c in codes) or summary['ranks_missing']:
        raise SystemExit(f'rank exit codes {codes}, missing results from ranks {summary["ranks_missing"]}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
