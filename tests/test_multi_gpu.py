"""The N>1 path: videos / replica streams are independent (no data-path collective), only the timing barrier and the
max-over-ranks reduction of bench.py use torch.distributed.  Covered here with world_size-2 gloo processes on CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        videos = [f'v{i}' for i in range(7)]
        lengths = [50, 10, 40, 30, 20, 60, 5]
        mine = bench.shard_videos(videos, lengths, rank, world)
        elapsed = 1.0 + rank                                    # rank 1 is the slow one
        t = bench.max_over_ranks(elapsed, torch.device('cpu'))
        frames = bench.sum_over_ranks(100, torch.device('cpu'))
        q.put((rank, mine, t, frames))
    finally:
        dist.destroy_process_group()


def test_sharding_and_reduction_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, v0, t0, f0), (r1, v1, t1, f1) = res
    assert set(v0) | set(v1) == {f'v{i}' for i in range(7)} and not (set(v0) & set(v1))
    assert v0 == ['v5', 'v3', 'v4'] and v1 == ['v0', 'v2', 'v1', 'v6']   # longest first, each to the least-loaded rank
    assert t0 == t1 == 2.0 and f0 == f1 == 200


def test_shard_single_process():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_videos(['a', 'b', 'c'], [1, 3, 2], 0, 1) == ['b', 'c', 'a']
    assert bench.max_over_ranks(1.5, torch.device('cpu')) == 1.5
    assert bench.gather_over_ranks(2.5, torch.device('cpu')) == [2.5]
    # equal lengths -> plain round-robin; every video exactly once for any world size
    vids = list(range(11))
    for world in (1, 2, 3, 8):
        parts = [bench.shard_videos(vids, [7] * 11, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == vids
        assert parts[0][:2] == ([0, 1] if world == 1 else [0, world])
    with pytest.raises(ValueError):
        bench.shard_videos(vids, [1] * 11, 2, 2)


def _make_videos(root, lengths):
    for i, n in enumerate(lengths):
        for sub in ('JPEGImages', 'Annotations'):
            d = root / sub / f'vid{i}'
            d.mkdir(parents=True)
            for t in range(n):
                (d / f'{t:05d}{".jpg" if sub == "JPEGImages" else ".png"}').write_bytes(b'x')


def test_launcher_world2_stub_runner(tmp_path):
    """xmem2_amd.launch with 2 ranks (stub run_on_video, CPU): the union of the per-rank outputs is every video exactly
    once, each rank processed the LPT share, and summary.json merges the per-rank results."""
    import json
    import subprocess
    lengths = [9, 3, 7, 5, 2]
    _make_videos(tmp_path, lengths)
    out = tmp_path / 'out'
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, 'tests'))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, '-m', 'xmem2_amd.launch', '--gpus', '2', '--device', 'cpu', '--runner', 'stub_runner:run',
                        '--videos', str(tmp_path / 'JPEGImages'), '--masks', str(tmp_path / 'Annotations'), '--out', str(out),
                        '--frames-with-masks', '0', '--compute-iou', '--config', '{"size": -1}'],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    summ = json.load(open(out / 'summary.json'))
    assert summ['n_gpus'] == 2 and summ['total_frames'] == sum(lengths) and not summ['ranks_missing']
    by_rank = {0: [], 1: []}
    for v in summ['videos']:
        by_rank[v['rank']].append(v['name'])
        assert abs(v['mean_iou'] - 0.5) < 1e-9
    assert sorted(by_rank[0] + by_rank[1]) == [f'vid{i}' for i in range(5)]
    assert set(by_rank[0]) == {'vid0', 'vid4', 'vid1'} and set(by_rank[1]) == {'vid2', 'vid3'}     # LPT: 9+2+3 | 7+5
    for i, n in enumerate(lengths):
        files = os.listdir(out / f'vid{i}' / 'masks')
        assert len(files) == n
        rank = 0 if f'vid{i}' in by_rank[0] else 1
        assert open(out / f'vid{i}' / 'masks' / files[0]).read().startswith(f'rank {rank} local {rank}')


def test_gpus_flag_is_not_silently_ignored():
    """`--gpus 2` on a box with fewer than 2 devices must fail, never report a 1-GPU number as n_gpus 2 (bench and launcher);
    under a torchrun environment a mismatching --gpus is refused as well."""
    import subprocess
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n >= 2:
        pytest.skip('box has >= 2 GPUs')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'refusing to run on fewer' in (p.stderr + p.stdout)
    assert '"n_gpus"' not in p.stdout
    env2 = dict(env, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env2, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'does not match WORLD_SIZE' in (p.stderr + p.stdout)


DAVIS_LIKE = [104, 100, 90, 84, 84, 82, 80, 80, 76, 75, 75, 70, 70, 69, 66, 65, 62, 60, 59, 55, 52, 50, 50, 50, 46, 43, 40, 40, 35, 34]


def test_lpt_deal_is_balanced_for_a_davis_like_list_at_world8():
    """BASELINE configs[2]: 30 videos over 8 GPUs.  The longest-first deal must leave the slowest rank within 5 % of the
    mean load (a plain round-robin of the unsorted list is off by > 15 %)."""
    sys.path.insert(0, ROOT)
    from xmem2_amd.launch import shard_videos
    import random
    vids = list(range(len(DAVIS_LIKE)))
    lengths = list(DAVIS_LIKE)
    random.Random(7).shuffle(lengths)
    parts = [shard_videos(vids, lengths, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == vids
    loads = [sum(lengths[v] for v in p) for p in parts]
    mean = sum(lengths) / 8
    assert max(loads) <= 1.05 * mean, f'LPT imbalance {max(loads) / mean:.3f} (loads {loads})'
    rr = [sum(lengths[v] for v in vids[r::8]) for r in range(8)]
    assert max(loads) <= max(rr)


def test_launcher_world8_stub_runner_stale_results_and_validation(tmp_path):
    """xmem2_amd.launch at world 8 (stub runner, CPU) on 30 DAVIS-like videos: every video once, per-rank frame loads within
    5 % of the mean, the summary carries per-rank totals; a result file left by an EARLIER run in the same directory is not
    merged (nonce); a video without an annotation folder is refused before any video runs."""
    import json
    import subprocess
    lengths = [max(1, n // 10) for n in DAVIS_LIKE]            # 30 videos, 3..10 frames (tiny files)
    _make_videos(tmp_path, lengths)
    out = tmp_path / 'out'
    out.mkdir()
    for r in range(8):                                          # stale results of "an earlier run"
        (out / f'_rank{r}.json').write_text(json.dumps(dict(rank=r, world=8, nonce='old', videos=[
            dict(name=f'ghost{r}', frames=1000, seconds=1.0, fps=1000.0, rank=r)])))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, 'tests'))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'XMEM_LAUNCH_NONCE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'xmem2_amd.launch', '--gpus', '8', '--device', 'cpu', '--runner', 'stub_runner:run',
           '--videos', str(tmp_path / 'JPEGImages'), '--masks', str(tmp_path / 'Annotations'), '--out', str(out),
           '--frames-with-masks', '0', '--config', '{"size": -1}']
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    summ = json.load(open(out / 'summary.json'))
    assert summ['n_gpus'] == 8 and summ['total_frames'] == sum(lengths) and not summ['ranks_missing']
    assert sorted(v['name'] for v in summ['videos']) == sorted(f'vid{i}' for i in range(30))      # no ghosts
    loads = [summ['per_rank'][str(r)]['frames'] for r in range(8)]
    assert sum(loads) == sum(lengths) and max(loads) <= 1.08 * sum(lengths) / 8, loads
    assert summ['slowest_rank_seconds'] >= summ['fastest_rank_seconds'] > 0
    # validation before work: remove one annotation folder -> every rank refuses, nothing is processed
    import shutil
    shutil.rmtree(tmp_path / 'Annotations' / 'vid29')
    out2 = tmp_path / 'out2'
    cmd2 = [c if c != str(out) else str(out2) for c in cmd]
    p = subprocess.run(cmd2, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and 'no annotation directory' in (p.stderr + p.stdout)
    assert not any((out2 / f'vid{i}').exists() for i in range(30))


def test_rank_placement_helpers():
    """CPU pinning of a rank: cpulist parsing, GPU-local CPUs from sysfs, even split among the ranks of one NUMA node, even
    slices of the allowed set when the topology is unknown; device isolation env."""
    sys.path.insert(0, ROOT)
    from xmem2_amd import launch as L
    assert L.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert L.parse_cpulist('') == []
    allowed = range(64)
    # unknown topology: 8 ranks -> 8 disjoint slices of 8 that cover the set
    slices = [L.rank_cpu_set(allowed, None, r, 8) for r in range(8)]
    assert sorted(sum(slices, [])) == list(range(64)) and all(len(s) == 8 for s in slices)
    # known: 4 GPUs per NUMA node of 32 cores -> each gets 8 of ITS node's cores
    node1 = list(range(32, 64))
    got = [L.rank_cpu_set(allowed, node1, r, 8, share_with=(4, r % 4)) for r in range(4, 8)]
    assert sorted(sum(got, [])) == node1
    # the process may be confined (cgroup / taskset): never outside the allowed set, never empty
    assert L.rank_cpu_set([2, 3], node1, 5, 8, share_with=(4, 1)) == [2, 3]
    assert L.rank_cpu_set([40, 41], node1, 7, 8, share_with=(4, 3)) == [40, 41]
    # sysfs lookup
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, '0000:05:00.0'))
        with open(os.path.join(d, '0000:05:00.0', 'local_cpulist'), 'w') as f:
            f.write('0-15,128-143\n')
        assert L.gpu_local_cpus('0000:05:00.0', sysfs=d) == list(range(16)) + list(range(128, 144))
        assert L.gpu_local_cpus('0000:06:00.0', sysfs=d) is None and L.gpu_local_cpus(None, sysfs=d) is None
    assert L.isolated_device_env(3, {}) == dict(HIP_VISIBLE_DEVICES='3', CUDA_VISIBLE_DEVICES='3', LOCAL_RANK='0', XMEM_DEVICE_ORDINAL='3')
    assert L.isolated_device_env(1, {'HIP_VISIBLE_DEVICES': '4,6,7'})['HIP_VISIBLE_DEVICES'] == '6'
    # a parent limited through the CUDA-style variable only: the ranks stay inside that set, and both variables agree in the child
    e = L.isolated_device_env(2, {'CUDA_VISIBLE_DEVICES': '4,5,6,7'})
    assert e['HIP_VISIBLE_DEVICES'] == e['CUDA_VISIBLE_DEVICES'] == e['XMEM_DEVICE_ORDINAL'] == '6'
    # HIP_VISIBLE_DEVICES wins where both are set (the HIP runtime's own precedence)
    assert L.isolated_device_env(0, {'HIP_VISIBLE_DEVICES': '2,3', 'CUDA_VISIBLE_DEVICES': '6,7'})['CUDA_VISIBLE_DEVICES'] == '2'
    with pytest.raises(ValueError):
        L.isolated_device_env(4, {'CUDA_VISIBLE_DEVICES': '4,5,6,7'})
    # pin_rank in a single-process world changes nothing
    before = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else None
    info = L.pin_rank(0, 1)
    assert info['cpus'] is None and (before is None or os.sched_getaffinity(0) == before)
