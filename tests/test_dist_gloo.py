"""CPU tests (gloo, world sizes 2 and 8) of the multi-GPU plumbing: contiguous frame sharding, the single
all-gather of padded detection records per batch (RCCL on the GPUs, gloo here) and bench.py's own launcher
(`python bench.py --gpus N` starts its N ranks itself)."""
import json
import subprocess
import sys
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolov3.dist import all_gather_records, counts_of, pack_records_host, regather_if_truncated, shard_range, unpack_records

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _fake_dets(frame_id):
    rs = np.random.RandomState(frame_id)
    k = int(rs.randint(0, 9))
    c = rs.randint(0, 500, size=(k, 2))
    tlbr = np.concatenate([c, c + rs.randint(1, 50, size=(k, 2))], axis=1).astype(np.int64)
    return [tlbr, rs.rand(k).astype(np.float32), rs.randint(0, 80, size=k).astype(np.int64),
            rs.randint(0, 22743, size=k).astype(np.int64)]


def _worker(rank, world, port, n_frames, kmax, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_frames, rank, world)
    dets = [_fake_dets(f) for f in range(lo, hi)]
    rec = pack_records_host(dets, kmax)
    all_rec = all_gather_records(torch.from_numpy(rec), world)          # ONE collective per batch
    assert counts_of(all_rec).shape == (n_frames,)
    out = unpack_records(all_rec)
    q.put((rank, [(d[0].tolist(), d[1].tolist(), d[2].tolist(), d[3].tolist(), d[4]) for d in out]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(128, 3, 8) == (48, 64)


def test_pack_unpack_roundtrip_and_truncation():
    dets = [_fake_dets(f) for f in range(5)]
    rec = pack_records_host(dets, 4)
    assert counts_of(rec).tolist() == [len(d[1]) for d in dets]      # the true count rides in the records
    back = unpack_records(rec)
    for d, b in zip(dets, back):
        k = min(len(d[1]), 4)
        assert (b[0] == d[0][:k]).all() and (b[1] == d[1][:k]).all() and (b[2] == d[2][:k]).all()
        assert b[4] == (len(d[1]) > 4)


@pytest.mark.parametrize("world,n_frames", [(2, 8), (8, 128)])
def test_all_gather_gloo(world, n_frames):
    """(8, 128) is BASELINE.json configs[4]: 128 frames over 8 ranks, 16 each."""
    kmax = 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, kmax, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_fake_dets(f) for f in range(n_frames)]
    for rank in range(world):
        assert len(got[rank]) == n_frames            # every rank holds every frame, in frame order
        for f in range(n_frames):
            tl, pr, cl, rw, trunc = got[rank][f]
            assert tl == want[f][0].tolist() and cl == want[f][2].tolist() and rw == want[f][3].tolist()
            assert np.allclose(pr, want[f][1]) and not trunc


def _big_dets(frame_id, k):
    rs = np.random.RandomState(1000 + frame_id)
    c = rs.randint(0, 500, size=(k, 2))
    tlbr = np.concatenate([c, c + rs.randint(1, 50, size=(k, 2))], axis=1).astype(np.int64)
    return [tlbr, rs.rand(k).astype(np.float32), rs.randint(0, 80, size=k).astype(np.int64), np.arange(k, dtype=np.int64)]


def _ragged_frames(n_frames, kmax):
    """Frame 1 keeps 2 x kmax boxes, frame n - 1 keeps kmax + 1, the others a handful."""
    return [_big_dets(f, 2 * kmax) if f == 1 else (_big_dets(f, kmax + 1) if f == n_frames - 1 else _fake_dets(f)) for f in range(n_frames)]


def _ragged_worker(rank, world, port, n_frames, kmax, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_frames, rank, world)
    dets = _ragged_frames(n_frames, kmax)[lo:hi]
    first = all_gather_records(torch.from_numpy(pack_records_host(dets, kmax)), world)
    calls = []

    def repack(kmax2):
        calls.append(kmax2)
        return pack_records_host(dets, kmax2)
    full = regather_if_truncated(first, kmax, repack, world)
    out = unpack_records(full)
    q.put((rank, calls, [(d[0].tolist(), d[1].tolist(), d[2].tolist(), d[3].tolist(), d[4]) for d in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_frames_with_more_than_kmax_boxes_are_gathered_in_full():
    """VERDICT r04 item 4a / SURVEY.md 8(e) "ragged alternative": with N ranks a frame that keeps more than kmax boxes is not
    truncated -- the true counts ride in the first gather, every rank sees the same maximum, packs again with room for it and
    ONE more all-gather returns every kept box of every frame on every rank, equal to the single-rank lists."""
    world, n_frames, kmax = 2, 6, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, n_frames, kmax, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, calls, frames = q.get(timeout=120)
        got[rank] = (calls, frames)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _ragged_frames(n_frames, kmax)
    for rank in range(world):
        calls, frames = got[rank]
        assert calls == [64]                          # one re-pack on EVERY rank (also the one whose frames all fit): 2 x 16 -> 64
        assert len(frames) == n_frames
        for f in range(n_frames):
            tl, pr, cl, rw, trunc = frames[f]
            assert not trunc and len(pr) == len(want[f][1])
            assert tl == want[f][0].tolist() and cl == want[f][2].tolist() and rw == want[f][3].tolist()
            assert np.array_equal(np.array(pr, dtype=np.float32), want[f][1])
    # nothing above kmax: the first gather is returned as it is, no second collective
    rec = torch.from_numpy(pack_records_host([_fake_dets(f) for f in range(4)], kmax))
    assert regather_if_truncated(rec, kmax, lambda k: (_ for _ in ()).throw(AssertionError("no repack expected")), 1) is rec


def _run_bench(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` with fewer than N GPUs visible must fail loudly, never run fewer ranks."""
    n_visible = torch.cuda.device_count()
    want = n_visible + 2
    r = _run_bench(["--gpus", str(want), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "%d GPUs requested, %d visible" % (want, n_visible) in (r.stderr + r.stdout)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launch_starts_n_ranks(world):
    """The driver runs `python bench.py --gpus N ...` without torchrun: the parent process must start N ranks itself
    (torch.distributed.run), relay rank 0's one JSON line and propagate failures.  Exercised here with the
    plumbing-only mode (Y3_BENCH_PLUMBING=gloo: no GPU work, the same shard / gather / timing skeleton over gloo)."""
    r = _run_bench(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "16", "--no-cpu-baseline"],
                   {"Y3_BENCH_PLUMBING": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["config"]["global_batch"] == 16 * world
    assert line["plumbing_only"] is True and line["ranks_seen_by_collective"] == world
    assert line["frames_gathered_per_step"] == 16 * world and line["scaling"] == "weak"
    pr = line["per_rank"]      # a slow or starved rank must be visible in the line, not only in the maximum
    assert len(pr["ms_per_step_by_rank"]) == world == len(pr["host_enqueue_ms_per_step_by_rank"])
    assert pr["ms_per_step_min"] <= pr["ms_per_step_max"] and pr["usable_cpus_per_rank"] >= 0
    # round 5: R timed windows (the median is reported, min / max alongside) and every rank's CPU placement in the line
    assert line["repeats"]["windows"] == 5
    assert line["repeats"]["ms_per_step_min"] <= line["ms_per_step"] <= line["repeats"]["ms_per_step_max"]
    assert len(pr["placement"]) == world and all("bound" in p and p["cpus"] >= 1 for p in pr["placement"])


def test_bench_strong_scaling_splits_128_frames_over_8_ranks():
    """128 frames in total, split over the ranks (BASELINE.json configs[4] at N = 8: 16 per GPU) -- the
    same launcher / shard / one-gather-per-step / max-over-ranks skeleton as the weak-scaling run."""
    r = _run_bench(["--gpus", "8", "--scaling", "strong", "--total-frames", "128", "--steps", "3", "--warmup", "1",
                    "--no-cpu-baseline"], {"Y3_BENCH_PLUMBING": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["scaling"] == "strong" and line["n_gpus"] == 8
    assert line["config"]["global_batch"] == 128 and line["config"]["frames_per_gpu"] == [16] * 8
    assert line["frames_gathered_per_step"] == 128
    assert len(line["per_rank"]["ms_per_step_by_rank"]) == 8
    r2 = _run_bench(["--gpus", "2", "--scaling", "strong", "--total-frames", "128", "--steps", "2", "--warmup", "0",
                     "--no-cpu-baseline"], {"Y3_BENCH_PLUMBING": "gloo"})
    line2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])
    assert line2["config"]["frames_per_gpu"] == [64, 64] and line2["frames_gathered_per_step"] == 128


def test_bench_self_launch_propagates_rank_failure():
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"Y3_BENCH_PLUMBING": "gloo", "Y3_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_launcher_kills_a_job_with_a_hanging_rank():
    """A rank that never reaches the rendezvous: the others give up after the bounded init timeout or the watchdog kills the
    whole process group at --launch-timeout -- either way non-zero, no JSON line, well inside a minute, nobody left behind."""
    import time
    t0 = time.time()
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-timeout", "25"],
                   {"Y3_BENCH_PLUMBING": "gloo", "Y3_BENCH_HANG_RANK": "1", "Y3_BENCH_INIT_TIMEOUT": "600"}, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 60
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "did not finish within 25 s" in r.stderr and "process group killed" in r.stderr
    # bounded init on its own: with a short init timeout the waiting rank fails by itself before the watchdog fires
    t0 = time.time()
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-timeout", "50"],
                   {"Y3_BENCH_PLUMBING": "gloo", "Y3_BENCH_HANG_RANK": "1", "Y3_BENCH_INIT_TIMEOUT": "8"}, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 60
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_launcher_reports_a_rank_that_dies_after_init():
    import time
    t0 = time.time()
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--launch-timeout", "50"],
                   {"Y3_BENCH_PLUMBING": "gloo", "Y3_BENCH_DIE_AFTER_INIT": "1"}, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 60
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "dying after init" in r.stderr          # the dead rank's last stderr lines are in the report
