"""world_size-2 CPU test (gloo) of the multi-GPU plumbing: contiguous frame sharding and the
single all-gather of padded detection records (RCCL on the GPUs, gloo here)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolov3.dist import all_gather_records, pack_records_host, shard_range, unpack_records


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
    rec, cnt = pack_records_host(dets, kmax)
    all_rec, all_cnt = all_gather_records(torch.from_numpy(rec), torch.from_numpy(cnt), world)
    out = unpack_records(all_rec, all_cnt)
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
    rec, cnt = pack_records_host(dets, 4)
    back = unpack_records(rec, cnt)
    for d, b in zip(dets, back):
        k = min(len(d[1]), 4)
        assert (b[0] == d[0][:k]).all() and (b[1] == d[1][:k]).all() and (b[2] == d[2][:k]).all()
        assert b[4] == (len(d[1]) > 4)


def test_all_gather_two_ranks_gloo():
    world, n_frames, kmax = 2, 8, 16
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
