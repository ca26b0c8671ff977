"""world_size-2 gloo test of the per-clip sharding + result gather (the N>1 path of bench.py), CPU only."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from crab_amd.parallel import gather_results, shard_clips
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, n_new, V = 3, 5, 7
    clip0 = rank * B
    ids = torch.stack([torch.arange(n_new) + 100 * (clip0 + i) for i in range(B)])
    logits = torch.stack([torch.full((V,), float(clip0 + i)) for i in range(B)])
    res = gather_results(ids, clip0, world, rank, logits)
    if rank == 0:
        cid, allids, lg = res
        q.put((cid.tolist(), allids.tolist(), lg[:, 0].tolist(), shard_clips(7, world, 1)))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    cid, ids, lg0, shard = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert cid == list(range(6))
    assert ids == [[100 * c + j for j in range(5)] for c in range(6)]
    assert lg0 == [float(c) for c in range(6)]
    assert shard == [1, 3, 5]


def test_single_rank_passthrough():
    from crab_amd.parallel import gather_results
    ids = torch.arange(6).view(2, 3)
    cid, out, lg = gather_results(ids, 4, 1, 0)
    assert cid.tolist() == [4, 5] and torch.equal(out, ids) and lg is None
