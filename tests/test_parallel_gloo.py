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


def _worker_uneven(rank, world, port, q):
    import torch.distributed as dist
    from crab_amd.parallel import block_of, gather_results
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for n_total in (7, 2):                          # 7 clips over 3 ranks: 3 + 2 + 2; 2 clips over 3 ranks: the last rank holds none
        clip0, B = block_of(n_total, world, rank)
        n_new, V = 4, 5
        # a rank WITHOUT clips knows neither n_new nor V and has no logits to pass: it still issues the same collectives
        ids = torch.stack([torch.arange(n_new) + 100 * (clip0 + i) for i in range(B)]) if B else torch.empty((0, 0), dtype=torch.int64)
        logits = torch.stack([torch.full((V,), float(clip0 + i)) for i in range(B)]) if B else None
        res = gather_results(ids, clip0, world, rank, logits)
        if rank == 0:
            out.append((res[0].tolist(), res[1].tolist(), res[2][:, 0].tolist()))
        else:
            assert res is None
    # ranks that disagree on the record (one passes a longer id row, one drops its logits) must ALL raise before the first gather - no hang
    raised = []
    for bad in ("n_new", "logits"):
        ids = torch.zeros((2, 5 if (bad == "n_new" and rank == 1) else 4), dtype=torch.int64)
        logits = None if (bad == "logits" and rank == 2) else torch.zeros((2, 5))
        try:
            gather_results(ids, 2 * rank, world, rank, logits)
            raised.append(False)
        except ValueError:
            raised.append(True)
    assert raised == [True, True], (rank, raised)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_uneven_shards_world3_gloo():
    """A clip count the world size does not divide (bench.py --strong, an eval set of any size): every clip arrives exactly once, in order,
    the padding rows never reach the caller, and a rank without clips takes part in the collective."""
    from crab_amd.parallel import block_of
    assert [block_of(7, 3, r) for r in range(3)] == [(0, 3), (3, 2), (5, 2)] and [block_of(2, 3, r) for r in range(3)] == [(0, 1), (1, 1), (2, 0)]
    assert sum(block_of(1000, 8, r)[1] for r in range(8)) == 1000
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for n_total, (cid, ids, lg0) in zip((7, 2), out):
        assert cid == list(range(n_total))
        assert ids == [[100 * c + j for j in range(4)] for c in range(n_total)]
        assert lg0 == [float(c) for c in range(n_total)]


def _worker_world8(rank, world, port, q):
    import torch.distributed as dist
    from crab_amd.parallel import block_of, gather_results
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for n_total in (11, 5, 64):                     # 11 over 8: 2 2 2 1 1 1 1 1; 5 over 8: three ranks hold nothing; 64: the even case
        clip0, B = block_of(n_total, world, rank)
        n_new, V = 3, 4
        ids = torch.stack([torch.arange(n_new) + 100 * (clip0 + i) for i in range(B)]) if B else torch.empty((0, 0), dtype=torch.int64)
        logits = torch.stack([torch.full((V,), float(clip0 + i)) for i in range(B)]) if B else None
        res = gather_results(ids, clip0, world, rank, logits)
        if rank == 0:
            out.append((res[0].tolist(), res[1].tolist(), res[2][:, 0].tolist()))
        else:
            assert res is None
    recs = [None] * world
    dist.all_gather_object(recs, {"rank": rank, "clips": block_of(11, world, rank)[1]})      # what bench.py collects per rank
    if rank == 0:
        q.put((out, recs))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_world8_gloo():
    """The node's world size (BASELINE configs[3]: 8 x MI355X): eight gloo ranks, clip counts the world does not divide (incl. ranks without any
    clip), the per-rank records gathered as bench.py gathers them - so that the first 8-GPU run is not the first 8-rank run of this code."""
    from crab_amd.parallel import block_of
    assert [block_of(11, 8, r)[1] for r in range(8)] == [2, 2, 2, 1, 1, 1, 1, 1] and [block_of(5, 8, r)[1] for r in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_world8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, recs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for n_total, (cid, ids, lg0) in zip((11, 5, 64), out):
        assert cid == list(range(n_total))
        assert ids == [[100 * c + j for j in range(3)] for c in range(n_total)]
        assert lg0 == [float(c) for c in range(n_total)]
    assert [r["rank"] for r in recs] == list(range(8)) and [r["clips"] for r in recs] == [2, 2, 2, 1, 1, 1, 1, 1]


def test_single_rank_passthrough():
    from crab_amd.parallel import gather_results
    ids = torch.arange(6).view(2, 3)
    cid, out, lg = gather_results(ids, 4, 1, 0)
    assert cid.tolist() == [4, 5] and torch.equal(out, ids) and lg is None
