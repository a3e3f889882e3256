"""world_size-2 gloo tests of the data-parallel path: GradSync must deliver, on every rank, the
average of the per-rank gradients (== the gradient of the global batch for a mean-reduced loss), with buckets handed
over out of order, and the skip-on-accumulation switch must leave gradients local."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from painter_amd import parallel
    r, _, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    names = ["decoder_embed.weight", "blocks.1.attn.qkv.weight", "blocks.1.attn.rel_pos_h", "blocks.0.mlp.fc1.bias", "pos_embed"]
    shapes = [(16, 8), (12, 4), (5, 4), (7,), (1, 3, 4)]
    G = {n: torch.randn(s, generator=g) for n, s in zip(names, shapes)}
    local = {n: t.clone() for n, t in G.items()}
    sync = parallel.GradSync()
    sync.BIG = 40                       # (16, 8) and (12, 4) take the in-place path, the rest the flattened small-tensor message
    sync.ready(G, names[:1])            # decoder bucket first, as the engine's backward does
    sync.ready(G, names[1:3])
    sync.ready(G, names[3:])
    sync.finish()
    # reference: average over ranks, rebuilt from the seeds
    ok = True
    for n, s in zip(names, shapes):
        ref = torch.zeros(s)
        for rk in range(world):
            gg = torch.Generator().manual_seed(100 + rk)
            vals = {nn: torch.randn(ss, generator=gg) for nn, ss in zip(names, shapes)}
            ref += vals[n]
        ref /= world
        ok &= bool(torch.allclose(G[n], ref, atol=1e-6)) and G[n].shape == torch.Size(s)
    # accumulation micro-step: no exchange, gradients stay local
    G2 = {n: t.clone() for n, t in local.items()}
    sync.set_sync(False)
    sync.ready(G2, names)
    sync.finish()
    ok &= all(torch.equal(G2[n], local[n]) for n in names)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(ok for _, ok in res), res
