"""world_size-2 gloo tests of the data-parallel path.

1. GradSync must deliver, on every rank, the average of the per-rank gradients, with buckets handed over out of order (in-place
   path for big tensors, flattened message for small ones).
2. The same with REAL model gradients: each rank differentiates the tiny Painter (the CPU oracle's autograd is the producer, the HIP
   path needs a GPU) on its own sample, hands the gradients to GradSync bucket by bucket in the engine's order (decoder first, blocks
   last to first, token/patch parameters), accumulates two micro-steps as engine_train.py:85-90 does, and must end up with the
   gradient of the global batch -- bit-equal across ranks, equal to the mean of the per-rank gradients to fp32 rounding and to the
   single-process big-batch gradient up to the loss's `+ 1e-2` denominator term (models_painter.py:462).
3. broadcast_parameters makes replicas that were seeded differently identical (main_train.py:190 seeds seed + rank)."""
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


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from painter_amd import parallel
    r, _, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    return parallel


def _worker(rank, world, port, q, mode="per_block"):
    parallel = _init(rank, world, port)
    g = torch.Generator().manual_seed(100 + rank)
    names = ["decoder_embed.weight", "blocks.1.attn.qkv.weight", "blocks.1.attn.rel_pos_h", "blocks.0.mlp.fc1.bias", "pos_embed"]
    shapes = [(16, 8), (12, 4), (5, 4), (7,), (1, 3, 4)]
    G = {n: torch.randn(s, generator=g) for n, s in zip(names, shapes)}
    sync = parallel.GradSync(mode=mode, bucket_bytes=600)      # coarse: 600 bytes -> the first bucket closes inside the second ready(), the rest at finish()
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
    ok &= not hasattr(sync, "set_sync")          # there is no way to skip a micro-step's exchange (replicas would diverge)
    # per_block: 2 matrices in place + 2 flattened messages; coarse: 2 coalesced launches; rs_ag: the two matrices and the 20-element message
    # as reduce-scatter + all-gather pairs, the 19-element message (2 does not divide it) as one all-reduce
    ok &= sync.launches == {"coarse": 2, "per_block": 4, "rs_ag": 7}[mode]
    # the engine's per-block arrangement: the small gradients are views of ONE pre-allocated flat buffer, exchanged in place as one
    # message (no flattening copy); the big matrix goes in place as before
    g2 = torch.Generator().manual_seed(500 + rank)
    flat = torch.randn(7 + 20, generator=g2)
    big = torch.randn(16, 8, generator=g2)
    stray = torch.randn(6, 4, generator=g2)          # a small tensor that is NOT a view of flat (a down-sized model's weight matrix): flattening path
    G2 = {"blocks.3.mlp.fc1.bias": flat[:7], "blocks.3.attn.rel_pos_h": flat[7:].view(5, 4), "blocks.3.mlp.fc1.weight": big,
          "blocks.3.attn.proj.weight": stray}
    ptrs = {n: t.data_ptr() for n, t in G2.items()}
    sync.ready(G2, list(G2), flat=flat)
    sync.finish()
    ref_flat, ref_big, ref_stray = torch.zeros(27), torch.zeros(16, 8), torch.zeros(6, 4)
    for rk in range(world):
        gg = torch.Generator().manual_seed(500 + rk)
        ref_flat += torch.randn(27, generator=gg)
        ref_big += torch.randn(16, 8, generator=gg)
        ref_stray += torch.randn(6, 4, generator=gg)
    ok &= bool(torch.allclose(flat, ref_flat / world, atol=1e-6)) and bool(torch.allclose(big, ref_big / world, atol=1e-6))
    ok &= bool(torch.allclose(G2["blocks.3.attn.proj.weight"], ref_stray / world, atol=1e-6))
    ok &= all(G2[n].data_ptr() == ptrs[n] for n in G2 if n != "blocks.3.attn.proj.weight")   # still the same views: nothing was copied or re-pointed
    try:
        sync.ready({"a": torch.zeros(3)}, ["a"], flat=torch.zeros(3))
        ok = False                                               # a flat buffer that does not hold the gradient must be refused
    except AssertionError:
        pass
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def _bucket_order(names, depth):
    """The order in which engine.HotPath.backward hands gradients to GradSync.ready()."""
    order = [["decoder_embed.weight", "decoder_embed.bias"], [n for n in names if n.startswith("decoder_pred.")]]
    for i in reversed(range(depth)):
        order.append([n for n in names if n.startswith("blocks.%d." % i)])
    order.append(["norm.weight", "norm.bias", "patch_embed.proj.weight", "patch_embed.proj.bias", "pos_embed", "segment_token_x",
                  "segment_token_y", "mask_token"])
    assert sorted(sum(order, [])) == sorted(names)
    return order


def _model_grads(P0, cfg, seeds):
    """Gradient dict of the tiny Painter oracle on the batch made of the samples `seeds` (eval mode)."""
    from oracle import painter_oracle as O
    P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    parts = [O.synthetic_batch(cfg, 1, s, "random") for s in seeds]
    imgs, tgts, mask, valid = [torch.cat([p_[i] for p_ in parts]) for i in range(4)]
    loss, _, _ = O.forward(P, cfg, imgs, tgts, mask, valid)
    loss.backward()
    return {k: v.grad.clone() for k, v in P.items()}


def _worker_model(rank, world, port, q, mode="per_block"):
    parallel = _init(rank, world, port)
    from oracle import painter_oracle as O
    torch.set_num_threads(2)
    cfg = O.tiny_config()
    # replicas start from different seeds (main_train.py:190) and are made identical by the broadcast
    P0 = O.random_params(cfg, 50 + rank)
    holder = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v) for k, v in P0.items()})
    versions = [p._version for p in holder.values()]
    parallel.broadcast_parameters(holder)
    # the broadcast must bump every parameter's version counter: the engine's cached bf16 operand copies are keyed on it (a replica
    # that ran a forward before the broadcast would otherwise keep multiplying by its old weights)
    assert all(p._version > v for p, v in zip(holder.values(), versions))
    P0 = {k.replace("/", "."): v.detach() for k, v in holder.items()}
    ref0 = O.random_params(cfg, 50)
    same = all(torch.equal(P0[k], ref0[k]) for k in ref0)
    sync = parallel.GradSync(mode=mode, bucket_bytes=1 << 20)   # coarse: a handful of coalesced launches per micro-step for the tiny model
    sync.BIG = 4096
    names = list(P0.keys())
    acc = {n: torch.zeros_like(t) for n, t in P0.items()}
    micro = [[10 + rank], [20 + rank]]                       # two accumulation micro-steps, one sample per rank each
    for seeds in micro:
        G = _model_grads(P0, cfg, seeds)
        for bucket in _bucket_order(names, cfg.depth):
            sync.ready(G, bucket)
        sync.finish()
        for n in names:
            acc[n] += G[n]                                   # what autograd does with the returned gradients
    q.put((rank, same, {n: t.numpy().copy() for n, t in acc.items()}))     # by value (tensors would travel as shm handles)
    dist.barrier()
    dist.destroy_process_group()


def _run(target, world=2, mode="per_block"):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    return sorted(res, key=lambda r: r[0])


import pytest  # noqa: E402


@pytest.mark.parametrize("mode", ["per_block", "coarse", "rs_ag"])
def test_gradsync_world2_gloo(mode):
    """mode "coarse" (round 5): the same gradients, collected and exchanged as a few coalesced launches -- the same averages.
    mode "rs_ag" (round 6): every message as reduce-scatter + all-gather (gloo: W reduce() calls stand in for the reduce-scatter)."""
    res = _run(_worker, mode=mode)
    assert [r for r, _ in res] == [0, 1]
    assert all(ok for _, ok in res), res


@pytest.mark.parametrize("mode", ["per_block", "coarse", "rs_ag"])
def test_gradsync_world2_model_gradients_with_accumulation(mode):
    from oracle import painter_oracle as O
    res = _run(_worker_model, mode=mode)
    assert all(same for _, same, _ in res), "broadcast_parameters did not make the replicas identical"
    g0, g1 = ({n: torch.from_numpy(a) for n, a in res[r][2].items()} for r in range(2))
    cfg = O.tiny_config()
    P0 = O.random_params(cfg, 50)
    per_rank = [[_model_grads(P0, cfg, [10 + r]), _model_grads(P0, cfg, [20 + r])] for r in range(2)]
    big = [_model_grads(P0, cfg, [10, 11]), _model_grads(P0, cfg, [20, 21])]
    worst_mean, worst_big = 0.0, 0.0
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n                  # replicas hold bit-identical accumulated gradients
        mean = sum((per_rank[0][m][n] + per_rank[1][m][n]) * 0.5 for m in range(2))
        bigb = big[0][n] + big[1][n]
        den = mean.abs().max().clamp_min(1e-12)
        worst_mean = max(worst_mean, float((g0[n] - mean).abs().max() / den))
        worst_big = max(worst_big, float((g0[n] - bigb).abs().max() / bigb.abs().max().clamp_min(1e-12)))
    assert worst_mean < 1e-6, worst_mean
    assert worst_big < 1e-4, worst_big                       # only the `+ 1e-2` in the loss denominator separates the two


def test_bench_refuses_a_line_for_fewer_gpus_than_asked(tmp_path):
    """bench.py --gpus N (N > 1): without a launcher and without N devices it exits with a message instead of running on one GPU; under a
    launcher it refuses a WORLD_SIZE that differs from --gpus.  Either way no JSON line is printed."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout) and "{" not in r.stdout
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout) and "{" not in r.stdout
