"""Two ranks, ONE MI355X: the data-parallel path with the HIP model as the gradient producer.

RCCL refuses two ranks on one device, so the process group here is gloo over the GPU tensors -- everything else is the production path:
`painter_amd.parallel.init_distributed`, `broadcast_parameters` on replicas that were seeded differently (main_train.py:190 seeds
seed + rank), `model.grad_sync = GradSync()` with the buckets handed over from INSIDE the hand-written backward (side stream, reverse
block order, in-place all-reduce of the weight matrices, flattened small tensors), two accumulation micro-steps as
engine_train.py:85-90 runs them.  Every rank must end with the same gradients, and they must be the mean over ranks of what each
rank's samples give without any exchange (computed by rank 0 alone afterwards)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ACCUM = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(cfg, seed):
    from functools import partial

    import torch.nn as nn

    from oracle import painter_oracle as O
    from painter_amd import models_painter
    m = models_painter.Painter(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
                               num_heads=cfg.num_heads, drop_path_rate=0.0, window_size=14, qkv_bias=True, mlp_ratio=4,
                               norm_layer=partial(nn.LayerNorm, eps=1e-6), window_block_indexes=([0, 1], [3, 4]), residual_block_indexes=[],
                               use_rel_pos=True, out_feature="last_feat", decoder_embed_dim=cfg.decoder_embed_dim, loss_func="smoothl1",
                               compute_dtype="fp32")
    m.load_state_dict(O.random_params(cfg, seed), strict=True)
    return m.cuda()


def _micro_steps(m, cfg, rank, params=None):
    """ACCUM forward/backward passes on this rank's samples; gradients accumulate in p.grad (loss / ACCUM, engine_train.py:83)."""
    from oracle import painter_oracle as O
    params = params if params is not None else m
    for p in params.parameters():
        p.grad = None
    for k in range(ACCUM):
        imgs, tgts, mask, valid = O.synthetic_batch(cfg, 1, 300 + 10 * rank + k, "random")
        loss, _, _ = m(imgs.cuda(), tgts.cuda(), bool_masked_pos=mask.reshape(1, *cfg.grid).cuda(), valid=valid.cuda())
        (loss / ACCUM).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in params.named_parameters()}


def _worker(rank, world, port, q, wrapper=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from oracle import painter_oracle as O
    from painter_amd import parallel
    torch.cuda.set_device(0)
    r, _, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = O.small_config()
    m = _build(cfg, 40 + rank)                                  # replicas start different, as the reference's seeding makes them
    if wrapper:                                                 # main_train.py:340: the wrapper broadcasts rank 0's parameters and owns the exchange
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=False)
    else:
        parallel.broadcast_parameters(m)
        m.grad_sync = parallel.GradSync()
    ref_sd = O.random_params(cfg, 40)
    same_as_rank0 = all(torch.equal(p.detach().cpu(), ref_sd[n]) for n, p in m.named_parameters())
    synced = _micro_steps(ddp if wrapper else m, cfg, rank, params=m)
    # every rank publishes a digest of what it ended with; rank 0 also recomputes both ranks' local gradients without any exchange
    digest = {n: (float(g.double().sum()), float(g.double().abs().sum())) for n, g in synced.items()}
    worst = None
    if rank == 0:
        m.grad_sync = None
        local = [_micro_steps(m, cfg, rk) for rk in range(world)]
        worst = 0.0
        for n, g in synced.items():
            mean = sum(loc[n] for loc in local) / world
            worst = max(worst, float((g - mean).abs().max() / mean.abs().max().clamp_min(1e-12)))
    q.put((rank, same_as_rank0, digest, worst))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wrapper", [False, True], ids=["GradSync", "DistributedDataParallel"])
def test_two_ranks_on_one_gpu_average_the_hip_models_gradients(wrapper):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, q, wrapper)) for rk in range(world)]
    for p in procs:
        p.start()
    res = {}
    import queue
    import time
    deadline = time.time() + 420
    while len(res) < world:
        try:
            rank, same, digest, worst = q.get(timeout=5)
            res[rank] = (same, digest, worst)
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, "a rank died (exit codes %s): see its traceback above" % dead      # fail in seconds, not after the full timeout
            assert time.time() < deadline, "ranks did not finish in time"
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][0] and res[1][0], "broadcast_parameters left the replicas different"
    assert res[0][1] == res[1][1], "ranks ended with different gradients"
    assert res[0][2] is not None and res[0][2] < 2e-6, res[0][2]          # = mean over ranks of the local gradients (fp32 sum of two terms)


def test_bench_multi_rank_branch_runs_with_two_gloo_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --backend gloo`: the N > 1 branch of the benchmark itself -- self-relaunch under torch.distributed.run,
    `init_distributed`, differently seeded replicas + `broadcast_parameters`, `GradSync` fed from inside the backward, the barrier-
    bracketed region, the MAX-over-ranks time, rank 0's single JSON line -- with two ranks sharing this box's one MI355X (gloo, because
    RCCL refuses two ranks per device).  The line must say it is the debug arrangement; value = 2 ranks x batch x steps / time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--batch", "1", "--min-seconds", "0", "--profile-steps", "0", "--no-cpu-baseline", "--no-optimizer", "--eval"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["config"]["global_batch"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["backend"].startswith("gloo") and out["config"]["rccl_ranks"] == 0
    assert out["value"] > 0 and abs(out["value"] - 2 * 1 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-2 * out["value"]
    assert out["loss"] == out["loss"] and out["build"]["lib_sha16"]
    # round 5: the N > 1 branch times three gradient-exchange arrangements back to back (per-block GradSync, four coarse coalesced
    # launches, the plain DDP wrapper); all three must have run, agree on the loss of their last step (same parameters, same batch, no
    # optimizer, eval mode so that DropPath draws nothing: 1e-5 -- the arrangements only differ in HOW the same averaged gradients travel) and `value` must be the best of them
    # round 6: a fourth arrangement (rs_ag: every per-block message as reduce-scatter + all-gather), and every arrangement carries its
    # exposed communication = its step time minus the step time of the same process with no exchange at all
    gv = out["extra"]["gradsync_variants"]
    assert set(gv) == {"per_block", "coarse", "rs_ag", "ddp"} and out["extra"]["gradsync_chosen"] in gv, (sorted(gv), out["extra"].get("gradsync_errors"))
    assert gv["rs_ag"]["collective_launches_per_step"] > gv["per_block"]["collective_launches_per_step"] > gv["coarse"]["collective_launches_per_step"] >= 1
    assert out["extra"]["no_exchange_ms_per_step"] > 0 and all("exposed_comm_ms" in v for v in gv.values())
    assert abs(out["extra"]["exposed_comm_ms"] - gv[out["extra"]["gradsync_chosen"]]["exposed_comm_ms"]) < 1e-9
    assert abs(out["value"] - max(v["value"] for v in gv.values())) < 1e-6 * out["value"]
    ls = [v["loss"] for v in gv.values()]
    assert max(ls) - min(ls) <= 1e-5 * abs(ls[0]), ls
