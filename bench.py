#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): images/sec of the Painter ViT-L forward+backward on 896x448 stitched pairs.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = one pass of the hot path over one batch: forward + backward of
painter_vit_large_patch16_input896x448 (train mode: DropPath active), per-GPU batch 8, bf16 operands / fp32 accumulate,
synthetic inputs already resident in HBM; for N > 1 the gradient all-reduce (RCCL, bucketed, overlapped with backward)
is inside the step.  Prints ONE JSON line on rank 0 (contract in the task statement):
  value       = N * B * K / t      images/sec, t = max over ranks of the barrier-bracketed wall time of exactly K steps
  roofline    = the dominant kernel family's largest forward instantiation (256x256 bf16 MFMA GEMM, Mlp.fc1 + GELU) measured
                live with HIP events on the launch stream over the timed region: achieved TFLOP/s = algorithmic
                FLOPs (2.M.N.K per launch) of those launches / their summed duration; peak = 2500 TFLOP/s dense bf16 MFMA;
                `traffic` = HBM bytes per launch from the PMC pass (profiles/roofline_traffic.json).  `other_kernels`
                carries the same measurement for the attention forward and the (side-stream, overlapped) weight gradient.
                `model_mfma_frac` = images/s/GPU * 4.034 TFLOP (attention+MLP fwd+bwd) / 2.5 PFLOP/s.
  cpu_baseline = the CPU oracle (oracle/painter_oracle.py, kind "port"; /root/reference does not exist on the GPU box)
                timed on the host cores: ONE forward+backward at B=1, 896x448, fp32 (about 10-30 s).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_BLOCKS_FWD_BWD = 4.034e12     # attention + MLP blocks, fwd+bwd, per image
FLOP_MODEL_FWD_BWD = 4.769e12      # whole model
PEAK_BF16_TFLOPS = 2500.0          # dense MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def randomize_parameters(model, seed):
    """Reference init leaves rel_pos / biases / LN affine at 0/1; zero operands also clock higher
    (DVFS), so every parameter is re-drawn.  Host RNG + copy = plumbing."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            r = torch.randn(p.shape, generator=g)
            if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n in ("norm.weight", "decoder_pred.1.weight"):
                p.copy_(1.0 + 0.1 * r)
            elif n.endswith("rel_pos_h") or n.endswith("rel_pos_w"):
                p.copy_(0.05 * r)
            elif p.ndim == 4 and p.shape[-1] > 1:
                p.copy_(r / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5)
            else:
                p.copy_(0.02 * r)


def synthetic_inputs(batch, H, W, L, seed, device):
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor([0.485, 0.456, 0.406])[None, :, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[None, :, None, None]
    imgs = ((torch.rand(batch, 3, H, W, generator=g) - mean) / std).to(device)
    tgts = ((torch.rand(batch, 3, H, W, generator=g) - mean) / std).to(device)
    mask = torch.zeros(batch, L, dtype=torch.bool)
    mask[:, L // 2:] = True                        # the reference's inference / half_mask case (pairdataset.py:183-186)
    valid = torch.ones(batch, 3, H, W, device=device)
    return imgs, tgts, mask.to(device), valid


class KernelTimer:
    """HIP-event brackets around every launch of a few ops inside the timed region.  Events are recorded on the stream the
    op is launched on (torch's current stream at the call: the caller's stream, or the engine's side stream for weight
    gradients).  -> per op: launches, summed duration, algorithmic FLOPs."""

    FLOPS = {
        "fc1": lambda a, k: 2.0 * a[0].shape[0] * a[0].shape[1] * a[1].shape[0],                    # linear_gelu(x, w, bias)
        "attn_fwd": lambda a, k: 4.0 * a[2] * a[4] * a[3] * a[3] * 64,                               # (qkv, rcat, batch, L, heads, ...)
        "wgrad": lambda a, k: 2.0 * a[0].shape[0] * a[0].shape[1] * a[1].shape[1],                   # linear_wgrad(dy, x)
    }
    TARGET = {"fc1": "linear_gelu", "attn_fwd": "attn_fwd", "wgrad": "linear_wgrad"}

    def __init__(self, ops_mod):
        self.ops, self.active = ops_mod, False
        self.rec = {k: {"events": [], "flops": 0.0} for k in self.TARGET}

    def install(self):
        for key, fname in self.TARGET.items():
            orig = getattr(self.ops, fname)

            def wrapped(*a, _orig=orig, _key=key, **kw):
                if not self.active:
                    return _orig(*a, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _orig(*a, **kw)
                e1.record()
                self.rec[_key]["events"].append((e0, e1))
                self.rec[_key]["flops"] += self.FLOPS[_key](a, kw)
                return r

            setattr(self.ops, fname, wrapped)

    def result(self, key):
        ev = self.rec[key]["events"]
        ms = sum(a.elapsed_time(b) for a, b in ev)
        return len(ev), ms, (self.rec[key]["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)


def cpu_baseline():
    """One fp32 forward+backward of the CPU oracle at B=1 (ViT-L, 896x448) on the host cores."""
    from oracle import painter_oracle as O
    torch.set_num_threads(min(os.cpu_count(), 32))     # more threads only add contention at B=1
    cfg = O.vit_large_config()
    P = {k: v.requires_grad_(True) for k, v in O.random_params(cfg, 1).items()}
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 1, 1234, "half")
    t0 = time.time()
    loss, _, _ = O.forward(P, cfg, imgs, tgts, mask, valid)
    loss.backward()
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 5), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/painter_oracle.py (CPU restatement of the reference, PyTorch fp32), ViT-L 896x448, B=1, "
                      "one forward+backward, %.1f s, no warm-up" % dt}


def optimizer_step_ms(model, step_fn):
    """Reported next to the headline number, never inside it ("+ optimizer step reported separately"):
    the fused unscale + clip(3.0) + AdamW of painter_amd/optim.py (52 layer-decay groups as util/lr_decay.py builds them) vs
    torch.optim.AdamW + clip_grad_norm_ on the same gradients, HIP events, 3 runs each."""
    from painter_amd import optim as PO
    nl = len(model.blocks) + 1
    groups = {}
    for n, p in model.named_parameters():                       # util/lr_decay.py:15-76
        lid = 0 if (n in ("cls_token", "pos_embed") or n.startswith("patch_embed")) else (int(n.split(".")[1]) + 1 if n.startswith("blocks") else nl)
        nd = p.ndim == 1 or n in ("pos_embed", "cls_token")
        key = (lid, nd)
        groups.setdefault(key, {"params": [], "weight_decay": 0.0 if nd else 0.05, "lr": 1e-3 * 0.8 ** (nl - lid)})["params"].append(p)
    step_fn()                                                    # fresh gradients
    res = {}
    for name, opt in (("fused", PO.AdamW(list(groups.values()), lr=1e-3, betas=(0.9, 0.999))),
                      ("torch", torch.optim.AdamW(list(groups.values()), lr=1e-3, betas=(0.9, 0.999)))):
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == "fused":
                opt.grad_sumsq()
                opt.step(max_norm=3.0)
            else:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0)
                opt.step()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[name + "_ms"] = round(min(ts[1:]), 3)
        del opt
    res["note"] = "unscale + global-norm clip + AdamW over 370.7 M fp32 parameters; not part of `value`"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE configs[1]: 8)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--roofline-kernel", default="fc1", choices=["fc1", "attn_fwd", "wgrad"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval", action="store_true", help="eval mode (no DropPath)")
    args = ap.parse_args()

    from painter_amd import models_painter, ops, parallel
    rank, local, world = parallel.init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    model = models_painter.painter_vit_large_patch16_input896x448(compute_dtype=args.dtype)
    randomize_parameters(model, seed=1)
    model = model.to(dev)
    model.train(not args.eval)
    if world > 1 or parallel._SELFTEST:
        import torch.distributed as dist
        for p in model.parameters():                      # C2: replicas start identical (same seed; broadcast anyway)
            dist.broadcast(p.data, src=0)
        model.grad_sync = parallel.GradSync()
    cfg = model._cfg
    imgs, tgts, mask, valid = synthetic_inputs(args.batch, cfg.H, cfg.W, cfg.L, 1234 + rank, dev)

    def step():
        for p in model.parameters():
            p.grad = None
        loss, _, _ = model(imgs, tgts, bool_masked_pos=mask, valid=valid)
        loss.backward()
        return loss

    timer = KernelTimer(ops)
    timer.install()
    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1 or parallel._SELFTEST:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    timer.active = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    timer.active = False
    lossv = float(loss.item())
    if world > 1 or parallel._SELFTEST:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ips = world * args.batch * args.steps / dt
        n, kms, tf = timer.result(args.roofline_kernel)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        KNAME = {
            "fc1": "g256::gemm256_kernel<false,false,Epi4BiasGelu> (Mlp.fc1 forward: [R,1024]x[1024,4096] + bias + erf-GELU, writes "
                   "pre-activation and activation; the largest forward instantiation of the 256x256 bf16 MFMA GEMM that carries "
                   "52% of the step's kernel time; forward launches do not overlap the side stream, so the event time is the "
                   "kernel's own)",
            "attn_fwd": "a2::fwd_kernel (fused attention forward with decomposed rel-pos bias)",
            "wgrad": "g256::gemm256_kernel<true,true,Epi4Slab> (weight gradient dW = dY^T.X on the side stream: runs CONCURRENTLY "
                     "with the main stream's dgrad/attention kernels, so its duration includes time-sharing)",
        }
        kname = KNAME[args.roofline_kernel]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")      # PMC pass (tools/pmc_traffic.py), bytes per launch
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.roofline_kernel, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        more = {}
        for k in KernelTimer.TARGET:
            if k != args.roofline_kernel:
                nn_, ms_, tf_ = timer.result(k)
                more[k] = {"launches": nn_, "kernel_ms_total": round(ms_, 3), "achieved": round(tf_, 2), "frac": round(tf_ / peak, 4)}
        out = {
            "metric": "images/sec (896x448 pairs) ViT-L fwd+bwd",
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "painter_vit_large_patch16_input896x448 %s batch=%d/GPU fwd+bwd on %dxMI355X (BASELINE configs[1])"
                                   % (args.dtype, args.batch, world),
                       "global_batch": world * args.batch, "image": "896x448x3 stitched pair", "tokens": cfg.L,
                       "mode": "eval" if args.eval else "train (DropPath 0.1)", "parallelism": "dp%d" % world,
                       "grad_allreduce": "RCCL bucketed, overlapped with backward" if world > 1 else "n/a"},
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                         "traffic": traffic, "kernel": kname, "launches": n, "kernel_ms_total": round(kms, 3),
                         "avg_us": round(kms / max(n, 1) * 1e3, 2), "other_kernels": more},
            "model_mfma_frac": round(ips / world * FLOP_BLOCKS_FWD_BWD / (peak * 1e12), 4),
            "model_tflops_per_gpu": round(ips / world * FLOP_MODEL_FWD_BWD / 1e12, 2),
            "loss": round(lossv, 6),
        }
        if world == 1 and args.dtype == "bf16":
            out["optimizer_step"] = optimizer_step_ms(model, step)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1 or parallel._SELFTEST:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
