#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): images/sec of the Painter ViT-L forward+backward on 896x448 stitched pairs.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started without a launcher (WORLD_SIZE unset) the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (what the reference's
train_painter_vit_large.sh:5-8 does); started by a launcher it checks WORLD_SIZE against --gpus and refuses a mismatch -- it never
prints a line for fewer GPUs than asked for.

A "step" = one pass of the hot path over one batch: forward + backward of painter_vit_large_patch16_input896x448 (train mode:
DropPath active), per-GPU batch 8, bf16 operands / fp32 accumulate, synthetic inputs already resident in HBM; for N > 1 the gradient
all-reduce (RCCL, bucketed, overlapped with backward) is inside the step.  Prints ONE JSON line on rank 0:
  value        = N * B * K / t     images/sec, t = max over ranks of the barrier-bracketed wall time of exactly K steps.  Nothing
                 is instrumented inside the timed region.
  roofline     = SURVEY.md 8(d): achieved = images/s/GPU * 4.034 TFLOP (attention + MLP blocks, fwd+bwd), peak =
                 2500 TFLOP/s dense bf16 MFMA, frac = achieved / peak (whole-model accounting, 4.769 TFLOP, beside it).
                 `dominant_kernel` = the kernel family with the largest summed duration in a SEPARATE profiled pass after the timed
                 region (HIP events around every launch on the launching stream; second stream off, so a duration is the kernel's
                 own): algorithmic FLOPs of its launches / their summed duration, against the same peak; `kernels` lists every family.
                 `traffic` = HBM bytes per launch of the dominant family's largest forward instantiation from the PMC passes
                 (profiles/roofline_traffic.json, tools/pmc_traffic.py) or null.
  sustained    = a second timed region right after the first: the same step repeated until --min-seconds (default 5 s) have elapsed
                 -- clocks and temperatures at steady state; `value` stays the contract's EXACTLY-K-steps number.
  extra        = sclk_mhz_mean / power_w_mean: shader clock and board power sampled (amdgpu hwmon, 100 ms) during the sustained region --
                 the step runs clock-managed (~2.0 GHz), so two boxes' lines are comparable only with these; the
                 roofline peak stays the contract's 2500 TFLOP/s.
  cpu_baseline = the unmodified reference class (kind "reference": Painter/models_painter.py through oracle/ref_import.py -- from
                 /root/reference, or on the GPU box from the subset oracle/stage_ref.py staged at build time; kind "port" = the restated
                 oracle/painter_oracle.py when neither exists) on the host cores, B = 1, fp32, thread count swept over {16, 32, 64, all
                 physical cores}: value = train forward+backward images/sec at the best count, the all-cores figure beside it.
  secondary    = the other single-GPU BASELINE configurations, measured AFTER the headline number and outside its timed region (N = 1):
                 seggpt_n32 = BASELINE configs[3] (seggpt_vit_large_patch16_input896x448, 32 prompts over one query, feature ensemble,
                 forward captured in a hipGraph, 5 replays), vit_huge = configs[4]'s per-GPU half (ViT-H/14 bf16, B = 4, 3 steps).
  gradsync_variants (N > 1, under `extra`) = the same K steps timed back to back under four gradient-exchange arrangements --
                 per-block RCCL messages from inside the backward (GradSync, the default), four coarse coalesced launches (GradSync
                 mode "coarse"), the per-block messages as reduce-scatter + all-gather pairs (mode "rs_ag": the all-peer pattern over the
                 seven xGMI links), the plain DistributedDataParallel wrapper (the reference's arrangement: no overlap with OUR backward,
                 the floor) -- `value` is the best of them and `config.grad_allreduce` says which.  Each carries `exposed_comm_ms` = its
                 step time minus the step time of the same process with NO exchange (`extra.no_exchange_ms_per_step`), so the line
                 explains its own scaling efficiency; `config.rccl_ranks` is asserted equal to N.
  reference_gpu = the same unmodified reference model on THIS GPU through PyTorch-ROCm eager (autocast bf16, and fp16 -- the reference's
                 literal torch.cuda.amp.autocast() -- beside it), B = 8, the bench model's parameters and batch, train forward+backward;
                 vs_reference_gpu = value / that.  A baseline leg outside every timed region of the headline number (N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_BLOCKS_FWD_BWD = 4.034e12     # attention + MLP blocks, fwd+bwd, per image
FLOP_MODEL_FWD_BWD = 4.769e12      # whole model
# --model vit_huge (BASELINE configs[4], SURVEY.md 8d config 5: ViT-H/14, not a reference factory): blocks-only / whole model
MODELS = {
    "vit_large": dict(factory="painter_vit_large_patch16_input896x448", blocks=FLOP_BLOCKS_FWD_BWD, whole=FLOP_MODEL_FWD_BWD, head_dim=64,
                      workload="painter_vit_large_patch16_input896x448 %s batch=%d/GPU fwd+bwd on %dxMI355X (BASELINE configs[1])"),
    "vit_huge": dict(factory="painter_vit_huge_patch14_input896x448", blocks=10.76e12, whole=11.66e12, head_dim=80,
                     workload="painter_vit_huge_patch14_input896x448 %s batch=%d/GPU fwd+bwd on %dxMI355X (BASELINE configs[4]; head_dim 80 on the generation-2 attention kernels)"),
}
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


def _mnk(a_rows, a_cols, b_rows):
    return 2.0 * a_rows * a_cols * b_rows


class KernelTimer:
    """HIP-event brackets around every launch of the MFMA-bound ops, used in a profiled pass AFTER the timed region.  Events are
    recorded on the stream the op is launched on (torch's current stream at the call).  -> per family: launches, summed duration,
    algorithmic FLOPs."""

    # op -> (family, algorithmic FLOPs of one call)
    OPS = {
        "linear_fwd": ("gemm256_fwd", lambda a, k: _mnk(a[0].shape[0], a[0].shape[1], a[1].shape[0])),          # (x [M,K], w [N,K])
        "linear_gelu": ("gemm256_fwd", lambda a, k: _mnk(a[0].shape[0], a[0].shape[1], a[1].shape[0])),
        "linear_pixshuf": ("gemm256_fwd", lambda a, k: _mnk(a[0].shape[0], a[0].shape[1], a[1].shape[0])),
        "linear_dgrad": ("gemm256_dgrad", lambda a, k: _mnk(a[0].shape[0], a[0].shape[1], a[1].shape[1])),      # (dy [M,N], w [N,K])
        "linear_wgrad": ("gemm256_wgrad", lambda a, k: _mnk(a[0].shape[0], a[0].shape[1], a[1].shape[1])),      # (dy [M,N], x [M,K])
        "attn_fwd": ("attention_fwd", lambda a, k: 4.0 * a[2] * a[4] * a[3] * a[3] * a[1].shape[1]),             # (qkv, rcat [NRP, hd], batch, L, heads)
        # "recompute in flash-attention backward is not counted" -> the backward's algorithmic work is 2 x the forward's
        # (dV, dP, dQ, dK: four L x L x hd contractions); the one unavoidable S recompute makes it 2.5 x.  `achieved` / `frac` follow 8(d);
        # results() adds frac_with_recompute beside it.
        "attn_bwd_core": ("attention_bwd", lambda a, k: 8.0 * a[6] * a[8] * a[7] * a[7] * a[1].shape[1]),
    }
    NAMES = {
        "gemm256_fwd": "g256::gemm256_kernel<false,false,*> (nn.Linear forward: qkv, proj, fc1+GELU, fc2, decoder_embed)",
        "gemm256_dgrad": "g256::gemm256_kernel<false,true,*> (nn.Linear data gradient dX = dY.W)",
        "gemm256_wgrad": "g256::gemm256_kernel<true,true,Epi4Slab> + slab_reduce (weight gradient dW = dY^T.X)",
        "attention_fwd": "a3::fwd_kernel (fused attention forward, rel-pos bias on the matrix pipe; head_dim 80: a2::fwd_kernel<1,1,80>)",
        "attention_bwd": "a3::bwd_dq_kernel (Delta = rowsum(dO * O) in its prologue, rel-pos table gradient contracted inside) + a3::bwd_dkv_kernel (fused attention backward; head_dim 80: a2::bwd_dq_kernel<2,2,80,WP32> + a2::bwd_dkv_kernel<2,80> + the delta launch)",
    }

    def __init__(self, ops_mod):
        self.ops, self.active = ops_mod, False
        self.rec = {}
        self.depth = 0          # ops.linear_gelu calls the (wrapped) module-global ops.linear_fwd: only the outermost bracket counts

    def install(self):
        for fname, (family, flops) in self.OPS.items():
            orig = getattr(self.ops, fname)

            def wrapped(*a, _orig=orig, _family=family, _flops=flops, **kw):
                if not self.active or self.depth > 0:
                    return _orig(*a, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.depth += 1
                try:
                    r = _orig(*a, **kw)
                finally:
                    self.depth -= 1
                e1.record()
                ent = self.rec.setdefault(_family, {"events": [], "flops": 0.0})
                ent["events"].append((e0, e1))
                ent["flops"] += _flops(a, kw)
                return r

            setattr(self.ops, fname, wrapped)

    def results(self, peak):
        out = {}
        for fam, ent in self.rec.items():
            ms = sum(a.elapsed_time(b) for a, b in ent["events"])
            n = len(ent["events"])
            tf = ent["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out[fam] = {"kernel": self.NAMES[fam], "launches": n, "kernel_ms_total": round(ms, 3), "avg_us": round(ms / max(n, 1) * 1e3, 2),
                        "achieved": round(tf, 2), "frac": round(tf / peak, 4)}
            if fam == "attention_bwd":
                out[fam].update({"frac_8d": round(tf / peak, 4), "frac_with_recompute": round(1.25 * tf / peak, 4),
                                 "accounting": "frac = frac_8d: 2 x the forward's FLOPs (SURVEY.md 8d: recompute not counted); frac_with_recompute: 2.5 x (the one S recompute a flash backward needs)"})
        return out


# Algorithmic HBM bytes per launch of the kernels profiles/roofline_traffic.json holds (ViT-L, B = 8: R = 12544 token rows, D = 1024, hidden
# 4096, 128 (sample, head) pairs of 1568 tokens x 64; bf16 operands): every operand read once, every output written once.
def _algorithmic_bytes():
    R, D, Hd, L, BH, hd = 12544, 1024, 4096, 1568, 128, 64
    qkv = BH * L * hd * 2                                        # one of q / k / v, all heads: 25.7 MB
    tables = BH * (L // 32) * 6144                               # per-query bias tables of generation 3: 38.5 MB
    return {
        "fc1": {"bytes": R * D * 2 + Hd * D * 2 + R * Hd * 2 + R * Hd, "what": "X [R, D] + W [4D, D] in; act bf16 + gelu' 8-bit code [R, 4D] out"},
        "wgrad": {"bytes": R * Hd * 2 + R * D * 2 + 4 * Hd * D, "what": "fc1 / fc2 weight gradient: dY [R, 4D] + X [R, D] in; dW fp32 [4D, D] out (split-K slabs and their reduction are overhead, not algorithm)"},
        # the PMC family is keyed by kernel + grid: the three data gradients with a [R, D] output (qkv: K = 3D, proj: K = D, fc1: K = 4D), one of
        # each per block -- the measured figure is their average, so is this one (round 5; the proj-only figure made the ratio look like 2.6)
        "dgrad": {"bytes": (R * (3 * D + D + Hd) * 2 + (3 * D + D + Hd) * D * 2 + 3 * R * D * 2) // 3,
                  "what": "average of the qkv / proj / fc1 data gradients (one each per block, same grid): dY [R, 3D | D | 4D] + W in; dX [R, D] out"},
        "attn_fwd": {"bytes": 3 * qkv + qkv + BH * L * 4 + tables, "what": "q, k, v in; out, lse, bias tables (kept for the backward) out"},
        "attn_bwd_dq": {"bytes": 3 * qkv + qkv + tables + qkv, "what": "q, k, v, dO, tables in; dQ out (+ the [166, 64] table gradient)"},
        "attn_bwd_dkv": {"bytes": 3 * qkv + qkv + tables + 2 * qkv, "what": "q, k, v, dO, tables in; dK, dV out"},
    }


def secondary_seggpt_n32(dev, replays=5):
    """BASELINE configs[3]: seggpt_vit_large_patch16_input896x448, N = 32 prompts sharing one query (merge_between_batch = 0, seg_type ones,
    bottom-half mask as [1, L]: seggpt_engine.py:36-47), bf16, forward only, captured in a hipGraph and replayed (tools/seggpt_bench.py is
    the long form with the reference-on-GPU leg).  Parameters are drawn on the device (a secondary leg has a ~10 s budget)."""
    from painter_amd import models_seggpt
    N = 32
    m = models_seggpt.seggpt_vit_large_patch16_input896x448(compute_dtype="bf16").to(dev).eval()
    _randomize_on_device(m, 1)
    c = m._cfg
    imgs, tgts, _, valid = synthetic_inputs(N, c.H, c.W, c.L, 1234, dev)
    imgs[:, :, c.H // 2:] = imgs[:1, :, c.H // 2:]
    mask = torch.zeros((1, c.L), dtype=torch.float32, device=dev)
    mask[:, c.L // 2:] = 1
    seg_type = torch.ones((N, 1), device=dev)

    def fwd():
        with torch.no_grad():
            return m(imgs, tgts, mask, valid, seg_type, 0)
    loss, pred, _ = fwd()
    torch.cuda.synchronize()
    ref_pred = pred.clone()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        fwd()                                  # warm-up on the capture stream (per-stream workspaces)
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        _, g_pred, _ = fwd()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / replays
    res = {"value": round(N / t, 2), "unit": "images/sec", "ms_per_forward": round(t * 1e3, 3), "prompts": N, "dtype": "bf16",
           "tflops": round(N * 1.5897 / t, 1), "frac_whole_model": round(N * 1.5897 / t / PEAK_BF16_TFLOPS, 4),
           "replay_equals_eager": bool(torch.equal(g_pred, ref_pred)), "loss": round(float(loss.detach()), 6),
           "workload": "seggpt_vit_large_patch16_input896x448 inference, batch=32 in-context prompts, 1xMI355X, forward-only hipGraph replay x%d (BASELINE configs[3])" % replays}
    del graph, m
    torch.cuda.empty_cache()
    return res


def secondary_vit_huge(dev, steps=3, batch=4):
    """BASELINE configs[4], the per-GPU half: painter_vit_huge_patch14_input896x448 (class constructor at ViT-H/14 sizes), bf16, B = 4,
    train mode, forward + backward (`bench.py --model vit_huge` is the long form with the per-family table)."""
    from painter_amd import models_painter
    spec = MODELS["vit_huge"]
    m = getattr(models_painter, spec["factory"])(compute_dtype="bf16").to(dev).train()
    _randomize_on_device(m, 1)
    c = m._cfg
    imgs, tgts, mask, valid = synthetic_inputs(batch, c.H, c.W, c.L, 1234, dev)

    def step():
        for p in m.parameters():
            p.grad = None
        m._hot.relpos_stale()                   # re-pack the rel-pos tables every step, as after an optimizer update
        loss, _, _ = m(imgs, tgts, bool_masked_pos=mask, valid=valid)
        loss.backward()
        return loss
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    ips = batch / t
    res = {"value": round(ips, 2), "unit": "images/sec", "ms_per_step": round(t * 1e3, 2), "batch": batch, "steps": steps, "dtype": "bf16",
           "frac": round(ips * spec["blocks"] / 1e12 / PEAK_BF16_TFLOPS, 4), "whole_model_frac": round(ips * spec["whole"] / 1e12 / PEAK_BF16_TFLOPS, 4),
           "loss": round(float(loss.detach()), 6), "workload": spec["workload"] % ("bf16", batch, 1)}
    del m
    torch.cuda.empty_cache()
    return res


def _randomize_on_device(model, seed):
    """randomize_parameters() with the draws made on the device (the secondary legs build 0.37 / 0.63 G-parameter models inside a ~10 s
    budget; the HEADLINE model keeps the host-RNG recipe so that its line is comparable across rounds)."""
    g = torch.Generator(device=next(model.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            r = torch.randn(p.shape, generator=g, device=p.device)
            if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n in ("norm.weight", "decoder_pred.1.weight"):
                p.copy_(1.0 + 0.1 * r)
            elif n.endswith("rel_pos_h") or n.endswith("rel_pos_w"):
                p.copy_(0.05 * r)
            elif p.ndim == 4 and p.shape[-1] > 1:
                p.copy_(r / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5)
            else:
                p.copy_(0.02 * r)


def cpu_baseline(budget_s=100.0):
    """The CPU oracle at B=1 (ViT-L, 896x448, fp32) on the host's cores.  More threads are not faster at B = 1 (round 2 measured 0.050
    images/s on 128 threads where 32 threads gave 0.11), so the thread count is SWEPT: one warm-up forward+backward, then one timed
    forward+backward at each of {16, 32, 64, all physical cores}; the best count gets up to two more timed runs and the eval forwards.
    value = forward+backward images/sec at the best thread count (median of its runs); the all-cores figure is reported beside it.
    Time-boxed to about `budget_s` seconds so that the default bench run stays within minutes."""
    import statistics

    from oracle import painter_oracle as O
    from oracle import ref_import
    logical = os.cpu_count()
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or logical
    except Exception:
        phys = logical
    phys = min(logical, phys)                      # one thread per physical core at most: SMT siblings only add contention at B = 1
    cfg = O.vit_large_config()
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 1, 1234, "half")
    kind = "port"
    if ref_import.reference_available():
        # the UNMODIFIED reference class (Painter/models_painter.py:464-472 through its own factory :476-487; from /root/reference, or on
        # the GPU box from the subset oracle/stage_ref.py staged at build time), random parameters as in the port
        ref = ref_import.load_reference_painter()
        rmodel = ref.painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1()
        rmodel.load_state_dict(O.random_params(cfg, 1), strict=True)
        rmodel.train()                                 # train mode as in the timed GPU step (DropPath 0.1 through the timm 0.3.2 stand-in)
        kind = "reference"

        def fwd_eval():
            rmodel.eval()
            with torch.no_grad():
                rmodel(imgs, tgts, mask, valid.clone())
            rmodel.train()

        def fwd_bwd():
            rmodel.zero_grad(set_to_none=True)
            loss, _, _ = rmodel(imgs, tgts, mask, valid.clone())
            loss.backward()
    else:
        P = {k: v.requires_grad_(True) for k, v in O.random_params(cfg, 1).items()}

        def fwd_eval():
            with torch.no_grad():
                O.forward(P, cfg, imgs, tgts, mask, valid.clone())

        def fwd_bwd():
            for v in P.values():
                v.grad = None
            loss, _, _ = O.forward(P, cfg, imgs, tgts, mask, valid.clone())
            loss.backward()

    def once(fn):
        t0 = time.time()
        fn()
        return time.time() - t0

    t_start = time.time()
    counts = sorted({n for n in (16, 32, 64, phys) if n <= phys} | {phys})
    torch.set_num_threads(min(32, phys))
    warm = once(fwd_bwd)                           # warm-up (allocator, thread pool, oneDNN primitives)
    sweep = {}
    for n in counts:
        torch.set_num_threads(n)
        sweep[n] = [once(fwd_bwd)]
    best = min(sweep, key=lambda n: sweep[n][0])
    torch.set_num_threads(best)
    while len(sweep[best]) < 3 and time.time() - t_start + sweep[best][-1] < 0.8 * budget_s:
        sweep[best].append(once(fwd_bwd))
    tt = statistics.median(sweep[best])
    tes = [once(fwd_eval)]
    while len(tes) < 3 and time.time() - t_start + tes[-1] < budget_s:
        tes.append(once(fwd_eval))
    te = statistics.median(tes)
    return {"value": round(1.0 / tt, 5), "unit": "images/sec", "cores": best, "logical_cpus": logical, "physical_cores": phys,
            "kind": kind, "eval_forward_images_per_sec": round(1.0 / te, 5),
            "all_cores_value": round(1.0 / sweep[phys][0], 5),
            "thread_sweep_fwd_bwd_seconds": {str(n): [round(t, 2) for t in ts] for n, ts in sweep.items()},
            "sample": ("the unmodified reference Painter.forward (models_painter.py:464-472, via oracle/ref_import.py), PyTorch-CPU fp32" if kind == "reference"
                       else "oracle/painter_oracle.py (the reference forward restated op for op, PyTorch-CPU fp32)") + ", ViT-L 896x448, B=1; warm-up "
                      "forward+backward %.1f s, then one timed forward+backward per thread count %s, the best count (%d threads) re-timed "
                      "%s s, eval forward there %s s; value = 1 / median forward+backward at %d threads; all %d physical cores: %.5f images/s"
                      % (warm, counts, best, ["%.2f" % t for t in sweep[best]], ["%.2f" % t for t in tes], best, phys, 1.0 / sweep[phys][0])}


def reference_gpu_baseline(model, inputs, dev, steps=5, warmup=2, train=True):
    """The UNMODIFIED reference model on THIS GPU through PyTorch-ROCm's own kernels (rocBLAS / hipBLASLt / ATen), driven as
    Painter/engine_train.py:56-75 drives it: forward under autocast, loss.backward().  Same factory, same parameters (copied from the
    bench model: the module trees have the same names), same synthetic batch, train mode.  The reference's literal autocast dtype is
    fp16 (torch.cuda.amp.autocast() + loss scaler); bf16 -- the dtype of the measured path -- is the primary figure, fp16 is reported
    beside it.  Baseline only: nothing of it runs inside the timed region of the headline number.  None when the reference sources
    (or the subset oracle/stage_ref.py staged at build time) are not available."""
    from oracle import ref_import
    if not ref_import.reference_available():
        return None
    ref = ref_import.load_reference_painter()
    rmodel = ref.painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1()
    rmodel.load_state_dict({k: v.detach().float().cpu() for k, v in model.state_dict().items()}, strict=True)
    rmodel = rmodel.to(dev).train(train)
    imgs, tgts, mask, valid = inputs
    res = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        def step():
            rmodel.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=dt):
                loss, _, _ = rmodel(imgs, tgts, mask, valid.clone())
            loss.backward()
            return loss
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.time() - t0) / steps * 1e3
        res[name] = (ms, float(loss.detach().float()))
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    del rmodel
    torch.cuda.empty_cache()
    B = imgs.shape[0]
    return {"value": round(B / res["bf16"][0] * 1e3, 2), "unit": "images/sec", "ms_per_step": round(res["bf16"][0], 2), "dtype": "bf16 autocast",
            "fp16_autocast_value": round(B / res["fp16"][0] * 1e3, 2), "fp16_autocast_ms_per_step": round(res["fp16"][0], 2),
            "kind": "reference", "steps": steps, "warmup": warmup, "loss_bf16": round(res["bf16"][1], 6),
            "peak_memory_gib_process": round(peak_mem, 1),
            "sample": "the unmodified reference Painter (models_painter.py:464-487 via oracle/ref_import.py) on this GPU, PyTorch %s eager, "
                      "autocast + loss.backward() as engine_train.py:56-75, ViT-L 896x448, B=%d, train mode, the bench model's parameters and "
                      "batch; %d warm-up + %d timed steps, host clock around a device synchronize" % (torch.__version__, B, warmup, steps)}


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


class ClockPowerSampler:
    """Shader clock and board power from the amdgpu hwmon nodes, sampled every 100 ms on a thread (host-side file reads: nothing is
    added to the GPU's queues).  A box may expose nodes of GPUs that are not ours: the node with the highest mean power under load is
    reported."""

    def __init__(self):
        import glob
        self.nodes = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.samples, self.on, self.alive = [], False, True

    @staticmethod
    def _read(path):
        try:
            return float(open(path).read().strip())
        except (OSError, ValueError):
            return None

    def _loop(self):
        while self.alive:
            if self.on:
                row = {}
                for n in self.nodes:
                    pw = self._read(os.path.join(n, "power1_average"))
                    if pw is None:
                        pw = self._read(os.path.join(n, "power1_input"))
                    row[n] = (pw, self._read(os.path.join(n, "freq1_input")))
                self.samples.append(row)
            time.sleep(0.1)

    def start(self):
        import threading
        if self.nodes:
            threading.Thread(target=self._loop, daemon=True).start()
        self.on = True

    def stop(self):
        self.on, self.alive = False, False
        best = None
        for n in self.nodes:
            pw = [r[n][0] for r in self.samples if r[n][0] is not None]
            ck = [r[n][1] for r in self.samples if r[n][1] is not None]
            if pw and (best is None or sum(pw) / len(pw) > best[0]):
                best = (sum(pw) / len(pw), (sum(ck) / len(ck)) if ck else None)
        if best is None:
            return {"sclk_mhz_mean": None, "power_w_mean": None, "samples": len(self.samples)}
        return {"sclk_mhz_mean": None if best[1] is None else round(best[1] / 1e6, 1), "power_w_mean": round(best[0] / 1e6, 1),
                "samples": len(self.samples), "source": "amdgpu hwmon (freq1_input, power1_average), 100 ms, during the sustained region"}


def _git_head():
    """PAINTER_AMD_GIT_HEAD, else `git rev-parse` (build container), else what painter_amd/build.py recorded beside the library (the GPU
    box's snapshot has no .git)."""
    h = os.environ.get("PAINTER_AMD_GIT_HEAD")
    if h and h != "unknown":
        return h
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=10)
            return r.stdout.strip() + ("-dirty" if dirty.stdout.strip() else "")
    except Exception:
        pass
    try:
        info = json.load(open(os.path.join(ROOT, "painter_amd", "lib", "build_info.json")))
        return "%s (recorded at build time)" % info["git_head"]
    except Exception:
        return None


def _lib_sha16():
    import hashlib
    from painter_amd._lib import LIB_PATH
    try:
        return hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_launcher(args):
    """--gpus N > 1 without a launcher: start N ranks ourselves (one per GPU, RCCL) and hand back their exit code."""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and args.backend != "gloo":
        raise SystemExit("bench.py: --gpus %d asked for, %d visible -- refusing to print a line for fewer GPUs" % (args.gpus, n_dev))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # RCCL's ring kernels share the CUs with the backward: a stand-in with a ring step's traffic hides under the backward at +2-2.5 % on
    # >= 32 workgroups and can no longer finish inside it on <= 16 (profiles/r03_gradsync_overlap_one_gpu.log) -- keep RCCL on the safe side
    env.setdefault("NCCL_MIN_NCHANNELS", "32")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (BASELINE configs[1]: 8; --model vit_huge: 4 = global 32 on 8 GPUs)")
    ap.add_argument("--model", default="vit_large", choices=sorted(MODELS))
    ap.add_argument("--min-seconds", type=float, default=5.0, help="length of the second, sustained timed region (0 = skip)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--profile-steps", type=int, default=2, help="steps of the separate, event-instrumented pass after the timed region (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-gpu", action="store_true", help="skip the reference-on-this-GPU leg (unmodified reference model, PyTorch eager)")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (SegGPT N = 32 hipGraph, ViT-H/14 B = 4) after the headline measurement")
    ap.add_argument("--gradsync", default="all", choices=["all", "per_block", "coarse", "rs_ag", "ddp"],
                    help="N > 1: which gradient-exchange arrangement(s) to time (all = the four back to back, value = the best)")
    ap.add_argument("--eval", action="store_true", help="eval mode (no DropPath)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = DEBUG: run the N > 1 branch (relaunch, parameter broadcast, GradSync inside the backward, MAX-reduced timing) "
                         "with the ranks sharing however many GPUs are visible (RCCL refuses two ranks per device); the line says so")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_launcher(args))

    from painter_amd import models_painter, ops, parallel
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("NCCL_MIN_NCHANNELS", "32")      # started by an external launcher: same floor as relaunch_under_launcher()
    rank, local, world = parallel.init_distributed(backend=args.backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if args.backend == "gloo":
        local = local % n_dev                                   # debug arrangement: ranks share the visible devices
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    distributed = world > 1 or parallel._SELFTEST
    if distributed:
        import torch.distributed as dist
        assert dist.get_world_size() == world

    spec = MODELS[args.model]
    if args.batch is None:
        args.batch = 8 if args.model == "vit_large" else 4
    model = getattr(models_painter, spec["factory"])(compute_dtype=args.dtype)
    randomize_parameters(model, seed=1 + rank)              # replicas start different (main_train.py:190) ...
    model = model.to(dev)
    model.train(not args.eval)
    if distributed:
        parallel.broadcast_parameters(model)                # ... and adopt rank 0's parameters, as under the DDP wrapper
    cfg = model._cfg
    imgs, tgts, mask, valid = synthetic_inputs(args.batch, cfg.H, cfg.W, cfg.L, 1234 + rank, dev)
    net = [model]                                            # what step() calls: the bare module, or the DDP wrapper around it

    def step():
        for p in model.parameters():
            p.grad = None
        # as in plain training, where the optimizer has just rewritten the rel-pos tables: every step re-packs them (one launch for all
        # blocks); the bf16 weight copies stay -- painter_amd.optim.AdamW rewrites those inside its own pass, outside forward + backward
        model._hot.relpos_stale()
        loss, _, _ = net[0](imgs, tgts, bool_masked_pos=mask, valid=valid)
        loss.backward()
        return loss

    timer = KernelTimer(ops)
    timer.install()                                          # inert (one attribute test per op call) until .active is set

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_region():
        """W untimed warm-up steps, then EXACTLY K steps between barriers; -> (seconds = MAX over ranks, last loss)."""
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, float(loss.item())

    # N > 1: the gradient exchange is the only thing an 8-GPU node adds, and it has never met real RCCL ranks -- so the same K steps are
    # timed under three arrangements back to back and the line reports all of them (extra.gradsync_variants); `value` is the best one.
    #   per_block  GradSync as the engine drives it: ~125 asynchronous RCCL collectives per step from inside the backward, reverse layer order
    #   coarse     GradSync(mode="coarse"): the same gradients in four coalesced launches (400 MB each)
    #   ddp        the plain DistributedDataParallel wrapper (the reference's own arrangement, main_train.py:340): its reducer only sees
    #              the gradients when our autograd node returns them all at once, i.e. NO overlap with the backward -- the floor
    n_ranks = torch.distributed.get_world_size() if distributed else 1
    variants = {}
    gs_errors = {}
    ddp_error = None
    arrangement = "n/a"
    if not distributed:
        dt, lossv = timed_region()
    else:
        assert n_ranks == args.gpus, (n_ranks, args.gpus)    # the group really has as many ranks as the line will claim (config.rccl_ranks)
        todo = ["per_block", "coarse", "rs_ag", "ddp"] if args.gradsync == "all" else [args.gradsync]
        gs_errors = {}
        # the same K steps with NO exchange at all (every rank differentiates its own batch; nothing is updated, so replicas stay equal):
        # what an arrangement adds to this is its EXPOSED communication (extra.exposed_comm_ms) -- the line explains its own efficiency
        model.grad_sync = None
        no_exchange_s, _ = timed_region()
        for name in [v for v in todo if v != "ddp"]:
            # one arrangement failing at the Python level leaves the others (the failure is the same on every rank: same code, same
            # arguments).  RuntimeError included: that is what torch.distributed / RCCL raise for an unsupported argument --
            # raised at call time, before anything of that call is enqueued; per_block (plain all_reduce, the most basic call) runs first.
            try:
                model.grad_sync = parallel.GradSync(mode=name)
                d_, l_ = timed_region()
                variants[name] = {"seconds": d_, "loss": l_, "collective_launches_per_step": model.grad_sync.launches // (args.steps + args.warmup)}
            except Exception as e:
                gs_errors[name] = "%s: %s" % (type(e).__name__, str(e)[:200])
        if gs_errors and not variants and "ddp" not in todo:
            raise RuntimeError("every gradient-exchange arrangement failed: %r" % gs_errors)
        gs_best = min((v for v in variants), key=lambda v: variants[v]["seconds"]) if variants else None

    # ---- sustained region (below) runs under the better GradSync arrangement; the DDP wrapper is timed after it (its reducer hooks stay on
    # the parameters for the life of the wrapper, so it goes last)
    if distributed and gs_best is not None:
        model.grad_sync = parallel.GradSync(mode=gs_best)

    # ---- sustained region: the same step until --min-seconds have gone by (every rank runs the same number of steps, fixed
    # beforehand from the first region's pace, so the collectives stay matched); `value` above stays the exactly-K-steps number
    sustained = None
    clock_power = None
    if distributed:
        dt = variants[gs_best]["seconds"] if gs_best is not None else None
    if args.min_seconds > 0 and dt is not None:
        n_sus = max(args.steps, int(args.min_seconds / (dt / args.steps)) + 1)
        sampler = ClockPowerSampler() if rank == 0 else None
        barrier()
        if sampler is not None:
            sampler.start()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        torch.cuda.synchronize()
        barrier()
        dts = time.perf_counter() - t0
        if sampler is not None:
            clock_power = sampler.stop()
        if distributed:
            t = torch.tensor([dts], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dts = float(t.item())
        sustained = {"steps": n_sus, "seconds": round(dts, 3), "value": round(n_ranks * args.batch * n_sus / dts, 3),
                     "ms_per_step": round(dts / n_sus * 1e3, 3)}
        if distributed:
            sustained["gradsync"] = gs_best

    if distributed:
        if args.gradsync in ("all", "ddp"):
            model.grad_sync = None
            try:                                            # the comparison leg never takes the line down (GradSync's numbers are already in)
                net[0] = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index] if args.backend == "nccl" else None)
                d_, l_ = timed_region()
                variants["ddp"] = {"seconds": d_, "loss": l_}
            except Exception as e:
                if not variants:
                    raise
                ddp_error = "%s: %s" % (type(e).__name__, str(e)[:200])
        arrangement = min(variants, key=lambda v: variants[v]["seconds"])
        dt, lossv = variants[arrangement]["seconds"], variants[arrangement]["loss"]
        for v in variants.values():
            v["value"] = round(n_ranks * args.batch * args.steps / v["seconds"], 3)
            v["ms_per_step"] = round(v["seconds"] / args.steps * 1e3, 3)
            v["exposed_comm_ms"] = round((v["seconds"] - no_exchange_s) / args.steps * 1e3, 3)      # against the same process's step without any exchange
            v["seconds"] = round(v["seconds"], 4)

    # ---- separate profiled pass (never compare a profiled arm with an un-profiled one: `value` above is un-instrumented)
    kernels = {}
    launch_check = None
    if args.profile_steps > 0 and not distributed:          # single rank only: the extra steps would need every rank's all-reduce
        from painter_amd._lib import lib as _lib
        side = model._hot.use_side_stream
        model._hot.use_side_stream = False                  # one stream: an event pair brackets exactly its own kernels
        # ... and the parameter-gradient kernels get the sizing they have when they own the chip (the engine sizes them for HALF the chip
        # because they normally run beside the data-gradient chain: timed alone at that size they looked 30 % slower than the
        # one-stream rocprof summary under profiles/, which is taken with PAINTER_AMD_SIDE_STREAM=0, i.e. full-chip sizing)
        knobs = {k: _lib.pa_debug_get(k) for k in (3, 6)}      # restored below to what was in effect, not re-derived
        _lib.pa_debug_set(3, 0)
        _lib.pa_debug_set(6, 0)
        step()
        timer.active = True
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize()
        timer.active = False
        model._hot.use_side_stream = side
        for k, v in knobs.items():
            _lib.pa_debug_set(k, v)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        kernels = timer.results(peak)
        for v in kernels.values():
            v["ms_per_step"] = round(v["kernel_ms_total"] / args.profile_steps, 3)
        # every nn.Linear is one bracket per pass: qkv, proj, fc1(+GELU), fc2 per block + decoder_embed = 4 * depth + 1 (round 2's
        # timer counted fc1 twice: ops.linear_gelu calls ops.linear_fwd)
        want = (4 * len(model.blocks) + 1) * args.profile_steps
        # (+ the patch-embed weight gradient where the bf16 im2col fast path is on: it is an ordinary nn.Linear weight gradient there)
        extra_w = args.profile_steps if ops.patch_cols_ok(model.compute_dtype, args.batch, cfg.L, cfg.P, cfg.D) else 0
        launch_check = {fam: {"launches": kernels.get(fam, {}).get("launches", 0), "expected": w_}
                        for fam, w_ in (("gemm256_fwd", want), ("gemm256_dgrad", want), ("gemm256_wgrad", want + extra_w))}
        bad = {f: v for f, v in launch_check.items() if v["launches"] != v["expected"]}
        if bad:
            # a dispatch regression (GEMMs falling off the gemm256 path, a double-counted bracket): the per-family table would be
            # computed over the wrong launch set, so it is WITHHELD (roofline.kernels / dominant_kernel = null) and the process exits
            # non-zero after printing the line -- the headline number above is already measured and stays in it
            print("bench.py: unexpected launch counts in the profiled pass, per-family table withheld: %s" % bad, file=sys.stderr)
            launch_check["mismatch"] = True
            kernels = {}

    if rank == 0:
        ips = n_ranks * args.batch * args.steps / dt
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        dominant = None
        if kernels:
            dk = max(kernels, key=lambda k: kernels[k]["kernel_ms_total"])
            dominant = dict(kernels[dk], family=dk)
        traffic = None
        traffic_meta = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")      # PMC passes (tools/pmc_traffic.py), bytes per launch
        if os.path.exists(tpath) and dominant is not None and args.model == "vit_large":
            try:
                tj = json.load(open(tpath))
                key = {"gemm256_fwd": "fc1", "gemm256_wgrad": "wgrad", "gemm256_dgrad": "dgrad", "attention_fwd": "attn_fwd",
                       "attention_bwd": "attn_bwd_dq"}.get(dominant["family"])
                meta = tj.get("_meta", {})
                if meta.get("lib_sha16") == _lib_sha16():    # the PMC passes were taken on THIS library; anything else is refused
                    traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
                    traffic_meta = meta
                else:
                    traffic_meta = {"refused": "profiles/roofline_traffic.json was measured on library %s, this run loads %s"
                                               % (meta.get("lib_sha16"), _lib_sha16())}
            except Exception:
                traffic = None
        traffic_by_family = None
        if traffic_meta is not None and "refused" not in traffic_meta:
            alg = _algorithmic_bytes()
            traffic_by_family = {k: {"hbm_bytes_per_launch": tj[k]["hbm_bytes_per_launch"], "hbm_read_bytes": tj[k].get("hbm_read_bytes"),
                                     "hbm_write_bytes": tj[k].get("hbm_write_bytes"), "algorithmic_bytes_per_launch": alg[k]["bytes"],
                                     "ratio": round(tj[k]["hbm_bytes_per_launch"] / alg[k]["bytes"], 2), "algorithmic": alg[k]["what"]}
                                 for k in alg if k in tj}
        achieved = ips / n_ranks * spec["blocks"] / 1e12
        out = {
            "metric": "images/sec (896x448 pairs) %s fwd+bwd" % ("ViT-L" if args.model == "vit_large" else "ViT-H/14"),
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": spec["workload"] % (args.dtype, args.batch, n_ranks),
                       "global_batch": n_ranks * args.batch, "image": "896x448x3 stitched pair", "tokens": cfg.L,
                       "mode": "eval" if args.eval else "train (DropPath 0.1)", "parallelism": "dp%d" % n_ranks,
                       "rccl_ranks": n_ranks if (distributed and args.backend == "nccl") else 0,
                       "backend": (args.backend if args.backend == "nccl" else "gloo -- DEBUG arrangement: %d ranks on %d visible GPU(s); not a scaling number" % (n_ranks, n_dev)) if distributed else "n/a",
                       "taps": list(cfg.taps),
                       "grad_allreduce": {"per_block": "RCCL, one message per weight matrix + one flat message per block, started from inside the backward (GradSync)",
                                          "coarse": "RCCL, four coalesced launches of ~400 MB from inside the backward (GradSync mode coarse)",
                                          "rs_ag": "RCCL, per_block's messages as reduce_scatter_tensor + all_gather_into_tensor (all-peer pattern over the seven xGMI links), started from inside the backward (GradSync)",
                                          "ddp": "torch DistributedDataParallel wrapper (the reference's arrangement; no overlap with the HIP backward)"}.get(arrangement, "n/a")},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "definition": "images/s/GPU x %.3f TFLOP (attention + MLP blocks, fwd+bwd; SURVEY.md 8d) / dense bf16 MFMA peak" % (spec["blocks"] / 1e12),
                         "whole_model_achieved": round(ips / n_ranks * spec["whole"] / 1e12, 2),
                         "whole_model_frac": round(ips / n_ranks * spec["whole"] / 1e12 / peak, 4),
                         "traffic": traffic, "traffic_source": "profiles/roofline_traffic.json (rocprofv3 PMC passes of this bench on this library, tools/pmc_traffic.py; not re-measured inside this run)" if traffic is not None else None,
                         "traffic_meta": traffic_meta, "traffic_by_family": traffic_by_family, "launch_count_check": launch_check,
                         "dominant_kernel": dominant, "kernels": kernels,
                         "profiled_pass": "separate pass of %d steps after the timed region, HIP events per launch, one stream, parameter-gradient kernels sized for the whole chip (as in profiles/*one_stream_kernel_stats.csv)" % args.profile_steps},
            "loss": round(lossv, 6),
            "sustained": sustained,
            "extra": clock_power,
            "build": {"git_head": _git_head(), "lib_sha16": _lib_sha16()},
        }
        if variants:
            out["extra"] = dict(out["extra"] or {}, gradsync_variants=variants, gradsync_chosen=arrangement,
                                no_exchange_ms_per_step=round(no_exchange_s / args.steps * 1e3, 3),
                                exposed_comm_ms=variants[arrangement]["exposed_comm_ms"],
                                sustained_arrangement=(sustained or {}).get("gradsync"))
            if ddp_error:
                out["extra"]["gradsync_ddp_error"] = ddp_error
            if gs_errors:
                out["extra"]["gradsync_errors"] = gs_errors
        if n_ranks == 1 and args.dtype == "bf16" and not args.no_optimizer and args.model == "vit_large":
            out["optimizer_step"] = optimizer_step_ms(model, step)
        if n_ranks == 1 and not distributed and args.dtype == "bf16" and args.model == "vit_large" and not args.no_secondary:
            # the other single-GPU BASELINE configurations, after (never inside) the headline measurement; each a few seconds
            out["secondary"] = {}
            for key, fn in (("seggpt_n32", secondary_seggpt_n32), ("vit_huge", secondary_vit_huge)):
                try:
                    t0 = time.perf_counter()
                    out["secondary"][key] = fn(dev)
                    out["secondary"][key]["leg_seconds"] = round(time.perf_counter() - t0, 1)
                except Exception as e:                  # a secondary leg never takes the line down
                    out["secondary"][key] = {"error": "%s: %s" % (type(e).__name__, e)}
        if n_ranks == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if n_ranks == 1 and not args.no_reference_gpu and args.model == "vit_large" and not args.eval:
            try:
                rg = reference_gpu_baseline(model, (imgs, tgts, mask, valid), dev)
            except Exception as e:                      # a baseline leg never takes the line down
                rg = {"error": "%s: %s" % (type(e).__name__, e)}
            if rg is not None:
                out["reference_gpu"] = rg
                if "value" in rg:
                    out["vs_reference_gpu"] = round(out["value"] / rg["value"], 2)
    else:
        out = None
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)
    if distributed:
        # RCCL writes a version banner ("RCCL version : ... Librccl path : ...") to stdout when the library is torn down at interpreter exit:
        # it would FOLLOW the JSON line.  The line is the last thing this process means to say: flush and leave without running the
        # library's exit handlers (every collective has completed behind the barrier above; the process group is destroyed).
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3 if (launch_check is not None and launch_check.get("mismatch")) else 0)
    if launch_check is not None and launch_check.get("mismatch"):
        raise SystemExit(3)


if __name__ == "__main__":
    main()
