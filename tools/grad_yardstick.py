"""The reference's OWN bf16 gradient deviation, measured on the GPU box ("Missing 5" / next-round item 1b).

BASELINE.md section 4 has a yardstick for `pred` (the reference under bf16 autocast deviates 1e-2 from its fp32 self); there was none for
gradients, so the bf16 gates of tests/test_model_gpu.py (sampled rel-max per tensor, rel-pos tables by full-tensor Frobenius) rested on an
argument.  This runs the UNMODIFIED reference Painter (oracle/ref_import.py; on the GPU box the subset staged by oracle/stage_ref.py) at
ViT-L, B = 1, eval mode, with the parameters and batch of tests/golden/painter_vitl.npz (random_params(cfg, 1), synthetic_batch(cfg, 1,
1234, "random")) through PyTorch-ROCm eager

    (a) in fp32                      (b) under torch.autocast("cuda", bfloat16), the arrangement of engine_train.py:65-75

and prints, per metric the tests use, the deviation of (b) from (a): every 997th element of every gradient with more than 4096 elements
as max|a - b| / max|b| per tensor (rel-pos tables and all other tensors apart), and the relative Frobenius error of every rel_pos_h /
rel_pos_w gradient over the whole tensor.  The HIP bf16 build is measured the same way against (a) in the same process, so that the two
deviations stand side by side; (a) is also checked against the committed CPU fixture.  Baseline / test infrastructure only.

    python tools/grad_yardstick.py [out.json]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import painter_oracle as O       # noqa: E402
from oracle import ref_import                # noqa: E402
from tests import golden_util as G           # noqa: E402

STRIDE, SMALL = 997, 4096


def deviation(grads, base):
    """-> dict of the metrics of tests/test_model_gpu.py::_check_bf16_samples for `grads` against `base` ({name: tensor})."""
    rel, oth, fro, fro_all = [], [], [], []
    for n, b in base.items():
        a = grads[n].detach().float().cpu().reshape(-1)
        b = b.detach().float().cpu().reshape(-1)
        is_rel = n.endswith("rel_pos_h") or n.endswith("rel_pos_w")
        if b.numel() > SMALL:
            e = G.rel_err(a[::STRIDE], b[::STRIDE])
            (rel if is_rel else oth).append((e, n))
        f = G.rel_fro(a, b)
        fro_all.append((f, n))
        if is_rel:
            fro.append((f, n))
    top = lambda v: {"worst": max(v)[0], "tensor": max(v)[1], "median": sorted(x[0] for x in v)[len(v) // 2], "count": len(v)} if v else None
    return {"sampled_relmax_relpos_tables": top(rel), "sampled_relmax_other_tensors": top(oth), "relpos_tables_full_frobenius": top(fro),
            "all_tensors_full_frobenius": top(fro_all)}


FAMILIES = ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight", "attn.rel_pos_h", "attn.rel_pos_w")


def per_block(grads, base, depth):
    """Round 6: per transformer block and weight family, the sampled rel-max AND the whole-tensor relative
    Frobenius error -- does the deviation grow towards block 0 (rounding accumulated along the residual chain) or is it flat (the
    weight-gradient GEMM's own bf16 operands)?  -> {family: {"sampled_relmax": [depth], "frobenius": [depth]}}"""
    out = {}
    for fam in FAMILIES:
        sm, fr = [], []
        for i in range(depth):
            n = "blocks.%d.%s" % (i, fam)
            a = grads[n].detach().float().cpu().reshape(-1)
            b = base[n].detach().float().cpu().reshape(-1)
            sm.append(G.rel_err(a[::STRIDE], b[::STRIDE]) if b.numel() > SMALL else G.rel_err(a, b))
            fr.append(G.rel_fro(a, b))
        out[fam] = {"sampled_relmax": sm, "frobenius": fr}
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "grad_yardstick.json")
    assert torch.cuda.is_available() and ref_import.reference_available()
    dev = torch.device("cuda:0")
    cfg = O.vit_large_config()
    P = O.random_params(cfg, 1)
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 1, 1234, "random")
    maskb = mask.reshape(1, *cfg.grid)
    ref = ref_import.load_reference_painter()
    rm = ref.painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1()
    rm.load_state_dict(P, strict=True)
    rm = rm.to(dev).eval()
    res = {"config": "ViT-L 896x448 (painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1), B = 1, eval mode, random_params(cfg, 1), "
                     "synthetic_batch(cfg, 1, 1234, 'random') = the case of tests/golden/painter_vitl.npz", "torch": torch.__version__,
           "device": torch.cuda.get_device_name(0)}

    def run_ref(autocast_dtype):
        rm.zero_grad(set_to_none=True)
        args = (imgs.to(dev), tgts.to(dev), maskb.to(dev), valid.clone().to(dev))
        if autocast_dtype is None:
            loss, pred, _ = rm(*args)
        else:
            with torch.autocast("cuda", dtype=autocast_dtype):
                loss, pred, _ = rm(*args)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach().float()), pred.detach().float().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in rm.named_parameters()}

    l32, p32, g32 = run_ref(None)
    res["reference_fp32_gpu"] = {"loss": l32}
    # (a) against the committed CPU fixture of the same case: the GPU fp32 run IS the reference the fixtures hold
    fx = G.load("painter_vitl.npz")
    case = "vitl_b1/"
    rep = []
    G.check_grad_digests(fx, case, list(g32.items()), 1e-3, 1e-3, 1e-3, report=rep)
    res["reference_fp32_gpu"]["vs_cpu_fixture"] = {"loss_rel": abs(l32 - float(fx[case + "loss"])) / abs(float(fx[case + "loss"])),
                                                   "worst_sampled_gradient_relmax": max(rep)[0], "tensor": max(rep)[1]}
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        l16, p16, g16 = run_ref(dt)
        d = deviation(g16, g32)
        d["loss_rel"] = abs(l16 - l32) / abs(l32)
        d["pred_rel_frobenius"] = G.rel_fro(p16, p32)
        d["pred_rel_max"] = G.rel_err(p16, p32)
        if name == "bf16":
            d["per_block"] = per_block(g16, g32, cfg.depth)
        res["reference_%s_autocast_vs_reference_fp32" % name] = d
    del rm
    torch.cuda.empty_cache()

    # the HIP bf16 build on the same case, against the same fp32 gradients
    from functools import partial

    import torch.nn as nn

    from painter_amd import models_painter
    m = models_painter.Painter(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                               drop_path_rate=0.1, window_size=14, qkv_bias=True, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                               window_block_indexes=([0, 1], [3, 4]), residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
                               decoder_embed_dim=cfg.decoder_embed_dim, loss_func=cfg.loss_func, compute_dtype="bf16")
    m.load_state_dict(P, strict=True)
    m = m.to(dev).eval()
    loss, pred, _ = m(imgs.to(dev), tgts.to(dev), bool_masked_pos=maskb.to(dev), valid=valid.clone().to(dev))
    loss.backward()
    torch.cuda.synchronize()
    gh = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    d = deviation(gh, g32)
    d["loss_rel"] = abs(float(loss.detach()) - l32) / abs(l32)
    d["pred_rel_frobenius"] = G.rel_fro(pred.detach().float().cpu(), p32)
    d["pred_rel_max"] = G.rel_err(pred.detach().float().cpu(), p32)
    d["per_block"] = per_block(gh, g32, cfg.depth)
    res["hip_bf16_build_vs_reference_fp32"] = d
    try:
        from painter_amd._lib import LIB_PATH
        import hashlib
        res["lib_sha16"] = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16]
    except Exception:
        pass
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items()}, indent=1)[:6000])
    # the per-block table, reference-bf16 next to HIP-bf16 (whole-tensor relative Frobenius error, then sampled rel-max)
    rb, hb = res["reference_bf16_autocast_vs_reference_fp32"]["per_block"], res["hip_bf16_build_vs_reference_fp32"]["per_block"]
    for metric in ("frobenius", "sampled_relmax"):
        print("\n%s per block: reference bf16 autocast | HIP bf16 build (both against the reference's fp32 gradients)" % metric)
        print("blk " + " ".join("%21s" % f.split(".", 1)[1] for f in FAMILIES))
        for i in range(cfg.depth):
            print("%3d " % i + " ".join("%10.2e|%10.2e" % (rb[f][metric][i], hb[f][metric][i]) for f in FAMILIES))


if __name__ == "__main__":
    main()
