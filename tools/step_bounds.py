"""Upper bounds for the side-stream work: times the training step (ViT-L, B = 8, bf16) with pieces of the parameter-gradient work
REMOVED (results are then wrong: timing only).  Tells how much of the step the column sums, the slab reductions, the rel-pos gradient
and the weight-gradient GEMMs cost once overlapped with the data-gradient chain.  Diagnostics."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench                                                    # noqa: E402
from painter_amd import models_painter, ops                     # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(model, seed=1)
    model = model.to(dev).train()
    cfg = model._cfg
    imgs, tgts, mask, valid = bench.synthetic_inputs(8, cfg.H, cfg.W, cfg.L, 1234, dev)

    def fwd():
        return model(imgs, tgts, bool_masked_pos=mask, valid=valid)[0]

    def step():
        for p in model.parameters():
            p.grad = None
        fwd().backward()

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    real = {k: getattr(ops, k) for k in ("colsum", "linear_wgrad", "attn_bwd_relpos")}
    cache = {}

    def fake(name):
        def f(*a, **k):
            key = (name,) + tuple(tuple(t.shape) for t in a if torch.is_tensor(t))
            if key not in cache:
                cache[key] = real[name](*a, **k)
            return cache[key]
        return f

    cases = [("baseline", ()), ("no column sums", ("colsum",)), ("no rel-pos table gradient", ("attn_bwd_relpos",)),
             ("no weight-gradient GEMMs", ("linear_wgrad",)), ("none of the three", ("colsum", "linear_wgrad", "attn_bwd_relpos"))]
    with torch.no_grad():
        print("%-40s %.2f ms" % ("forward only (no_grad)", timed(fwd)), flush=True)
    print("%-40s %.2f ms" % ("forward only (saving activations)", timed(fwd)), flush=True)
    for _ in range(2):
        for name, off in cases:
            for k in real:
                setattr(ops, k, fake(k) if k in off else real[k])
            print("%-40s %.2f ms/step" % (name, timed(step)), flush=True)
    for k in real:
        setattr(ops, k, real[k])


if __name__ == "__main__":
    main()
