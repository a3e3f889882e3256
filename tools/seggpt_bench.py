"""BASELINE configs[3]: seggpt_vit_large_patch16_input896x448 inference, N in-context prompts sharing one query
(merge_between_batch=0, seg_type ones, bottom-half mask broadcast as [1, L] -- seggpt_engine.py:36-47), bf16, forward only,
captured once in a hipGraph and replayed.  Prints one JSON line: images/sec (one image = one 896x448 stitched pair) for the
eager forward and for graph replay.  The replayed output is compared with the eager one (bit-identical: same kernels)."""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from painter_amd import models_seggpt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompts", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-reference-gpu", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    N = args.prompts
    m = models_seggpt.seggpt_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).eval()
    c = m._cfg
    imgs, tgts, _, valid = bench.synthetic_inputs(N, c.H, c.W, c.L, 1234, dev)
    imgs[:, :, c.H // 2:] = imgs[:1, :, c.H // 2:]          # every prompt is stitched above the same query image
    mask = torch.zeros((1, c.L), dtype=torch.float32, device=dev)
    mask[:, c.L // 2:] = 1
    seg_type = torch.ones((N, 1), device=dev)

    def fwd():
        with torch.no_grad():
            return m(imgs, tgts, mask, valid, seg_type, 0)

    for _ in range(2):
        loss, pred, _ = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        loss, pred, _ = fwd()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / args.iters
    ref_pred, ref_loss = pred.clone(), float(loss.item())

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # warm-up on the capture stream (per-stream workspaces, weight casts)
        fwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        g_loss, g_pred, _ = fwd()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        graph.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / args.iters
    same = bool(torch.equal(g_pred, ref_pred))
    flop = N * 1.5897e12
    # the unmodified reference model (SegGPT_inference/models_seggpt.py through oracle/ref_import.py) on this GPU, PyTorch-ROCm eager: fp32 as
    # seggpt_engine.py:36-47 calls it (no autocast), and under bf16 autocast -- same parameters, same batch.  Baseline only.
    reference_gpu = None
    try:
        from oracle import ref_import
        if ref_import.reference_available() and not args.no_reference_gpu:
            rmod = ref_import.load_reference_seggpt().seggpt_vit_large_patch16_input896x448()
            rmod.load_state_dict({k: v.detach().float().cpu() for k, v in m.state_dict().items()}, strict=True)
            rmod = rmod.to(dev).eval()
            res = {}
            for name, ctx in (("fp32", None), ("bf16_autocast", torch.bfloat16)):
                def rfwd():
                    with torch.no_grad():
                        if ctx is None:
                            return rmod(imgs, tgts, mask, valid, seg_type, 0)
                        with torch.autocast("cuda", dtype=ctx):
                            return rmod(imgs, tgts, mask, valid, seg_type, 0)
                rfwd()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    rl, rp, _ = rfwd()
                torch.cuda.synchronize()
                res[name] = ((time.perf_counter() - t0) / 3, float(rl.item()))
            reference_gpu = {"fp32_ms": round(res["fp32"][0] * 1e3, 2), "fp32_images_per_sec": round(N / res["fp32"][0], 2),
                             "bf16_autocast_ms": round(res["bf16_autocast"][0] * 1e3, 2),
                             "bf16_autocast_images_per_sec": round(N / res["bf16_autocast"][0], 2), "loss_fp32": res["fp32"][1],
                             "torch": torch.__version__}
            del rmod
    except Exception as e:
        reference_gpu = {"error": "%s: %s" % (type(e).__name__, e)}
    print(json.dumps({
        "metric": "images/sec (896x448 pairs) SegGPT ViT-L forward, N prompts + feature ensemble", "prompts": N, "dtype": "bf16",
        "eager_ms": round(t_eager * 1e3, 3), "eager_images_per_sec": round(N / t_eager, 2),
        "hipgraph_ms": round(t_graph * 1e3, 3), "hipgraph_images_per_sec": round(N / t_graph, 2),
        "tflops_hipgraph": round(flop / t_graph / 1e12, 1), "replay_equals_eager": same, "loss": ref_loss, "reference_gpu": reference_gpu}))


if __name__ == "__main__":
    main()
