"""Locates the first intermediate of the two-stream backward that differs between two identical runs (PAINTER_AMD_DEBUG_TRACE)."""
import os
import sys

import torch

sys.path.insert(0, ".")
os.environ["PAINTER_AMD_DEBUG_TRACE"] = "1"
import bench  # noqa: E402
from painter_amd import models_painter  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).eval()
    c = m._cfg
    inp = bench.synthetic_inputs(2, c.H, c.W, c.L, 1234, dev)

    def run():
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()
        torch.cuda.synchronize()
        return [(n, float(v)) for n, v in m._hot.trace]

    m._hot.use_side_stream = False
    base = run()
    m._hot.use_side_stream = True
    runs = [run() for _ in range(6)]
    for k in range(6):
        diffs = [(a[0], a[1], b[1]) for a, b in zip(base, runs[k]) if a[1] != b[1]]
        print("run", k, "vs single-stream baseline: differs in", len(diffs), "of", len(base), "checksums; first:", diffs[:4])
    # and the parameter gradients themselves (no checksum kernels in between: PAINTER_AMD_DEBUG_TRACE only adds main-stream work)
    def grads(side):
        m._hot.use_side_stream = side
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters()}
    g0 = grads(False)
    for k in range(4):
        g1 = grads(True)
        bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
        print("grads run", k, "two-stream vs single-stream: mismatching", len(bad), bad[:6])


if __name__ == "__main__":
    main()
