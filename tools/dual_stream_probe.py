"""Experiment: does running two independent half-batches on two HIP streams fill the tails / store bursts of each other's
kernels?  Two model replicas (B=4 each) on two streams vs one model at B=8 on one stream."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from painter_amd import models_painter  # noqa: E402


def make(batch, seed, dev):
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).train()
    c = m._cfg
    return m, bench.synthetic_inputs(batch, c.H, c.W, c.L, seed, dev)


def step(m, inp):
    for p in m.parameters():
        p.grad = None
    loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
    loss.backward()


def main():
    dev = torch.device("cuda", 0)
    m8, i8 = make(8, 1234, dev)
    for _ in range(3):
        step(m8, i8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step(m8, i8)
    torch.cuda.synchronize()
    t8 = (time.perf_counter() - t0) / 8
    print("one stream, B=8: %.2f ms/step" % (t8 * 1e3))
    # host-only cost of enqueueing one step (GPU idle at the start: measures launch overhead + queue depth limits)
    mA, iA = make(4, 1234, dev)
    mB, iB = make(4, 1235, dev)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(sA):
            step(mA, iA)
        with torch.cuda.stream(sB):
            step(mB, iB)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        with torch.cuda.stream(sA):
            step(mA, iA)
        with torch.cuda.stream(sB):
            step(mB, iB)
    torch.cuda.synchronize()
    t44 = (time.perf_counter() - t0) / 8
    print("two streams, B=4+4: %.2f ms per pair" % (t44 * 1e3))
    t0 = time.perf_counter()
    for _ in range(8):
        step(mA, iA)
        step(mB, iB)
    torch.cuda.synchronize()
    t4s = (time.perf_counter() - t0) / 8
    print("one stream, B=4 then B=4: %.2f ms per pair" % (t4s * 1e3))


if __name__ == "__main__":
    main()
