"""Would a batch-split, two-chain schedule pay?  Times one B = 8 training step against two INDEPENDENT B = 4 steps (two model copies,
two HIP streams, two host threads) running concurrently: the second number bounds what pipelining two half-batches through the step
could reach (every kernel of one half overlaps some kernel of the other, filling the idle CUs that tile quantisation leaves).
Diagnostics."""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
import bench                                                    # noqa: E402
from painter_amd import models_painter                          # noqa: E402


def make(B, dev, seed):
    model = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(model, seed=seed)
    model = model.to(dev).train()
    cfg = model._cfg
    data = bench.synthetic_inputs(B, cfg.H, cfg.W, cfg.L, 1234 + seed, dev)
    return model, data


def step(model, data):
    imgs, tgts, mask, valid = data
    for p in model.parameters():
        p.grad = None
    loss, _, _ = model(imgs, tgts, bool_masked_pos=mask, valid=valid)
    loss.backward()


def main():
    dev = torch.device("cuda", 0)
    m8, d8 = make(8, dev, 1)
    for _ in range(2):
        step(m8, d8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step(m8, d8)
    torch.cuda.synchronize()
    print("one B=8 step                         %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
    for side in (True, False):
        pairs = [make(4, dev, 2), make(4, dev, 3)]
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        for m, _ in pairs:
            m._hot.use_side_stream = side

        def run(i, n):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    step(*pairs[i])

        for n in (2, 6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=run, args=(i, n)) for i in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
        print("two concurrent B=4 steps (side streams %s)  %.2f ms per pair" % ("on" if side else "off", dt), flush=True)
        with torch.cuda.stream(streams[0]):
            step(*pairs[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                step(*pairs[0])
            torch.cuda.synchronize()
        print("one B=4 step alone (side stream %s)         %.2f ms" % ("on" if side else "off", (time.perf_counter() - t0) / 5 * 1e3), flush=True)
        del pairs


if __name__ == "__main__":
    main()
