"""What does the gradient exchange cost the backward on ONE GPU?  

A 1-rank RCCL group (PAINTER_AMD_DDP_SELFTEST=1) exercises GradSync's host path, streams and buckets, but RCCL launches no ring kernel
for a single rank -- so the CU / bandwidth share of a real exchange cannot be observed here.  This tool therefore times a ViT-L B = 8
training step (bf16, train mode) in three arrangements, interleaved in one process:
    none      no exchange
    gradsync  painter_amd.parallel.GradSync on the 1-rank RCCL group (host overhead, stream ordering, the flattening of small tensors)
    standin:N the same bucket schedule, but every bucket launches `pa_debug_rmw` (read gradient + read-modify-write a scratch of the same
              size, 2 passes = the 2 (n-1)/n traffic of a ring all-reduce) on N persistent workgroups on its own stream -- an EMULATION
              of an N-channel RCCL kernel's footprint next to the backward; N = 8 / 16 / 32 / 64
Prints ms/step per arrangement (median over rounds) and the delta against `none`."""
import os
import statistics
import sys

import torch

sys.path.insert(0, ".")
os.environ.setdefault("PAINTER_AMD_DDP_SELFTEST", "1")
import bench  # noqa: E402
from painter_amd import models_painter, ops, parallel  # noqa: E402
from painter_amd._lib import check, lib  # noqa: E402


class StandIn:
    """GradSync's interface; each bucket's tensors drive the stand-in kernel on a dedicated stream."""

    def __init__(self, nblocks):
        self.nblocks = nblocks
        self.stream = torch.cuda.Stream()
        self.scratch = None
        self.events = []

    def ready(self, G, names, flat=None):
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        n = sum(G[k].numel() for k in names)
        if self.scratch is None or self.scratch.numel() < 70_000_000:
            self.scratch = torch.zeros(70_000_000, dtype=torch.float32, device="cuda")
        with torch.cuda.stream(self.stream):
            for k in names:
                t = G[k]
                if t.numel() >= (1 << 20) and t.is_contiguous():
                    t.record_stream(self.stream)
                    check(lib.pa_debug_rmw(t.data_ptr(), self.scratch.data_ptr(), t.numel(), 2, self.nblocks, self.stream.cuda_stream), "pa_debug_rmw")

    def finish(self):
        torch.cuda.current_stream().wait_stream(self.stream)


def main():
    rank, local, world = parallel.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).train()
    c = m._cfg
    inp = bench.synthetic_inputs(8, c.H, c.W, c.L, 1234, dev)

    def step():
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()

    arms = [("none", None), ("gradsync(1-rank RCCL)", parallel.GradSync())] + [("standin:%d" % n, StandIn(n)) for n in (8, 16, 32, 64)]
    res = {k: [] for k, _ in arms}
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rounds, steps = 4, 6
    for r in range(rounds):
        for name, sync in arms:
            m.grad_sync = sync
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / steps)
    base = statistics.median(res["none"])
    for name, _ in arms:
        v = statistics.median(res[name])
        print("%-24s %.2f ms/step  (%+.2f ms, %+.1f %%)   rounds %s" % (name, v, v - base, 100 * (v - base) / base, ["%.2f" % x for x in res[name]]), flush=True)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
