"""With an initialised process group (host jitter), trace the backward's intermediate checksums in two-stream mode against the
single-stream reference and report the first diverging intermediate of every failing run."""
import os
import sys

import torch

sys.path.insert(0, ".")
os.environ["PAINTER_AMD_DEBUG_TRACE"] = "1"
import bench  # noqa: E402
from painter_amd import models_painter  # noqa: E402


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1")
    torch.distributed.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
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
    m._hot.side_filter = {"dec", "fc2", "fc1", "proj", "qkv", "nocolsum"}
    nfail = 0
    for k in range(40):
        r = run()
        diffs = [(a[0], a[1], b[1]) for a, b in zip(base, r) if a[1] != b[1]]
        if diffs:
            nfail += 1
            print("run", k, "diverges at", diffs[0], "(", len(diffs), "of", len(base), "checksums differ )")
    print("failing runs:", nfail, "of 40")
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
