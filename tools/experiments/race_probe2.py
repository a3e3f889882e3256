"""Run-to-run determinism of the two-stream backward with / without an initialised (but unused) RCCL process group."""
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from painter_amd import models_painter  # noqa: E402


def main():
    use_pg = os.environ.get("PROBE_PG", "0") == "1"
    batch = int(os.environ.get("PROBE_B", "2"))
    if use_pg:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
        torch.distributed.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
        if os.environ.get("PROBE_COLL", "0") == "1":
            t = torch.ones(1 << 20, device="cuda")
            torch.distributed.all_reduce(t)
            torch.cuda.synchronize()
    dev = torch.device("cuda", 0)
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).eval()
    c = m._cfg
    inp = bench.synthetic_inputs(batch, c.H, c.W, c.L, 1234, dev)

    def grads():
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters()}

    m._hot.use_side_stream = False
    ref = grads()                                  # single-stream reference
    m._hot.use_side_stream = True
    configs = [None, {"dec"}, {"fc2"}, {"fc1"}, {"proj"}, {"qkv"},
               {"dec", "fc2", "fc1", "proj", "qkv", "nocolsum"}, {"dec", "fc2", "fc1", "proj", "qkv", "nowgrad"}]
    for cfgf in configs:
        m._hot.side_filter = cfgf
        res = []
        for k in range(8):
            g1 = grads()
            res.append(sum(1 for n in ref if not torch.equal(ref[n], g1[n])))
        print("pg", use_pg, "B", batch, "side ops", "all" if cfgf is None else sorted(cfgf), "-> mismatching tensors vs single-stream per run:", res)
    if use_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
