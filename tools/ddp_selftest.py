"""Single-GPU exercise of the multi-GPU gradient path (run with PAINTER_AMD_DDP_SELFTEST=1): a 1-rank RCCL group, GradSync
started from inside the two-stream backward (in-place all-reduce of the weight matrices, flattened small tensors), then the same
step without the exchange -- gradients must be bit-identical (AVG over one rank is the identity)."""
import os
import sys

import torch

sys.path.insert(0, ".")
os.environ["PAINTER_AMD_DDP_SELFTEST"] = "1"
import bench  # noqa: E402
from painter_amd import models_painter, parallel  # noqa: E402


def main():
    rank, local, world = parallel.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).eval()                   # eval: no DropPath randomness between the two runs
    c = m._cfg
    inp = bench.synthetic_inputs(2, c.H, c.W, c.L, 1234, dev)

    def grads(sync):
        m.grad_sync = sync
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters()}, float(loss.detach())

    ga, la = grads(parallel.GradSync())
    gb, lb = grads(None)
    gc, lc = grads(None)
    gd, ld = grads(parallel.GradSync(mode="coarse"))        # round 5: the four coalesced launches (dist.all_reduce_coalesced with AVG on RCCL)
    coarse_bad = [n for n in gd if not torch.equal(gd[n], gb[n])]
    print("coarse arrangement: mismatching", len(coarse_bad), coarse_bad[:8], "collective launches", m.grad_sync.launches, flush=True)
    assert not coarse_bad and ld == lb
    ge, le = grads(parallel.GradSync(mode="rs_ag"))         # round 6: every message as reduce_scatter_tensor + all_gather_into_tensor (AVG on RCCL)
    rs_bad = [n for n in ge if not torch.equal(ge[n], gb[n])]
    print("rs_ag arrangement: mismatching", len(rs_bad), rs_bad[:8], "collective launches", m.grad_sync.launches, flush=True)
    assert not rs_bad and le == lb
    bad = [n for n in ga if not torch.equal(ga[n], gb[n])]
    nondet = [n for n in gb if not torch.equal(gb[n], gc[n])]
    worst = max([float((ga[n] - gb[n]).abs().max() / gb[n].abs().max().clamp_min(1e-30)) for n in bad] + [0.0])
    print("world", torch.distributed.get_world_size(), "backend", torch.distributed.get_backend(), "loss", la, lb,
          "tensors", len(ga), "mismatching", len(bad), bad[:8], "worst rel", worst, "| run-to-run (no exchange) mismatching", len(nondet), nondet[:8])
    assert not bad and not nondet and la == lb
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print("DDP selftest OK")


if __name__ == "__main__":
    main()
