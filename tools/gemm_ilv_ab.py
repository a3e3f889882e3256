"""A/B of the gemm256 DMA placement (csrc/gemm256.h, template parameter ILV): 0 = both LDS-DMA pieces of a phase issued in front of
the phase's first barrier (rounds 1-2), 1 / 2 = one / both issued between the MFMAs of the phase's matrix segment.  Needs the
experiment build (all three schedules in one library):
    PA_EXTRA_FLAGS=-DG256_ILV_AB PA_LIB_NAME=libpainter_hip_ilv.so python -m painter_amd.build
    PAINTER_AMD_LIB=painter_amd/lib/libpainter_hip_ilv.so python tools/gemm_ilv_ab.py [step]
Per shape (ViT-L, B = 8): forward with each epilogue, data gradient, weight gradient -- time per schedule, interleaved rounds in one
process, and bit-equality of every output with schedule 0 (the arithmetic order is untouched, only when the copies are issued).
`step`: additionally the whole training step (bench.py's loop) per schedule."""
import statistics
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops                                                # noqa: E402
from painter_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, lib  # noqa: E402

DEV, T = "cuda", torch.bfloat16
ILVS = (0, 1, 2)


def timeit(fn, iters=20, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.rand(s, generator=g) * 2 - 1).to(T).to(DEV)
    shapes = [("qkv", 12544, 3072, 1024), ("proj", 12544, 1024, 1024), ("fc1", 12544, 4096, 1024), ("fc2", 12544, 1024, 4096),
              ("fc1_2B", 25088, 4096, 1024), ("dec", 12544, 16384, 4096)]
    ok = True
    for name, M, N, K in shapes:
        x, w, dy = rnd(M, K), rnd(N, K) * 0.05, rnd(M, N)
        b = (torch.rand(N, generator=g) - 0.5).to(DEV)
        resid = torch.randn(M, N, generator=g).to(DEV)
        cases = {"fwd bias": lambda: (ops.linear_fwd(x, w, b, EPI_BIAS),)}
        if name.startswith("fc1"):
            cases["fwd gelu"] = lambda: ops.linear_gelu(x, w, b)
        if name in ("proj", "fc2"):
            cases["fwd resid"] = lambda: (ops.linear_fwd(x, w, b, EPI_BIAS_RESID, resid=resid),)
        cases["dgrad"] = lambda: (ops.linear_dgrad(dy, w),)
        if name != "dec":
            cases["wgrad"] = lambda: (ops.linear_wgrad(dy, x),)
        fl = 2.0 * M * N * K
        for tag, fn in cases.items():
            ref = None
            times = {i: [] for i in ILVS}
            for rnd_ in range(3):
                for i in ILVS:
                    lib.pa_debug_set(5, 1 + i)
                    out = fn()
                    torch.cuda.synchronize()
                    if i == 0 and ref is None:
                        ref = [o.clone() for o in out]
                    elif rnd_ == 0:
                        same = all(torch.equal(a, r) for a, r in zip(out, ref))
                        ok &= same
                        if not same:
                            print("   !!! %s %s: schedule %d differs from schedule 0" % (name, tag, i))
                    times[i].append(timeit(fn))
            med = {i: statistics.median(v) for i, v in times.items()}
            print("%-7s %-9s " % (name, tag) + "  ".join("ILV%d %7.1f us %6.0f TF/s" % (i, med[i], fl / med[i] / 1e6) for i in ILVS)
                  + "   best ILV%d (%+.1f %% vs 0)" % (min(med, key=med.get), 100 * (min(med.values()) / med[0] - 1)), flush=True)
    lib.pa_debug_set(5, 0)
    print("bit-identical across schedules:", ok)
    if len(sys.argv) > 1 and sys.argv[1] == "step":
        import bench
        from painter_amd import models_painter
        m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
        bench.randomize_parameters(m, seed=1)
        m = m.to(DEV).train()
        c = m._cfg
        inp = bench.synthetic_inputs(8, c.H, c.W, c.L, 1234, torch.device(DEV))

        def step():
            for p in m.parameters():
                p.grad = None
            loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
            loss.backward()
        for _ in range(3):
            step()
        res = {i: [] for i in ILVS}
        for _ in range(4):
            for i in ILVS:
                lib.pa_debug_set(5, 1 + i)
                step()
                torch.cuda.synchronize()
                res[i].append(timeit(step, iters=8, warm=0) / 1e3)
        for i in ILVS:
            v = statistics.median(res[i])
            print("whole step ILV%d: %.2f ms = %.1f images/s   rounds %s" % (i, v, 8e3 / v, ["%.2f" % t for t in res[i]]), flush=True)
        lib.pa_debug_set(5, 0)
    assert ok


if __name__ == "__main__":
    main()
