"""Round 6 diagnostics: what does a round of full tiles and a round of half tiles cost inside the MIXED instantiation of gemm256?
Times the fc1-shaped forward GEMM (12544 x 4096, K = 1024 and 4096, bias epilogue) under uniform 224-row tiles, uniform 256-row tiles and
mixed plans with a forced number of full row tiles (pa_debug_set(12, 2) + pa_debug_set(14, nfull)).  Diagnostics only.

    python tools/gemm_mixed_probe.py
"""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops                      # noqa: E402
from painter_amd._lib import EPI_BIAS, lib       # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    M, N = 12544, 4096
    for K in (1024, 4096):
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        fn = lambda: ops.linear_fwd(x, w, b, EPI_BIAS)
        ref = None
        rows = []
        for name, knobs in (("uniform 224-row tiles (896 = 3.5 rounds)", {4: 2, 12: 1}), ("uniform 256-row tiles (784 = 3.06 rounds)", {4: 1, 12: 1}),
                            ("mixed: rule (48 full rows + 2 half rows)", {12: 0}),
                            ("mixed nfull = 48: 768 full (3 rounds) +  32 half", {12: 2, 14: 48}), ("mixed nfull = 40: 640 full (2.5)      + 288 half", {12: 2, 14: 40}),
                            ("mixed nfull = 32: 512 full (2 rounds) + 544 half", {12: 2, 14: 32}), ("mixed nfull = 16: 256 full (1 round)  + 1056 half", {12: 2, 14: 16}),
                            ("mixed nfull = 1 :  16 full            + 1536 half (6 rounds)", {12: 2, 14: 1})):
            for k in (4, 12, 14):
                lib.pa_debug_set(k, 0)
            for k, v in knobs.items():
                assert lib.pa_debug_set(k, v) == 0
            out = fn()
            if ref is None:
                ref = out
            same = bool(torch.equal(out, ref))
            rows.append((name, timeit(fn), same))
        for k in (4, 12, 14):
            lib.pa_debug_set(k, 0)
        print("M = %d, N = %d, K = %d (bias epilogue, bf16 out)" % (M, N, K))
        for name, t, same in rows:
            print("  %-62s %8.1f us   %s" % (name, t, "bit-identical" if same else "DIFFERENT"), flush=True)


if __name__ == "__main__":
    main()
