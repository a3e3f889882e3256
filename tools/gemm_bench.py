"""Times the nn.Linear forward / dgrad / wgrad kernels on the ViT-L shapes of BASELINE configs[1] (B=8) and prints TFLOP/s
next to torch.matmul (hipBLASLt) on the same operands -- the library GEMM is an A/B comparator only, never on the path."""
import sys
import time

import torch

sys.path.insert(0, ".")
from painter_amd import ops                                                # noqa: E402
from painter_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID       # noqa: E402

DEV = "cuda"
T = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.rand(s, generator=g) * 2 - 1).to(T).to(DEV)
    shapes = [("qkv", 12544, 3072, 1024), ("proj", 12544, 1024, 1024), ("fc1", 12544, 4096, 1024), ("fc2", 12544, 1024, 4096),
              ("qkv2B", 25088, 3072, 1024), ("dec", 12544, 16384, 4096)]
    if len(sys.argv) > 1:
        shapes = [s for s in shapes if s[0] in sys.argv[1:]]
    for name, M, N, K in shapes:
        x, w, dy = rnd(M, K), rnd(N, K) * 0.05, rnd(M, N)
        b = torch.zeros(N, device=DEV)
        resid = torch.zeros(M, N, device=DEV)
        fl = 2.0 * M * N * K
        rows = []
        out = torch.empty(M, N, dtype=T, device=DEV)
        rows.append(("fwd bias", timeit(lambda: ops.linear_fwd(x, w, b, EPI_BIAS, out=out))))
        if name == "fc1":
            out2 = torch.empty(M, N, dtype=T, device=DEV)
            rows.append(("fwd gelu", timeit(lambda: ops.linear_fwd(x, w, b, EPI_BIAS_GELU, out=out, out2=out2))))
        if name in ("proj", "fc2"):
            o32 = torch.empty(M, N, device=DEV)
            rows.append(("fwd resid", timeit(lambda: ops.linear_fwd(x, w, b, EPI_BIAS_RESID, out=o32, resid=resid))))
        dx = torch.empty(M, K, dtype=T, device=DEV)
        rows.append(("dgrad", timeit(lambda: ops.linear_dgrad(dy, w, out=dx))))
        dw = torch.empty(N, K, device=DEV)
        rows.append(("wgrad", timeit(lambda: ops.linear_wgrad(dy, x, out=dw))))
        wt = w.t().contiguous()
        rows.append(("torch fwd", timeit(lambda: torch.matmul(x, wt))))
        rows.append(("torch dgrad", timeit(lambda: torch.matmul(dy, w))))
        rows.append(("torch wgrad", timeit(lambda: torch.matmul(dy.t(), x))))
        # correctness spot checks (fp32 reference of the bf16-rounded operands)
        ref = (x[:512].float() @ w.float().t())
        e1 = float((ops.linear_fwd(x, w, b, EPI_BIAS)[:512].float() - ref).abs().max() / ref.abs().max())
        refd = dy[:512].float() @ w.float()
        e2 = float((ops.linear_dgrad(dy, w)[:512].float() - refd).abs().max() / refd.abs().max())
        refw = dy.float().t()[:256] @ x.float()
        e3 = float((ops.linear_wgrad(dy, x)[:256] - refw).abs().max() / refw.abs().max())
        print("%-6s M=%d N=%d K=%d  err fwd %.1e dgrad %.1e wgrad %.1e" % (name, M, N, K, e1, e2, e3))
        for tag, ms in rows:
            print("    %-12s %8.3f ms  %7.1f TFLOP/s" % (tag, ms, fl / ms / 1e9))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
