"""Helpers shared by the CPU (oracle) and GPU (HIP path) golden tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def probe_vector(name: str, numel: int) -> torch.Tensor:
    """Must match tests/golden/make_golden.py::probe_vector."""
    s = sum((i + 1) * ord(c) for i, c in enumerate(name)) % (2 ** 31)
    g = torch.Generator().manual_seed(s)
    return torch.randn(numel, generator=g, dtype=torch.float32)


def rel_err(a, b):
    """max|a-b| / max|b| -- the 'element-wise relative to max-abs' gate of SURVEY.md section 7."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_fro(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check_grad_digests(fx, prefix, named_grads, rtol_norm, atol_dot_frac, small_rtol, sample_rtol=None, report=None):
    """Compare gradients with the reference digests: L2 norm, probe dot product (error measured
    against norm(g)*norm(probe), i.e. as a cosine-scale quantity), the full small tensors, and -- the part that can tell a wrong
    gradient from a right one on the big matrices -- the reference's strided sample of every larger gradient (every 997th element,
    grad_sample/<name>): max|a - b| / max|b| over the sample must stay below sample_rtol (default small_rtol).
    report: optional list that receives (measured sampled error, name) pairs for the test to print."""
    if sample_rtol is None:
        sample_rtol = small_rtol
    stride = int(fx[prefix + "grad_sample_stride"]) if (prefix + "grad_sample_stride") in fx.files else 0
    n_sampled = 0
    names = [str(n) for n in fx[prefix + "grad_names"]]
    norms = fx[prefix + "grad_norm"]
    dots = fx[prefix + "grad_dot"]
    idx = {n: i for i, n in enumerate(names)}
    worst = []
    seen = set()
    for name, g in named_grads:
        assert name in idx, f"unexpected parameter {name}"
        seen.add(name)
        i = idx[name]
        g = g.detach().float().cpu().reshape(-1)
        n_ref = norms[i]
        n_got = float(g.double().norm())
        pv = probe_vector(name, g.numel()).double()
        d_got = float((g.double() * pv).sum())
        scale = max(n_ref, 1e-30) * float(pv.norm())
        e_norm = abs(n_got - n_ref) / max(n_ref, 1e-30)
        e_dot = abs(d_got - dots[i]) / scale
        worst.append((max(e_norm / rtol_norm, e_dot / atol_dot_frac), name, e_norm, e_dot))
        key = f"{prefix}grad/{name}"
        if key in fx.files:
            e = rel_fro(g, fx[key])
            worst.append((e / small_rtol, name + "[full]", e, 0.0))
        skey = f"{prefix}grad_sample/{name}"
        if stride and skey in fx.files:
            e = rel_err(g[::stride], fx[skey])
            worst.append((e / sample_rtol, name + "[sample]", e, 0.0))
            n_sampled += 1
            if report is not None:
                report.append((e, name))
    assert seen == set(names), f"missing grads for {sorted(set(names) - seen)[:5]}"
    if stride:
        assert n_sampled > 0 or all(np.size(fx[prefix + "grad/" + n]) <= 4096 for n in names if (prefix + "grad/" + n) in fx.files)
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, f"gradient mismatch (ratio,name,e_norm,e_dot): {worst[:6]}"
    return worst[0]
