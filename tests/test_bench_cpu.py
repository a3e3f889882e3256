"""Host logic of bench.py that needs no GPU: the per-launch event timer must bracket every op ONCE (round 2 counted fc1 twice: the wrapped
ops.linear_gelu calls the wrapped module-global ops.linear_fwd), and the model table must name the BASELINE configurations."""
import types

import torch

import bench


class _Event:
    """Stand-in for torch.cuda.Event on a GPU-less host: record() stamps a counter, elapsed_time() returns the distance."""
    clock = 0

    def __init__(self, enable_timing=True):
        self.t = None

    def record(self):
        _Event.clock += 1
        self.t = _Event.clock

    def elapsed_time(self, other):
        return float(other.t - self.t)


def test_kernel_timer_counts_a_nested_op_once(monkeypatch):
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    ops = types.SimpleNamespace()
    calls = []

    def linear_fwd(x, w, *a, **k):
        calls.append("fwd")
        return x

    def linear_gelu(x, w, *a, **k):
        calls.append("gelu")
        return ops.linear_fwd(x, w), None            # the module-global (wrapped) name, as painter_amd/ops.py does

    for name in bench.KernelTimer.OPS:
        setattr(ops, name, lambda *a, **k: None)
    ops.linear_fwd, ops.linear_gelu = linear_fwd, linear_gelu
    t = bench.KernelTimer(ops)
    t.install()
    x, w = torch.zeros(4, 8), torch.zeros(16, 8)
    ops.linear_gelu(x, w)                            # inactive: passes through, records nothing
    assert not t.rec
    t.active = True
    ops.linear_gelu(x, w)
    ops.linear_fwd(x, w)
    t.active = False
    fam = t.rec["gemm256_fwd"]
    assert len(fam["events"]) == 2                   # one bracket for the fc1 + GELU call, one for the plain forward: NOT three
    assert fam["flops"] == 2 * (2.0 * 4 * 8 * 16)
    assert calls == ["gelu", "fwd", "gelu", "fwd", "fwd"] and t.depth == 0
    res = t.results(2500.0)
    assert res["gemm256_fwd"]["launches"] == 2


def test_model_table_names_the_baseline_configurations():
    from painter_amd import models_painter
    for key, spec in bench.MODELS.items():
        assert hasattr(models_painter, spec["factory"]) and spec["blocks"] < spec["whole"]
    assert bench.MODELS["vit_large"]["blocks"] == 4.034e12 and bench.MODELS["vit_huge"]["head_dim"] == 80


def test_committed_pmc_traffic_is_stamped_with_the_built_library():
    """`roofline.traffic` is read from profiles/roofline_traffic.json and refused when that file was measured on another library
    (bench.py compares `lib_sha16`): a tree whose kernels changed after the last PMC pass would silently report `traffic: null`.  The build is
    deterministic (same sources + flags -> same bytes), so the stamp can be checked wherever the library has been built; it also pins the
    algorithmic-bytes table to the families the PMC file holds."""
    import json
    import os
    tj = json.load(open(os.path.join(bench.ROOT, "profiles", "roofline_traffic.json")))
    sha = bench._lib_sha16()
    if sha is None:
        import pytest
        pytest.skip("library not built here")
    if tj["_meta"]["lib_sha16"] != sha:
        # not a failure of the code under test: the library was rebuilt somewhere that does not reproduce the measured bytes (another path or
        # toolchain), or a kernel changed after the last PMC pass -- bench.py will report `traffic: null` until tools/gpu_visit.sh pmc5 is re-run
        import pytest
        pytest.skip("profiles/roofline_traffic.json is stamped %s, the library here is %s" % (tj["_meta"]["lib_sha16"], sha))
    alg = bench._algorithmic_bytes()
    assert set(alg) == {k for k in tj if not k.startswith("_")}
    for k, v in alg.items():
        assert 0.9 <= tj[k]["hbm_bytes_per_launch"] / v["bytes"] < 4.0, (k, tj[k]["hbm_bytes_per_launch"], v["bytes"])
