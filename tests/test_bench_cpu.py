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


def test_closing_profiles_of_the_round_carry_one_library_stamp():
    """Round 6: the closing artefacts of rounds 4 and 5 were taken one commit before the shipped library.  Every
    profiles/r06_* file that tools/adopt_profiles.py filed starts with `# library <sha16> git <head> | <command>`; the CLOSING set (the
    kernel statistics of the one- and two-stream step, the matrix-pipe counters, the PMC traffic file, the bench line) must all name the
    SAME library, and where this tree's build reproduces the measured binary that hash must be the built library's."""
    import glob
    import json
    import os
    import re
    prof = os.path.join(bench.ROOT, "profiles")
    stamps = {}
    for f in sorted(glob.glob(os.path.join(prof, "r06_*.csv")) + glob.glob(os.path.join(prof, "r06_*.log"))):
        first = open(f).readline()
        if os.path.basename(f).startswith("r06_ab_"):
            continue                                     # A/B logs describe experiments (often reverted): free-form headers
        m = re.match(r"# library ([0-9a-f]{16}) git (\S+) \| ", first)
        assert m, "%s lacks the `# library <sha16> git <head> | <command>` header of tools/adopt_profiles.py" % os.path.basename(f)
        stamps[os.path.basename(f)] = m.group(1)
    closing = ["r06_bench_one_stream_kernel_stats.csv", "r06_bench_two_stream_kernel_stats.csv", "r06_pmc_mfma_busy_per_kernel.csv"]
    have = [c for c in closing if c in stamps]
    if not have:
        import pytest
        pytest.skip("the closing profiles of round 6 are not filed yet")
    shas = {stamps[c] for c in have}
    tj = json.load(open(os.path.join(prof, "roofline_traffic.json")))
    shas.add(tj["_meta"]["lib_sha16"])
    line = os.path.join(prof, "r06_bench_line.json")
    if os.path.exists(line):
        shas.add(json.loads(open(line).read().strip().splitlines()[-1])["build"]["lib_sha16"])
    assert len(shas) == 1, "closing artefacts were taken on different libraries: %s" % {c: stamps[c] for c in have}
    built = bench._lib_sha16()
    if built is not None and built not in shas:
        import pytest
        pytest.skip("closing artefacts are stamped %s, the library built here is %s (another path / toolchain, or kernels changed after the closing visit)" % (shas, built))
