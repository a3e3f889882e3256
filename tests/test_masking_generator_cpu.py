"""painter_amd.masking_generator.MaskingGenerator against the reference's sampler: recorded masks of the
unmodified Painter/util/masking_generator.py (tests/golden/masks.npz, tests/golden/make_golden_masks.py) and, where /root/reference is
mounted, the reference class itself run side by side on fresh seeds -- same seeds, same masks, bit for bit."""
import os
import random

import numpy as np
import pytest

from oracle import ref_import
from painter_amd.masking_generator import MaskingGenerator
from tests.golden.make_golden_masks import CONFIGS, SEEDS

HERE = os.path.dirname(os.path.abspath(__file__))


def _seed(s):
    random.seed(s)
    np.random.seed(s)


def test_masks_equal_the_recorded_reference_masks():
    fx = np.load(os.path.join(HERE, "golden", "masks.npz"))
    for c, kw in enumerate(CONFIGS):
        gen = MaskingGenerator(**kw)
        for seed in SEEDS:
            _seed(seed)
            got = np.stack([gen(), gen(), gen()])
            assert got.dtype == np.int32 and got.shape[1:] == gen.get_shape()
            assert np.array_equal(got, fx["c%d_s%d" % (c, seed)]), (c, seed)
            assert (got.sum(axis=(1, 2)) == kw["num_masking_patches"]).all()
    assert len(fx.files) == len(CONFIGS) * len(SEEDS)


def test_repr_and_shape_follow_the_reference():
    g = MaskingGenerator((56, 28), num_masking_patches=784, max_num_patches=392, min_num_patches=16)
    assert g.get_shape() == (56, 28) and g.num_patches == 1568 and g.max_num_patches == 392
    assert repr(g) == "Generator(56, 28 -> [16 ~ 392], max = 784, -1.204 ~ 1.204)"
    assert MaskingGenerator(14, 118).get_shape() == (14, 14) and MaskingGenerator(14, 118).max_num_patches == 118


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")
def test_masks_equal_a_live_run_of_the_reference():
    Ref = ref_import.load_reference_masking_generator().MaskingGenerator
    for kw in CONFIGS:
        ref, ours = Ref(**kw), MaskingGenerator(**kw)
        assert repr(ref) == repr(ours)
        for seed in range(200, 232):
            _seed(seed)
            a = [ref() for _ in range(2)]
            state = (random.getstate(), np.random.get_state()[1][:4].tolist())
            _seed(seed)
            b = [ours() for _ in range(2)]
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (kw, seed)
            assert state == (random.getstate(), np.random.get_state()[1][:4].tolist())          # the generators were advanced identically
