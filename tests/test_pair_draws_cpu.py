"""The draw order of the device input pipeline's host half against the reference's transform stack.

tests/golden/pair_draws.json holds what the UNMODIFIED Painter/data/pair_transforms.py did, stacked as main_train.py:233-251 stacks
it, on twelve seeds (tests/golden/make_golden_pair_draws.py: which functional op was called on which picture with which numbers).
Here `painter_amd.pair_pipeline.sample_*` are driven from the same seeds in the order `PairSpecDataset._pair_spec` / `__getitem__`
call them, and the events their results imply (crop both pictures with the shared box and the per-picture interpolation, jitter
ops on the image only in the drawn order, flip both) must be the recorded ones, number for number.  Where /root/reference is
mounted the same comparison also runs against a live run of the reference, so a stale fixture cannot hide a change."""
import importlib.util
import json
import os

import pytest
import torch

from painter_amd import pair_pipeline as PP

HERE = os.path.dirname(os.path.abspath(__file__))
OP_NAME = {PP.BRIGHTNESS: "adjust_brightness", PP.CONTRAST: "adjust_contrast", PP.SATURATION: "adjust_saturation", PP.HUE: "adjust_hue"}


def _maker():
    spec = importlib.util.spec_from_file_location("make_golden_pair_draws", os.path.join(HERE, "golden", "make_golden_pair_draws.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _our_events(kind, h, w, i1, i2, input_size=(896, 448), min_random_scale=0.3):
    """The events one transform-stack call implies, from OUR draws (pair_dataset.py `_pair_spec` / the second crop of `__getitem__`)."""
    ev = []
    if kind == "seccrop":
        box = PP.sample_resized_crop(h, w, (min_random_scale, 1.0), ratio=(0.3, 0.7))
        size = list(input_size)
        ops, factors, flip = (), (), False
    else:
        size = [input_size[1], input_size[1]]
        if kind == "train":
            box = PP.sample_resized_crop(h, w, (min_random_scale, 1.0))
            ops, factors = PP.sample_color_jitter(0.4, 0.4, 0.2, 0.1, p=0.8)
            flip = PP.sample_flip(0.5)
        else:
            box = PP.sample_resized_crop(h, w, (0.9999, 1.0))
            ops, factors, flip = (), (), False
    ev.append(["resized_crop", "img", *box, size, i1])
    ev.append(["resized_crop", "tgt", *box, size, i2])
    for o, f in zip(ops, factors):
        ev.append([OP_NAME[o], "img", f])
    if flip:
        ev += [["hflip", "img"], ["hflip", "tgt"]]
    return ev


def _replay(seed, cases):
    torch.manual_seed(seed)
    for kind, h, w, i1, i2, recorded in cases:
        ours = _our_events(kind, h, w, i1, i2)
        assert ours == recorded, (seed, kind, ours, recorded)


def test_draws_match_the_recorded_reference_stack():
    with open(os.path.join(HERE, "golden", "pair_draws.json")) as f:
        fx = json.load(f)
    assert len(fx) == 12
    kinds = set()
    jitter = flips = 0
    for seed, cases in fx.items():
        _replay(int(seed), cases)
        for kind, *_, ev in cases:
            kinds.add(kind)
            jitter += any(e[0].startswith("adjust_") for e in ev)
            flips += any(e[0] == "hflip" for e in ev)
    assert kinds == {"train", "plain", "seccrop"} and jitter > 10 and flips > 10          # the fixture exercises every branch


def test_fixture_covers_the_skipped_jitter_and_the_fallback_crop():
    """RandomApply's skip branch (no adjust_* events, no factor draws) and the ten-tries-failed central crop both occur in the seeds."""
    with open(os.path.join(HERE, "golden", "pair_draws.json")) as f:
        fx = json.load(f)
    skipped = sum(1 for cases in fx.values() for kind, *_, ev in cases if kind == "train" and not any(e[0].startswith("adjust_") for e in ev))
    full_frame = sum(1 for cases in fx.values() for kind, h, w, _, _, ev in cases if kind == "plain" and ev[0][2:6] == [0, 0, h, w])
    assert skipped >= 3 and full_frame >= 6


@pytest.mark.skipif(not os.path.exists("/root/reference/Painter/data/pair_transforms.py"), reason="the reference tree is not mounted here")
def test_draws_match_a_live_run_of_the_reference_stack():
    mk = _maker()
    pt = mk.load_reference_transforms()
    try:
        for seed in (100, 101, 102, 103):
            _replay(seed, mk.record(pt, seed))
    finally:
        import sys
        for k in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
            sys.modules.pop(k, None)
