"""Shared seeded cases for the training-input-pipeline tests.  TEST INFRASTRUCTURE -- imports oracle/."""
import numpy as np

from painter_amd import pair_pipeline as PP
from tests.seggpt_io_cases import picture


def pair(seed, h, w, crop, ops=(), factors=(), flip=False, flat_target=True):
    return PP.PairSpec(image=picture(seed, h, w), target=picture(seed + 500, h, w, flat=flat_target), crop=crop, jitter_ops=ops,
                       jitter_factors=factors, flip=flip)


def batch_specs():
    """Four heterogeneous two-pair samples: every interpolation combination, every valid rule, jitter orders with factors on both sides
    of 1 and the 0 / 1 shortcuts, flips, second crops (only where the target side is nearest, so `valid` stays exact)."""
    return [
        PP.SampleSpec(pair_type="coco_image2panoptic_sem_seg", seccrop=(100, 40, 600, 300), pairs=[
            pair(1, 480, 640, (30, 50, 400, 500), (PP.CONTRAST, PP.HUE, PP.BRIGHTNESS, PP.SATURATION), (1.31, -0.07, 0.66, 1.18), True),
            pair(2, 375, 500, (0, 0, 375, 500), (), (), False)]),
        PP.SampleSpec(pair_type="nyuv2_image2depth", pairs=[
            pair(3, 448, 448, (0, 0, 448, 448), (PP.BRIGHTNESS, PP.SATURATION, PP.CONTRAST, PP.HUE), (1.0, 0.0, 0.75, 0.1), False, flat_target=False),
            pair(4, 300, 700, (17, 123, 280, 333), (PP.HUE, PP.SATURATION, PP.BRIGHTNESS, PP.CONTRAST), (0.031, 0.93, 1.4, 1.0), True, flat_target=False)]),
        PP.SampleSpec(pair_type="coco_image2pose", pairs=[
            pair(5, 256, 192, (3, 2, 250, 188)),
            pair(6, 256, 192, (0, 0, 256, 192), flip=True)]),
        PP.SampleSpec(pair_type="ssid_2image_denoise", seccrop=(0, 10, 896, 400), pairs=[
            pair(7, 512, 512, (64, 64, 448, 448), (PP.SATURATION, PP.BRIGHTNESS, PP.HUE, PP.CONTRAST), (0.8, 0.6, -0.1, 0.61), False, flat_target=False),
            pair(8, 600, 450, (100, 0, 448, 300), (), (), True, flat_target=False)]),
    ]


def oracle_spec(s):
    """SampleSpec -> the dict oracle.pair_pipeline_oracle.build_sample takes."""
    from oracle import pair_pipeline_oracle as O
    pairs = []
    for p in s.pairs:
        jit = None
        if len(p.jitter_ops):
            jit = (list(p.jitter_ops), [O.hue_shift_byte(f) if o == PP.HUE else f for o, f in zip(p.jitter_ops, p.jitter_factors)])
        pairs.append(dict(image=p.image, target=p.target, crop=p.crop, jitter=jit, flip=p.flip))
    return dict(pairs=pairs, interpolation1=s.interpolation[0], interpolation2=s.interpolation[1], pair_type=s.pair_type, seccrop=s.seccrop)
