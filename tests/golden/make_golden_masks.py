"""Records masks of the UNMODIFIED reference sampler (Painter/util/masking_generator.py) for fixed seeds -> tests/golden/masks.npz.
    python tests/golden/make_golden_masks.py          (build container only: needs /root/reference)
Configurations: main_train.py's defaults for the 896x448 / patch-16 grid (56 x 28 window, 784 masked, blocks of 16..392) and two small
grids that exercise the trim / top-up branches."""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_import          # noqa: E402

CONFIGS = [dict(input_size=(56, 28), num_masking_patches=784, max_num_patches=392, min_num_patches=16),
           dict(input_size=14, num_masking_patches=118, min_num_patches=16),
           dict(input_size=(8, 12), num_masking_patches=90, min_num_patches=4, max_num_patches=30),
           dict(input_size=(10, 10), num_masking_patches=7, min_num_patches=4, max_num_patches=6, min_aspect=0.5, max_aspect=1.5)]
SEEDS = list(range(16))


def main():
    Ref = ref_import.load_reference_masking_generator().MaskingGenerator
    out = {}
    for c, kw in enumerate(CONFIGS):
        gen = Ref(**kw)
        for seed in SEEDS:
            random.seed(seed)
            np.random.seed(seed)
            out["c%d_s%d" % (c, seed)] = np.stack([gen(), gen(), gen()]).astype(np.uint8)      # three consecutive calls per seed
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **out)
    print("masks.npz: %d entries" % len(out))


if __name__ == "__main__":
    main()
