"""Block-wise mask sampler of the training input pipeline: the drop-in for the reference's
`util.masking_generator.MaskingGenerator` (Painter/util/masking_generator.py:15-93, built at Painter/main_train.py:256-260, called once
per sample at Painter/data/pairdataset.py:187).

It stays on the host on purpose.  The sampler is a rejection loop over Python's `random` (Mersenne Twister) and numpy's global
generator whose every draw depends on the cells accepted so far; a sample's mask is 56 x 28 cells and costs ~50 us of host time in a
DataLoader worker that is otherwise waiting for JPEG decode.  A device version would have to replay the very same MT19937 streams
serially to return the reference's masks -- there is no parallel work to win and nothing to overlap.  What this class guarantees
instead is the reference's masks bit for bit from the same seeds: the same draws in the same order (area, log-aspect, then top and
left only when the rectangle fits; the top-up / trim through `np.random.choice` on the row-major cell list).
tests/test_masking_generator_cpu.py runs it beside the unmodified reference class and against recorded masks."""
import math
import random

import numpy as np


class MaskingGenerator:
    """`MaskingGenerator(input_size, num_masking_patches, min_num_patches=4, max_num_patches=None, min_aspect=0.3, max_aspect=None)`;
    calling it returns an int32 [height][width] array with exactly `num_masking_patches` ones."""

    ATTEMPTS = 10                                   # rectangles tried per block before giving up (masking_generator.py:42)

    def __init__(self, input_size, num_masking_patches, min_num_patches=4, max_num_patches=None, min_aspect=0.3, max_aspect=None):
        self.height, self.width = input_size if isinstance(input_size, tuple) else (input_size, input_size)
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = max_num_patches if max_num_patches is not None else num_masking_patches
        hi = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(hi))

    def __repr__(self):
        return "Generator(%d, %d -> [%d ~ %d], max = %d, %.3f ~ %.3f)" % (
            self.height, self.width, self.min_num_patches, self.max_num_patches, self.num_masking_patches, *self.log_aspect_ratio)

    def get_shape(self):
        return self.height, self.width

    def _add_block(self, mask, budget):
        """Tries up to ATTEMPTS rectangles; paints the first one that adds between 1 and `budget` new cells.  -> cells added."""
        lo, hi = self.log_aspect_ratio
        for _ in range(self.ATTEMPTS):
            area = random.uniform(self.min_num_patches, budget)
            aspect = math.exp(random.uniform(lo, hi))
            h = int(round(math.sqrt(area * aspect)))
            w = int(round(math.sqrt(area / aspect)))
            if not (w < self.width and h < self.height):
                continue                            # no position draws for a rectangle that does not fit
            top = random.randint(0, self.height - h)
            left = random.randint(0, self.width - w)
            block = mask[top:top + h, left:left + w]
            fresh = h * w - int(block.sum())
            if 0 < fresh <= budget:
                block[...] = 1
                return fresh
        return 0

    def __call__(self):
        mask = np.zeros((self.height, self.width), dtype=np.int32)
        count = 0
        while count < self.num_masking_patches:
            added = self._add_block(mask, min(self.num_masking_patches - count, self.max_num_patches))
            if added == 0:
                break
            count += added
        # exact count: random cells are cleared / set through numpy's global generator, indexed in row-major order
        if count != self.num_masking_patches:
            surplus = count > self.num_masking_patches
            rows, cols = np.nonzero(mask if surplus else mask == 0)
            pick = np.random.choice(rows.shape[0], abs(count - self.num_masking_patches), replace=False)
            mask[rows[pick], cols[pick]] = 0 if surplus else 1
        assert int(mask.sum()) == self.num_masking_patches, "mask count %d" % int(mask.sum())
        return mask
