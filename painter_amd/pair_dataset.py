"""Host half of the device input pipeline: the reference's `PairDataset` (Painter/data/pairdataset.py:22-193)
with the pixel work taken out.

`PairSpecDataset.__getitem__` makes every decision `PairDataset.__getitem__` makes, in the same order and from the same generators
(torch's global RNG for the transform draws and the half-mask coin, Python's `random` for the partner pair, the mask generator's
own), decodes the files exactly as `_load_image` does -- and returns `(SampleSpec, mask)` instead of tensors.  A DataLoader over it
(collate_fn=`collate_specs`) feeds `painter_amd.pair_pipeline.DevicePairPipeline.build_batch`, which does on the MI355X what the
reference's transform stack (main_train.py:232-251) did in the worker.  tests/test_pair_dataset_cpu.py runs the UNMODIFIED reference
`PairDataset` beside this class on the same seeds.
"""
import json
import os.path
import random
import time

import numpy as np
import torch
from PIL import Image

from . import pair_pipeline as PP

TYPE_WEIGHTS = [0.1, 0.2, 0.15, 0.25, 0.2, 0.15, 0.05, 0.05]          # pairdataset.py:57, one weight per json list


class PairSpecDataset(torch.utils.data.Dataset):
    """Arguments as `PairDataset` (pairdataset.py:39-52) with the transform objects replaced by the numbers main_train.py:232-251
    builds them from: `input_size` (H, W), `min_random_scale`, the ColorJitter strengths and probabilities.  `train=False` gives
    the validation stack (scale (0.9999, 1), no jitter, no flip, no second crop; main_train.py:251-254)."""

    def __init__(self, root, json_path_list, masked_position_generator=None, use_two_pairs=True, half_mask_ratio=0.,
                 input_size=(896, 448), min_random_scale=0.3, train=True, jitter=(0.4, 0.4, 0.2, 0.1), jitter_p=0.8, flip_p=0.5):
        self.root = root
        self.pairs, self.weights = [], []
        for list_index, path in enumerate(json_path_list):           # one json list per task; each list shares its task weight evenly
            with open(path) as f:
                entries = json.load(f)
            self.pairs += entries
            self.weights += [TYPE_WEIGHTS[list_index] / len(entries)] * len(entries)
        self.use_two_pairs = use_two_pairs
        if use_two_pairs:                                             # partner candidates: indices of the pairs of each type (:66-73)
            self.pair_type_dict = {}
            for index, entry in enumerate(self.pairs):
                if "type" in entry:
                    self.pair_type_dict.setdefault(entry["type"], []).append(index)
        self.masked_position_generator = masked_position_generator
        self.half_mask_ratio = half_mask_ratio
        self.input_size = tuple(input_size)
        self.min_random_scale = min_random_scale
        self.train = train
        self.jitter, self.jitter_p, self.flip_p = tuple(jitter), jitter_p, flip_p

    def _load_image(self, path):
        """What pairdataset.py:81-98 hands to the transforms: the file as an RGB PIL picture.  A file that cannot be opened is
        retried once a second for ever (a flaky network mount stalls the worker, it does not kill the epoch); nyuv2 depth maps
        (16-bit, 1e-4 m units, 10 m range) are rescaled to 0..255 floats before the RGB conversion."""
        full = os.path.join(self.root, path)
        picture = None
        while picture is None:
            try:
                picture = Image.open(full)
            except OSError as err:
                print("could not open %s (%s), trying again in 1 s" % (full, err))
                time.sleep(1)
        if "sync_depth" in path:
            picture = Image.fromarray(np.array(picture) / 10000. * 255)
        return picture.convert("RGB")

    def _pair_spec(self, pair, stack):
        """One call of `cur_transforms(image, target, ...)` (pairdataset.py:135, :144): the draws of the chosen stack, in its order."""
        image = np.array(self._load_image(pair['image_path']))
        target = np.array(self._load_image(pair['target_path']))
        h, w = image.shape[:2]
        if stack == "full":                             # main_train.py:233-241
            crop = PP.sample_resized_crop(h, w, (self.min_random_scale, 1.0))
            ops, factors = PP.sample_color_jitter(*self.jitter, p=self.jitter_p)
            flip = PP.sample_flip(self.flip_p)
        else:                                           # transform_train2 / 3 / val: main_train.py:242-254
            crop = PP.sample_resized_crop(h, w, (0.9999, 1.0))
            ops, factors, flip = (), (), False
        return PP.PairSpec(image=image, target=target, crop=crop, jitter_ops=ops, jitter_factors=factors, flip=flip)

    def __getitem__(self, index):
        pair = self.pairs[index]
        pair_type = pair['type']
        # the reference loads both files of the first pair, then transforms; then the partner pair (pairdataset.py:107-146)
        if not self.train:
            stack = "plain"
        elif "inst" in pair_type or "pose" in pair_type:  # no augmentation for instance segmentation / pose (:127-131)
            stack = "plain"
        else:
            stack = "full"
        specs = [self._pair_spec(pair, stack)]
        if self.use_two_pairs:
            pair2 = self.pairs[random.choice(self.pair_type_dict[pair_type])]
            assert pair2['type'] == pair_type
            specs.append(self._pair_spec(pair2, stack))
        use_half_mask = bool(torch.rand(1)[0] < self.half_mask_ratio)
        seccrop = None
        if self.train and not ("inst" in pair_type or "pose" in pair_type or use_half_mask):
            canvas_h = self.input_size[1] * len(specs)
            seccrop = PP.sample_resized_crop(canvas_h, self.input_size[1], (self.min_random_scale, 1.0), ratio=(0.3, 0.7))
        if use_half_mask:
            mask = np.zeros(self.masked_position_generator.get_shape(), dtype=np.int32)
            mask[mask.shape[0] // 2:, :] = 1
        else:
            mask = self.masked_position_generator()
        return PP.SampleSpec(pairs=specs, pair_type=pair_type, seccrop=seccrop), mask

    def __len__(self):
        return len(self.pairs)


def collate_specs(items):
    """DataLoader collate_fn: -> (list of SampleSpec, int32 mask tensor [B][h][w])."""
    specs = [s for s, _ in items]
    masks = torch.from_numpy(np.stack([np.asarray(m) for _, m in items]))
    return specs, masks
