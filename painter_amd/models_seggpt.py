"""Drop-in replacement for the reference's `models_seggpt` module (SegGPT/SegGPT_inference/models_seggpt.py).

`SegGPT` = the Painter network + two segmentation-type tokens (models_seggpt.py:285-286, :415-420), the cross-prompt
feature ensemble inside each block (`Block.forward(x, merge)`, :220-232, schedule :426-429) and the loss without the
ignore rule (:448-469).  Forward and backward run entirely in libpainter_hip.so.  The reference only ever runs the ensemble under
@torch.no_grad (seggpt_engine.py:26); like the reference's Block.forward it is differentiable here too (engine.py), in eval and in
train mode, and the two type tokens get their gradients.
"""
from functools import partial

import torch.nn as nn

from .models_painter import Painter


class SegGPT(Painter):
    _SEGGPT = True

    def forward(self, imgs, tgts, bool_masked_pos=None, valid=None, seg_type=None, merge_between_batch=-1):
        if seg_type is None:
            raise ValueError("SegGPT.forward needs seg_type ([N,1] of 0 = semantic / 1 = instance), models_seggpt.py:415-418")
        return self._run(imgs, tgts, bool_masked_pos, valid, seg_type=seg_type, merge_between_batch=merge_between_batch)


def seggpt_vit_large_patch16_input896x448(**kwargs):
    """models_seggpt.py:483-494."""
    model = SegGPT(
        img_size=(896, 448), patch_size=16, embed_dim=1024, depth=24, num_heads=16,
        drop_path_rate=0.1, window_size=14, qkv_bias=True,
        mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
        window_block_indexes=(list(range(0, 2)) + list(range(3, 5)) + list(range(6, 8)) + list(range(9, 11)) +
                              list(range(12, 14)), list(range(15, 17)), list(range(18, 20)), list(range(21, 23))),
        residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
        decoder_embed_dim=64,
        loss_func="smoothl1",
        **kwargs)
    return model
