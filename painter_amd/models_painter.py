"""Drop-in replacement for the reference's `models_painter` module (Painter/models_painter.py).

Same public surface -- `Painter(...)` constructor arguments, `forward(imgs, tgts, bool_masked_pos, valid)`
-> (loss, patchify(pred), bool_masked_pos), `patchify/unpatchify`, `no_weight_decay`, `patch_embed.num_patches`,
`blocks`, parameter names/shapes (checkpoint ABI) and the factory
`painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1` -- so engine_train.py and the eval scripts run unchanged.
The sub-modules below are *parameter containers only*: forward and backward of the whole network run in the
hand-written HIP kernels of libpainter_hip.so (painter_amd/engine.py); nothing is computed with ATen ops and there is
no CPU fallback (CPU tensors raise).

Extra constructor argument: compute_dtype ("bf16" default | "fp32", env PAINTER_AMD_DTYPE) selects the operand type
of the GEMM/attention kernels; parameters, residual stream, gradients and the loss are fp32 either way.
"""
import os
from functools import partial

import torch
import torch.nn as nn

from .engine import HotPath, HotPathConfig


def _compute_dtype(arg):
    s = arg if arg is not None else os.environ.get("PAINTER_AMD_DTYPE", "bf16")
    if isinstance(s, torch.dtype):
        return s
    s = str(s).lower()
    if s in ("bf16", "bfloat16"):
        return torch.bfloat16
    if s in ("fp32", "float32", "f32"):
        return torch.float32
    raise ValueError("compute_dtype must be 'bf16' or 'fp32', got %r" % (arg,))


# ------------------------------------------------------------------------------------------ parameter containers
class PatchEmbed(nn.Module):
    """util/vitdet_utils.py:160-186 (parameters of the k=P, s=P conv)."""

    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)


class LayerNorm2D(nn.Module):
    """util/vitdet_utils.py:189-209 (parameters)."""

    def __init__(self, normalized_shape, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.normalized_shape = (normalized_shape,)


class Attention(nn.Module):
    """models_painter.py:33-71 (parameters)."""

    def __init__(self, dim, num_heads=8, qkv_bias=True, use_rel_pos=False, rel_pos_zero_init=True, input_size=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if self.use_rel_pos:
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))
            if not rel_pos_zero_init:
                nn.init.trunc_normal_(self.rel_pos_h, std=0.02)
                nn.init.trunc_normal_(self.rel_pos_w, std=0.02)


class Mlp(nn.Module):
    """timm==0.3.2 Mlp (parameters): fc1 -> GELU -> fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Block(nn.Module):
    """models_painter.py:153-214 (parameters; window attention / residual conv blocks are dead code in the reference
    factories -- SURVEY.md fact 2 -- and are rejected here)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, drop_path=0.0, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, use_residual_block=False, input_size=None):
        super().__init__()
        if window_size != 0 or use_residual_block:
            raise NotImplementedError("window attention / residual blocks are never instantiated by the reference factories")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init, input_size=input_size)
        self.drop_path_prob = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)
        self.window_size = window_size
        self.use_residual_block = use_residual_block


# ------------------------------------------------------------------------------------------ autograd bridge
class _HotPathFn(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward both run in libpainter_hip.so."""

    @staticmethod
    def forward(ctx, hp, names, opts, imgs, tgts, mask_u8, valid, seg_type, *params):
        P = dict(zip(names, params))
        need = bool(opts.get("need_grad", True)) and any(ctx.needs_input_grad[8:])   # nothing is saved for inference
        loss_out, pred, pred_patch, S = hp.forward(P, imgs, tgts, mask_u8, valid, seg_type, opts.get("merge", -1),
                                                   opts.get("drop"), need_grad=need)
        ctx.hp, ctx.S, ctx.names = hp, S, names
        ctx.save_for_backward(*params)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(pred_patch)
        hook = opts.get("grad_sync")
        ctx.grad_sync = hook
        return loss_out[0].clone(), pred_patch

    @staticmethod
    def backward(ctx, dloss, dpatch):
        if dloss is None:
            return (None,) * (8 + len(ctx.names))
        params = ctx.saved_tensors
        P = dict(zip(ctx.names, params))
        dl = dloss.detach().to(torch.float32).reshape(1).contiguous()
        G = ctx.hp.backward(P, ctx.S, dl, sync=ctx.grad_sync)
        ctx.S = None
        grads = []
        for n, p_ in zip(ctx.names, params):
            g = G.get(n)
            grads.append(None if g is None else g.reshape(p_.shape))
        return (None,) * 8 + tuple(grads)


class Painter(nn.Module):
    """Masked-image-modelling ViT with the Painter decoder head; constructor mirrors models_painter.py:241-266."""

    _SEGGPT = False

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.,
                 qkv_bias=True, drop_path_rate=0., norm_layer=nn.LayerNorm, act_layer=nn.GELU, use_abs_pos=True,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, window_block_indexes=(), residual_block_indexes=(),
                 use_act_checkpoint=False, pretrain_img_size=224, pretrain_use_cls_token=True, out_feature="last_feat",
                 decoder_embed_dim=128, loss_func="smoothl1", compute_dtype=None, feature_taps=None):
        """compute_dtype, feature_taps: extensions (keyword-only in practice; every reference argument keeps its position and default).
        feature_taps: the four blocks whose output feeds the decoder.  None = the reference's hard-coded [5, 11, 17, 23]
        (models_painter.py:416), which is only a usable schedule at depth 24; any other depth has to name its taps (the ViT-H/14 factory
        below passes depth/4*k - 1) -- a checkpoint trained with the reference class at such a depth would compute a different function."""
        super().__init__()
        if in_chans != 3 or not use_abs_pos or not qkv_bias:
            raise NotImplementedError("HIP path is built for in_chans=3, use_abs_pos=True, qkv_bias=True (the reference factories)")
        self.pretrain_use_cls_token = pretrain_use_cls_token
        self.patch_size = patch_size
        self.patch_embed = PatchEmbed(kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size),
                                      in_chans=in_chans, embed_dim=embed_dim)
        self.patch_embed.num_patches = (img_size[0] // patch_size) * (img_size[1] // patch_size)

        self.mask_token = nn.Parameter(torch.zeros(1, 1, 1, embed_dim))
        self.segment_token_x = nn.Parameter(torch.zeros(1, 1, 1, embed_dim))
        self.segment_token_y = nn.Parameter(torch.zeros(1, 1, 1, embed_dim))
        if self._SEGGPT:
            self.type_token_cls = nn.Parameter(torch.zeros(1, 1, 1, embed_dim))
            self.type_token_ins = nn.Parameter(torch.zeros(1, 1, 1, embed_dim))
        num_patches = (pretrain_img_size // patch_size) * (pretrain_img_size // patch_size)
        num_positions = (num_patches + 1) if pretrain_use_cls_token else num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_positions, embed_dim), requires_grad=True)

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList()
        for i in range(depth):
            # NB: `i in window_block_indexes` is never true for the reference factories' tuple-of-lists
            self.blocks.append(Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop_path=dpr[i],
                                     norm_layer=norm_layer, act_layer=act_layer, use_rel_pos=use_rel_pos,
                                     rel_pos_zero_init=rel_pos_zero_init,
                                     window_size=window_size if i in window_block_indexes else 0,
                                     use_residual_block=i in residual_block_indexes,
                                     input_size=(img_size[0] // patch_size, img_size[1] // patch_size)))
        self._out_feature_channels = {out_feature: embed_dim}
        self._out_feature_strides = {out_feature: patch_size}
        self._out_features = [out_feature]
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.norm = norm_layer(embed_dim)

        self.decoder_embed_dim = decoder_embed_dim
        self.decoder_embed = nn.Linear(embed_dim * 4, patch_size ** 2 * self.decoder_embed_dim, bias=True)
        self.decoder_pred = nn.Sequential(
            nn.Conv2d(self.decoder_embed_dim, self.decoder_embed_dim, kernel_size=3, padding=1),
            LayerNorm2D(self.decoder_embed_dim),
            nn.GELU(),
            nn.Conv2d(self.decoder_embed_dim, 3, kernel_size=1, bias=True),
        )
        self.loss_func = loss_func
        torch.nn.init.normal_(self.mask_token, std=.02)
        torch.nn.init.normal_(self.segment_token_x, std=.02)
        torch.nn.init.normal_(self.segment_token_y, std=.02)
        if self._SEGGPT:
            torch.nn.init.normal_(self.type_token_cls, std=.02)
            torch.nn.init.normal_(self.type_token_ins, std=.02)
        self.apply(self._init_weights)

        ln_eps = getattr(self.norm, "eps", 1e-5)
        self.compute_dtype = _compute_dtype(compute_dtype)
        self._cfg = HotPathConfig(img_size=tuple(img_size), patch_size=patch_size, embed_dim=embed_dim, depth=depth,
                                  num_heads=num_heads, mlp_ratio=mlp_ratio, decoder_embed_dim=decoder_embed_dim,
                                  pretrain_img_size=pretrain_img_size, pretrain_use_cls_token=pretrain_use_cls_token,
                                  use_rel_pos=use_rel_pos, ln_eps=ln_eps, loss_func=loss_func, seggpt=self._SEGGPT,
                                  drop_path_rate=drop_path_rate, taps=feature_taps)
        self._hot = HotPath(self._cfg, self.compute_dtype)
        self.grad_sync = None          # optional painter_amd.parallel.GradSync (bucketed RCCL all-reduce inside backward)

    # models_painter.py:342-349
    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def set_compute_dtype(self, dtype):
        self.compute_dtype = _compute_dtype(dtype)
        self._hot = HotPath(self._cfg, self.compute_dtype)

    # models_painter.py:355-383 -- pure index permutations (views/copies only, no arithmetic)
    def patchify(self, imgs):
        p = self.patch_size
        assert imgs.shape[2] == 2 * imgs.shape[3] and imgs.shape[2] % p == 0
        w = imgs.shape[3] // p
        h = w * 2
        x = imgs.reshape(shape=(imgs.shape[0], 3, h, p, w, p))
        x = x.permute(0, 2, 4, 3, 5, 1)
        return x.reshape(shape=(imgs.shape[0], h * w, p ** 2 * 3))

    def unpatchify(self, x):
        p = self.patch_size
        w = int((x.shape[1] * 0.5) ** .5)
        h = w * 2
        assert h * w == x.shape[1]
        x = x.reshape(shape=(x.shape[0], h, w, p, p, 3))
        x = x.permute(0, 5, 1, 3, 2, 4)
        return x.reshape(shape=(x.shape[0], 3, h * p, w * p))

    # ------------------------------------------------------------------ forward
    def _drop_scales(self, batch, device):
        """timm 0.3.2 DropPath factors: floor(keep + U[0,1)) / keep per sample, independent for the two branches."""
        if not self.training:
            return None
        # one uniform draw, one floor, one division for every block (was six tiny launches per block = ~140 per step in front of the
        # forward): row 2 i + b of `s` = branch b of block i, 2 * batch wide (blocks after the stream merge use the first `batch`)
        n = len(self.blocks)
        probs = tuple(float(blk.drop_path_prob) for blk in self.blocks)
        cached = getattr(self, "_drop_keep", None)
        if cached is None or cached[0] != probs or cached[1].device != device:
            keep = torch.tensor([1.0 - p for p in probs for _ in range(2)], dtype=torch.float32).clamp_min(1e-12)
            cached = self._drop_keep = (probs, keep.to(device)[:, None])
        keep_d = cached[1]
        r = torch.rand((2 * n, 2 * batch), device=device, dtype=torch.float32)
        s = torch.floor(r + keep_d) / keep_d
        out = []
        for i, blk in enumerate(self.blocks):
            if blk.drop_path_prob <= 0.0:
                out.append((None, None))
            else:
                bc = 2 * batch if i <= self._cfg.merge_idx else batch
                out.append((s[2 * i, :bc], s[2 * i + 1, :bc]))         # contiguous row prefixes: no copies
        return out

    def _run(self, imgs, tgts, bool_masked_pos, valid, seg_type=None, merge_between_batch=-1):
        if not imgs.is_cuda:
            raise RuntimeError("painter_amd runs the hot path on an MI355X only (HIP kernels, no CPU/PyTorch fallback); "
                               "move the module and inputs to 'cuda'. The CPU oracle lives in oracle/ for tests.")
        B = imgs.shape[0]
        L = self.patch_embed.num_patches
        if bool_masked_pos is None:
            bool_masked_pos = torch.zeros((B, L), dtype=torch.bool, device=imgs.device)
        else:
            bool_masked_pos = bool_masked_pos.flatten(1).to(torch.bool).to(imgs.device)
        assert bool_masked_pos.shape[1] == L and bool_masked_pos.shape[0] in (1, B)
        mask_u8 = bool_masked_pos.contiguous().view(torch.uint8)
        imgs_c = imgs.detach().to(torch.float32).contiguous()
        tgts_c = tgts.detach().to(torch.float32).contiguous()
        assert imgs_c.shape == (B, 3, self._cfg.H, self._cfg.W) and tgts_c.shape == imgs_c.shape, imgs_c.shape
        if valid is None:
            valid = torch.ones_like(tgts_c)
        in_place = valid.dtype == torch.float32 and valid.is_contiguous() and valid.is_cuda
        valid_c = valid if in_place else valid.detach().to(device=imgs.device, dtype=torch.float32).contiguous()
        st = None
        if self._SEGGPT:
            st = seg_type.reshape(-1).to(device=imgs.device, dtype=torch.float32).contiguous()
            assert st.numel() == B
        names, params = zip(*self.named_parameters())
        drop = self._drop_override if getattr(self, "_drop_override", None) is not None else self._drop_scales(B, imgs.device)
        opts = {"merge": merge_between_batch, "drop": drop, "grad_sync": self.grad_sync, "need_grad": torch.is_grad_enabled()}
        loss, pred_patch = _HotPathFn.apply(self._hot, names, opts, imgs_c, tgts_c, mask_u8, valid_c, st, *params)
        if not in_place and valid.shape == valid_c.shape:
            valid.copy_(valid_c)          # the reference mutates the caller's `valid` (models_painter.py:448)
        return loss, pred_patch, bool_masked_pos

    def forward(self, imgs, tgts, bool_masked_pos=None, valid=None):
        return self._run(imgs, tgts, bool_masked_pos, valid)


def painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1(**kwargs):
    """models_painter.py:476-487 (including the tuple-of-lists window_block_indexes quirk, which disables windowing)."""
    model = Painter(
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


def painter_vit_huge_patch14_input896x448(**kwargs):
    """BASELINE.json configs[4].  NOT a reference factory: the reference only ships the ViT-L factory above;
    this is its class constructor (models_painter.py:241-266) called with ViT-H/14 sizes -- patch 14, embed 1280, depth 32, 16 heads
    (head_dim 80), 64 x 32 tokens, decoder_embed 5120 -> 14*14*64 -- with the feature taps generalised to depth/4*k - 1 = 7, 15, 23,
    31 (the reference's hard-coded [5, 11, 17, 23], models_painter.py:416, would leave blocks 24-31 without gradient)."""
    return Painter(
        img_size=(896, 448), patch_size=14, embed_dim=1280, depth=32, num_heads=16,
        drop_path_rate=0.1, window_size=14, qkv_bias=True,
        mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
        window_block_indexes=(), residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
        decoder_embed_dim=64, loss_func="smoothl1", feature_taps=(7, 15, 23, 31), **kwargs)


# names used by BASELINE.json
PainterViT = Painter
painter_vit_large_patch16_input896x448 = painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1
painter_vit_large_patch16 = painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1
