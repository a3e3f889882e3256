"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the Painter / SegGPT ViT hot path.

A functional, from-the-spec restatement (plain PyTorch CPU ops, any float dtype) of the reference
forward, used as the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; painter_amd/ never does.

PARITY PINNING: the reference ships no tests or golden vectors, so this
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF run in the build container:
tests/golden/make_golden.py imports the unmodified reference through oracle/ref_import.py and
commits its outputs under tests/golden/; tests/test_oracle_golden.py checks this file against
them (and, when /root/reference is mounted, against the live reference).

Every function cites the reference lines it restates (paths relative to /root/reference).
Because it is built from differentiable torch ops, torch.autograd on it is the gradient oracle.

Parameter dict `P` uses the reference's state_dict names.
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass
class OracleConfig:
    """Mirrors Painter.__init__ arguments that affect arithmetic (Painter/models_painter.py:241-266)."""
    img_size: Tuple[int, int] = (896, 448)
    patch_size: int = 16
    embed_dim: int = 1024
    depth: int = 24
    num_heads: int = 16
    mlp_ratio: float = 4.0
    decoder_embed_dim: int = 64
    pretrain_img_size: int = 224
    pretrain_use_cls_token: bool = True
    use_rel_pos: bool = True
    ln_eps: float = 1e-6           # partial(nn.LayerNorm, eps=1e-6), models_painter.py:480
    merge_idx: int = 2             # models_painter.py:408
    taps: Tuple[int, ...] = (5, 11, 17, 23)   # models_painter.py:416
    loss_func: str = "smoothl1"
    seggpt: bool = False           # SegGPT variant (models_seggpt.py)

    @property
    def grid(self):
        return self.img_size[0] // self.patch_size, self.img_size[1] // self.patch_size


# ----------------------------------------------------------------------------- index math
def patchify(imgs: torch.Tensor, p: int) -> torch.Tensor:
    """Painter/models_painter.py:355-368. (N,3,H,W) -> (N, L, p*p*3), pixel-major channel-last."""
    assert imgs.shape[2] == 2 * imgs.shape[3] and imgs.shape[2] % p == 0
    n = imgs.shape[0]
    w = imgs.shape[3] // p
    h = w * 2
    x = imgs.reshape(n, 3, h, p, w, p)
    x = x.permute(0, 2, 4, 3, 5, 1)      # nchpwq -> nhwpqc
    return x.reshape(n, h * w, p * p * 3)


def unpatchify(x: torch.Tensor, p: int) -> torch.Tensor:
    """Painter/models_painter.py:370-383."""
    w = int((x.shape[1] * 0.5) ** 0.5)
    h = w * 2
    assert h * w == x.shape[1]
    n = x.shape[0]
    x = x.reshape(n, h, w, p, p, 3)
    x = x.permute(0, 5, 1, 3, 2, 4)      # nhwpqc -> nchpwq
    return x.reshape(n, 3, h * p, w * p)


def expand_mask(bool_masked_pos: torch.Tensor, p: int, dtype) -> torch.Tensor:
    """Painter/models_painter.py:440-441: [N,L] -> pixel mask [N,3,H,W]."""
    m = bool_masked_pos.to(dtype)[:, :, None].repeat(1, 1, p * p * 3)
    return unpatchify(m, p)


def rel_pos_index(q_size: int, k_size: int) -> torch.Tensor:
    """Painter/util/vitdet_utils.py:88-93 (index built in float then .long())."""
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel.long()


def get_rel_pos(q_size: int, k_size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """Painter/util/vitdet_utils.py:63-93 (with the linear-interpolation branch)."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1),
                          size=max_rel_dist, mode="linear")
        r = r.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        r = rel_pos
    return r[rel_pos_index(q_size, k_size)]


def abs_pos_operator(src: int, h: int, w: int, dtype=torch.float32) -> torch.Tensor:
    """The constant operator M [h*w, src*src] with get_abs_pos(P) == M @ P[0,1:]
    (Painter/util/vitdet_utils.py:140-157; bicubic, align_corners=False). SURVEY.md Appendix A."""
    eye = torch.eye(src * src, dtype=dtype).reshape(1, src * src, src, src)
    m = F.interpolate(eye, size=(h, w), mode="bicubic", align_corners=False)   # [1, s*s, h, w]
    return m.reshape(src * src, h * w).t().contiguous()


def get_abs_pos(abs_pos: torch.Tensor, has_cls_token: bool, hw: Tuple[int, int]) -> torch.Tensor:
    """Painter/util/vitdet_utils.py:128-157."""
    h, w = hw
    if has_cls_token:
        abs_pos = abs_pos[:, 1:]
    xy_num = abs_pos.shape[1]
    size = int(math.sqrt(xy_num))
    assert size * size == xy_num
    if size != h or size != w:
        new = F.interpolate(abs_pos.reshape(1, size, size, -1).permute(0, 3, 1, 2),
                            size=(h, w), mode="bicubic", align_corners=False)
        return new.permute(0, 2, 3, 1)
    return abs_pos.reshape(1, h, w, -1)


# ----------------------------------------------------------------------------- blocks
def attention(x: torch.Tensor, P: Dict[str, torch.Tensor], pre: str, cfg: OracleConfig) -> torch.Tensor:
    """Painter/models_painter.py:73-89 + util/vitdet_utils.py:96-125. x: [B,H,W,C]."""
    B, H, W, C = x.shape
    nh = cfg.num_heads
    hd = C // nh
    scale = hd ** -0.5
    qkv = F.linear(x, P[pre + "qkv.weight"], P[pre + "qkv.bias"])
    qkv = qkv.reshape(B, H * W, 3, nh, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * nh, H * W, hd).unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1)
    if cfg.use_rel_pos:
        Rh = get_rel_pos(H, H, P[pre + "rel_pos_h"])
        Rw = get_rel_pos(W, W, P[pre + "rel_pos_w"])
        r_q = q.reshape(B * nh, H, W, hd)                       # UNSCALED q
        rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
        attn = (attn.view(B * nh, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]
                ).view(B * nh, H * W, H * W)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).view(B, nh, H, W, hd).permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
    return F.linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"])


def feature_ensemble(a: torch.Tensor, merge: int) -> torch.Tensor:
    """SegGPT/SegGPT_inference/models_seggpt.py:220-230 (applied to the attention-branch output)."""
    if merge <= 0:
        return a
    prompt, inputs = a.split(a.shape[1] // 2, dim=1)
    if merge == 1:
        num_prompts = a.shape[0] // 2
        inputs = inputs.reshape(2, num_prompts, -1)
        inputs = inputs.mean(dim=1, keepdim=True).expand_as(inputs)
        inputs = inputs.reshape(*prompt.shape)
    else:
        inputs = inputs.mean(dim=0, keepdim=True).expand_as(inputs)
    return torch.cat([prompt, inputs], dim=1)


def block(x, P, i: int, cfg: OracleConfig, drop_scale: Optional[torch.Tensor] = None, merge: int = 0):
    """Painter/models_painter.py:216-235 (window_size==0, no residual block: SURVEY fact 2).
    drop_scale: per-sample DropPath factor mask/keep_prob [B'] (timm 0.3.2 semantics) or None."""
    pre = f"blocks.{i}."
    C = x.shape[-1]
    shortcut = x
    h = F.layer_norm(x, (C,), P[pre + "norm1.weight"], P[pre + "norm1.bias"], cfg.ln_eps)
    a = attention(h, P, pre + "attn.", cfg)
    a = feature_ensemble(a, merge)
    ds = 1.0 if drop_scale is None else drop_scale.view(-1, 1, 1, 1).to(x.dtype)
    x = shortcut + a * ds
    h = F.layer_norm(x, (C,), P[pre + "norm2.weight"], P[pre + "norm2.bias"], cfg.ln_eps)
    h = F.linear(h, P[pre + "mlp.fc1.weight"], P[pre + "mlp.fc1.bias"])
    h = F.gelu(h)                                               # nn.GELU default = erf form
    h = F.linear(h, P[pre + "mlp.fc2.weight"], P[pre + "mlp.fc2.bias"])
    return x + h * ds


def layer_norm_2d(x, weight, bias, eps=1e-6):
    """Painter/util/vitdet_utils.py:204-209."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight[:, None, None] * x + bias[:, None, None]


# ----------------------------------------------------------------------------- model
def forward_encoder(P, cfg: OracleConfig, imgs, tgts, bool_masked_pos, seg_type=None,
                    merge_between_batch: int = -1, drop_scales: Optional[Sequence] = None) -> List[torch.Tensor]:
    """Painter/models_painter.py:385-418; SegGPT: models_seggpt.py:391-434."""
    p = cfg.patch_size
    x = F.conv2d(imgs, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=p).permute(0, 2, 3, 1)
    y = F.conv2d(tgts, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=p).permute(0, 2, 3, 1)
    B, Hp, Wp, C = x.shape
    mask_token = P["mask_token"].expand(B, Hp, Wp, -1)
    w = bool_masked_pos.unsqueeze(-1).type_as(mask_token).reshape(-1, Hp, Wp, 1)
    y = y * (1 - w) + mask_token * w
    x = x + P["segment_token_x"]
    y = y + P["segment_token_y"]
    pos = get_abs_pos(P["pos_embed"], cfg.pretrain_use_cls_token, (Hp, Wp))
    x = x + pos
    y = y + pos
    if cfg.seggpt:
        type_emb = torch.zeros(B, 1, 1, C, dtype=x.dtype)
        type_emb[seg_type.reshape(-1) == 0] = P["type_token_cls"].reshape(1, 1, C)
        type_emb[seg_type.reshape(-1) == 1] = P["type_token_ins"].reshape(1, 1, C)
        x = x + type_emb
        y = y + type_emb
    x = torch.cat((x, y), dim=0)
    out = []
    for idx in range(cfg.depth):
        merge = 0
        if cfg.seggpt and merge_between_batch >= 0 and idx >= merge_between_batch:
            merge = 1 if cfg.merge_idx >= idx else 2
        ds = None if drop_scales is None else drop_scales[idx]
        x = block(x, P, idx, cfg, ds, merge)
        if idx == cfg.merge_idx:
            x = (x[: x.shape[0] // 2] + x[x.shape[0] // 2:]) * 0.5
        if idx in cfg.taps:
            out.append(F.layer_norm(x, (C,), P["norm.weight"], P["norm.bias"], cfg.ln_eps))
    return out


def forward_decoder(P, cfg: OracleConfig, latent: List[torch.Tensor]) -> torch.Tensor:
    """Painter/models_painter.py:420-431."""
    x = torch.cat(latent, dim=-1)
    x = F.linear(x, P["decoder_embed.weight"], P["decoder_embed.bias"])
    p = cfg.patch_size
    h, w = x.shape[1], x.shape[2]
    x = x.reshape(x.shape[0], h, w, p, p, cfg.decoder_embed_dim)
    x = x.permute(0, 5, 1, 3, 2, 4)                              # nhwpqc -> nchpwq
    x = x.reshape(x.shape[0], -1, h * p, w * p)
    x = F.conv2d(x, P["decoder_pred.0.weight"], P["decoder_pred.0.bias"], padding=1)
    x = layer_norm_2d(x, P["decoder_pred.1.weight"], P["decoder_pred.1.bias"])
    x = F.gelu(x)
    return F.conv2d(x, P["decoder_pred.3.weight"], P["decoder_pred.3.bias"])


def forward_loss(cfg: OracleConfig, pred, tgts, bool_masked_pos, valid):
    """Painter/models_painter.py:433-462 (ignore rule mutates `valid` IN PLACE, :444-448);
    SegGPT: models_seggpt.py:448-469 (no ignore rule, no +1e-2)."""
    mask = expand_mask(bool_masked_pos, cfg.patch_size, pred.dtype)
    if not cfg.seggpt:
        mean = torch.tensor(IMAGENET_MEAN, dtype=tgts.dtype)[None, :, None, None]
        std = torch.tensor(IMAGENET_STD, dtype=tgts.dtype)[None, :, None, None]
        inds_ign = ((tgts * std + mean) * (1 - 1.0 * mask)).sum((1, 2, 3)) < 100 * 3
        if inds_ign.sum() > 0:
            valid[inds_ign] = 0.0
    mask = mask * valid
    d = pred - tgts
    if cfg.loss_func == "l1l2":
        loss = (d.abs() + d ** 2.0) * 0.5
    elif cfg.loss_func == "l1":
        loss = d.abs()
    elif cfg.loss_func == "l2":
        loss = d ** 2.0
    else:
        loss = F.smooth_l1_loss(pred, tgts, reduction="none", beta=0.01)
    denom = mask.sum() if cfg.seggpt else mask.sum() + 1e-2
    return (loss * mask).sum() / denom


def forward(P, cfg: OracleConfig, imgs, tgts, bool_masked_pos=None, valid=None, seg_type=None,
            merge_between_batch: int = -1, drop_scales=None, return_pred_image: bool = False):
    """Painter.forward (models_painter.py:464-472) / SegGPT.forward (models_seggpt.py:471-479).
    -> (loss, patchify(pred), bool_masked_pos[bool])."""
    L = cfg.grid[0] * cfg.grid[1]
    if bool_masked_pos is None:
        bool_masked_pos = torch.zeros((imgs.shape[0], L), dtype=torch.bool)
    else:
        bool_masked_pos = bool_masked_pos.flatten(1).to(torch.bool)
    latent = forward_encoder(P, cfg, imgs, tgts, bool_masked_pos, seg_type, merge_between_batch, drop_scales)
    pred = forward_decoder(P, cfg, latent)
    loss = forward_loss(cfg, pred, tgts, bool_masked_pos, valid)
    if return_pred_image:
        return loss, patchify(pred, cfg.patch_size), bool_masked_pos, pred
    return loss, patchify(pred, cfg.patch_size), bool_masked_pos


# ----------------------------------------------------------------------------- parameters / inputs
def param_shapes(cfg: OracleConfig) -> "Dict[str, Tuple[int, ...]]":
    """The checkpoint ABI, in state_dict order."""
    D, p = cfg.embed_dim, cfg.patch_size
    Hp, Wp = cfg.grid
    hd = D // cfg.num_heads
    hid = int(D * cfg.mlp_ratio)
    npos = (cfg.pretrain_img_size // p) ** 2 + (1 if cfg.pretrain_use_cls_token else 0)
    s = {}
    s["mask_token"] = (1, 1, 1, D)
    s["segment_token_x"] = (1, 1, 1, D)
    s["segment_token_y"] = (1, 1, 1, D)
    if cfg.seggpt:
        s["type_token_cls"] = (1, 1, 1, D)
        s["type_token_ins"] = (1, 1, 1, D)
    s["pos_embed"] = (1, npos, D)
    s["patch_embed.proj.weight"] = (D, 3, p, p)
    s["patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + "attn.rel_pos_h"] = (2 * Hp - 1, hd)
        s[b + "attn.rel_pos_w"] = (2 * Wp - 1, hd)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.qkv.bias"] = (3 * D,)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (hid, D)
        s[b + "mlp.fc1.bias"] = (hid,)
        s[b + "mlp.fc2.weight"] = (D, hid)
        s[b + "mlp.fc2.bias"] = (D,)
    s["norm.weight"] = (D,)
    s["norm.bias"] = (D,)
    de = cfg.decoder_embed_dim
    s["decoder_embed.weight"] = (p * p * de, 4 * D)
    s["decoder_embed.bias"] = (p * p * de,)
    s["decoder_pred.0.weight"] = (de, de, 3, 3)
    s["decoder_pred.0.bias"] = (de,)
    s["decoder_pred.1.weight"] = (de,)
    s["decoder_pred.1.bias"] = (de,)
    s["decoder_pred.3.weight"] = (3, de, 1, 1)
    s["decoder_pred.3.bias"] = (3,)
    return s


def random_params(cfg: OracleConfig, seed: int = 1, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded parameters that exercise EVERY term (the reference init leaves
    rel_pos_*, biases and LN affine at 0/1, which hides those paths).  Recipe (part of the golden
    contract, do not change): one torch.Generator(seed); parameters in param_shapes() order;
    matrices / conv kernels ~ N(0, std_w) with std_w = 0.02 except conv kernels which use
    1/sqrt(fan_in); LayerNorm weights = 1 + N(0, 0.1); everything else N(0, 0.02)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shape in param_shapes(cfg).items():
        t = torch.randn(shape, generator=g, dtype=torch.float32)
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name in ("norm.weight", "decoder_pred.1.weight"):
            t = 1.0 + 0.1 * t
        elif name in ("decoder_pred.0.weight", "decoder_pred.3.weight", "patch_embed.proj.weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            t = t / math.sqrt(fan_in)
        elif name.endswith("rel_pos_h") or name.endswith("rel_pos_w"):
            t = 0.2 * t                       # large enough that a wrong bias index is visible
        else:
            t = 0.02 * t
        P[name] = t.to(dtype)
    return P


def synthetic_batch(cfg: OracleConfig, batch: int, seed: int = 1234, mask_kind: str = "half",
                    dtype=torch.float32):
    """SURVEY.md section 8d 'Synthetic inputs': normalised uniform images, bottom-half or seeded
    random 50% mask, valid = 1."""
    g = torch.Generator().manual_seed(seed)
    H, W = cfg.img_size
    mean = torch.tensor(IMAGENET_MEAN)[None, :, None, None]
    std = torch.tensor(IMAGENET_STD)[None, :, None, None]
    imgs = ((torch.rand(batch, 3, H, W, generator=g) - mean) / std).to(dtype)
    tgts = ((torch.rand(batch, 3, H, W, generator=g) - mean) / std).to(dtype)
    L = cfg.grid[0] * cfg.grid[1]
    mask = torch.zeros(batch, L, dtype=torch.int32)
    if mask_kind == "half":
        mask[:, L // 2:] = 1
    else:
        for b in range(batch):
            perm = torch.randperm(L, generator=g)
            mask[b, perm[: L // 2]] = 1
    valid = torch.ones(batch, 3, H, W, dtype=dtype)
    return imgs, tgts, mask, valid


def tiny_config(seggpt: bool = False) -> OracleConfig:
    """Painter(img_size=(128,64), embed_dim=64, depth=24, num_heads=4,
    decoder_embed_dim=8) -> L=32, rel_pos (15,16)/(7,16).  pretrain grid 14x14 -> bicubic to 8x4."""
    return OracleConfig(img_size=(128, 64), embed_dim=64, depth=24, num_heads=4,
                        decoder_embed_dim=8, seggpt=seggpt)


def small_config(seggpt: bool = False) -> OracleConfig:
    """Smallest shape the HIP path supports (head_dim 64, decoder_embed_dim 64, 8x4 token grid): the GPU tests'
    fast whole-model case.  Golden vectors for it come from the unmodified reference (tests/golden/painter_small.npz)."""
    return OracleConfig(img_size=(128, 64), embed_dim=128, depth=24, num_heads=2, decoder_embed_dim=64, seggpt=seggpt)


def generalised_taps(depth: int) -> Tuple[int, ...]:
    """depth/4*k - 1: the reference's [5, 11, 17, 23] (models_painter.py:416) at depth 24, extended to other depths (SURVEY.md 8d note H:
    with the hard-coded list a depth-32 model would leave blocks 24-31 without gradient)."""
    return tuple(depth // 4 * k - 1 for k in range(1, 5))


def h14_small_config(depth: int = 16) -> OracleConfig:
    """Smallest ViT-H/14-shaped case the HIP path takes: patch 14, head_dim 80 (embed 160 / 2 heads), 8 x 4 tokens, the 16 x 16
    pre-training position grid.  depth 24 = the configuration the unmodified reference can run (tests/golden/painter_h14.npz);
    other depths use the generalised taps."""
    return OracleConfig(img_size=(112, 56), patch_size=14, embed_dim=160, depth=depth, num_heads=2, decoder_embed_dim=64,
                        taps=generalised_taps(depth))


def h14_grid_config(which: str) -> OracleConfig:
    """ViT-H/14 ARITHMETIC (patch 14, head_dim 80 = embed 160 / 2 heads, depth 24 so that the unmodified reference can run it) on token
    grids wide enough for the kernels bench.py --model vit_huge times (csrc/attn2.hip, head_dim-80 instantiation: key rows of 12..32
    tokens; h14_small_config's 8 x 4 grid runs on the generic kernels):
      "w12": 336 x 168 -> 24 x 12 tokens (key rows of 12);  "w32": 896 x 448 -> 64 x 32 tokens, ViT-H/14's own grid (key row = one 32-key
      tile, the WP32 code path).  Golden vectors: tests/golden/painter_h14_grids.npz (make_golden.py --h14-grids)."""
    img = {"w12": (336, 168), "w32": (896, 448)}[which]
    return OracleConfig(img_size=img, patch_size=14, embed_dim=160, depth=24, num_heads=2, decoder_embed_dim=64, taps=generalised_taps(24))


def vit_huge_config() -> OracleConfig:
    """BASELINE configs[4] / SURVEY.md 8d config 5: ViT-H/14 through the class constructor (models_painter.py:241-266) -- patch 14,
    embed 1280, depth 32, 16 heads (head_dim 80), 64 x 32 tokens.  NOT a reference factory; taps generalised."""
    return OracleConfig(img_size=(896, 448), patch_size=14, embed_dim=1280, depth=32, num_heads=16, decoder_embed_dim=64,
                        taps=generalised_taps(32))


def vit_large_config(seggpt: bool = False) -> OracleConfig:
    """Painter/models_painter.py:476-487 / models_seggpt.py:483-494."""
    return OracleConfig(seggpt=seggpt)
