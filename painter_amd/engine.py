"""Forward / backward orchestration of the Painter / SegGPT ViT hot path over the C ABI.

This is the host-side mirror of Painter.forward_encoder / forward_decoder / forward_loss
(Painter/models_painter.py:385-472) and SegGPT's variants (SegGPT/SegGPT_inference/models_seggpt.py:391-479):
the control flow, buffer ownership and kernel order live here; every FLOP happens in libpainter_hip.so.
Backward is written by hand -- nothing is delegated to torch autograd or ATen.

Data layout in HBM (per rank):
  residual stream x        fp32 [B'*L, D]           B' = 2B for blocks 0..merge_idx, B afterwards
  GEMM operands / acts     T    (bf16 or fp32)      ln out [R,D], qkv [R,3D], attn out [R,D], fc1 act [R,4D] (+ gelu aux: fp32 pre-activation, or the 8-bit gelu' code)
  tap concat               T    [B*L, 4D]           LayerNorm of the 4 taps written straight into column slices
  decoder image            T    NHWC [B, H, W, 64]  pixel shuffle fused into decoder_embed's epilogue
  pred / loss              fp32 NCHW [B,3,H,W], [2]
"""
import os

import torch

from . import hostmath, ops
from ._lib import EPI_BIAS, EPI_BIAS_F32, EPI_BIAS_RESID


# diagnostics for the two-stream backward: keep every tensor the side stream reads alive until the end of the backward (no
# allocator reuse), or serialise the two streams (no concurrency)
_DBG_KEEP = os.environ.get("PAINTER_AMD_DEBUG_KEEPALIVE", "0") == "1"
_DBG_SERIAL = os.environ.get("PAINTER_AMD_DEBUG_SERIAL", "0") == "1"
_SIDE_EXTRA = os.environ.get("PAINTER_AMD_SIDE_EXTRA", "1") != "0"     # rel-pos and conv weight gradients on the side stream too
_DBG_TRACE = os.environ.get("PAINTER_AMD_DEBUG_TRACE", "0") == "1"     # checksums of the backward's intermediates -> HotPath.trace


# A/B switches of two round-5 / round-4 arrangements (tools/step_engine_ab.py toggles the module globals in one process):
#   _ATTN_PREP   "fused": the dQ kernel computes Delta itself (no launch);  "launch": one pa_attn_bwd_prep launch per block (round 4)
#   _FC1_COLSUM  "epilogue": fc1's bias gradient from the fc2 data-gradient GEMM's epilogue (round 4);  "separate": a column-sum pass on the side stream
_ATTN_PREP = os.environ.get("PAINTER_AMD_ATTN_PREP", "fused")
_FC1_COLSUM = os.environ.get("PAINTER_AMD_FC1_COLSUM", "epilogue")
_SIDE_STREAM = os.environ.get("PAINTER_AMD_SIDE_STREAM", "1") != "0"
_SIDE_PRIORITY = int(os.environ.get("PAINTER_AMD_SIDE_PRIORITY", "0"))
_RELPOS_PACK = os.environ.get("PAINTER_AMD_RELPOS_PACK", "batch")      # "per_block": the per-block pack launches of rounds 4 - 5 (A/B only)
_configured = False
# sizing of the parameter-gradient kernels when they run on the side stream, beside the data-gradient chain (0 = stand-alone sizing)
WGRAD_SIDE_TARGET = 96       # round 5 re-sweep on the lighter side stream (tools/step_knob_ab.py, profiles/r05_wgrad_side_target_sweep.log): 96 -> 53.08, 128 -> 53.42, 160 -> 54.09, 192 -> 54.47 ms/step
RELPOS_SIDE_SPLITS = 8


def _configure_library():
    """Process-wide tuning knobs of libpainter_hip.so, set ONCE from the environment (they are globals of the library: setting them per
    module instance would let the last constructed model decide for every model of the process)."""
    global _configured
    if _configured:
        return
    from ._lib import lib
    lib.pa_debug_set(6, RELPOS_SIDE_SPLITS if _SIDE_STREAM else 0)       # K splits of the rel-pos table-gradient GEMM beside the main chain: round 2: 16 -> 54.54, 4 -> 54.35, 2 -> 55.0 ms/step; round 3 (tools/knob_sweep.py): 2 -> 56.18, 4 -> 55.10, 8 -> 54.89
    lib.pa_debug_set(3, WGRAD_SIDE_TARGET if _SIDE_STREAM else 0)     # wgrad GEMM workgroup target (gemm.hip: wgrad_fast_splits); round-2 sweep: 64 -> 59.7, 96 -> 57.7, 128 -> 57.7, 192 -> 58.9, 256 -> 59.2 ms/step
    _configured = True


class HotPathConfig:
    def __init__(self, img_size, patch_size, embed_dim, depth, num_heads, mlp_ratio, decoder_embed_dim,
                 pretrain_img_size, pretrain_use_cls_token, use_rel_pos, ln_eps, loss_func, seggpt, drop_path_rate, taps=None):
        self.H, self.W = img_size
        self.P = patch_size
        self.D = embed_dim
        self.depth = depth
        self.heads = num_heads
        self.hidden = int(embed_dim * mlp_ratio)
        self.dec = decoder_embed_dim
        self.Hp, self.Wp = self.H // patch_size, self.W // patch_size
        self.L = self.Hp * self.Wp
        self.src = pretrain_img_size // patch_size
        self.cls = 1 if pretrain_use_cls_token else 0
        self.use_rel_pos = use_rel_pos
        self.ln_eps = ln_eps
        self.loss_func = loss_func
        self.seggpt = seggpt
        self.merge_idx = 2                                   # models_painter.py:408
        # models_painter.py:416 hard-codes [5, 11, 17, 23]; another schedule has to be asked for explicitly (Painter(feature_taps=...))
        self.taps = [5, 11, 17, 23] if taps is None else [int(t) for t in taps]
        self.dpr = hostmath.drop_path_rates(drop_path_rate, depth)
        self.scale = (embed_dim // num_heads) ** -0.5
        self.check()

    def check(self):
        """The HIP path's structural requirements (fail loudly, there is no fallback)."""
        hd = self.D // self.heads
        err = []
        if hd not in (64, 80) or hd * self.heads != self.D:
            err.append("head_dim must be 64 (the reference factories) or 80 (ViT-H/14, BASELINE configs[4]); got %d" % hd)
        if self.dec != 64: err.append("decoder_embed_dim must be 64 (got %d)" % self.dec)
        if self.L % 32 or self.Hp % 4 or self.Wp % 4: err.append("token grid %dx%d must have Hp,Wp %% 4 == 0 and L %% 32 == 0" % (self.Hp, self.Wp))
        if self.H % self.P or self.W % self.P or self.W % 4: err.append("image %dx%d must be a whole number of %d-pixel patches, width a multiple of 4" % (self.H, self.W, self.P))
        if self.D % 8 or self.hidden % 8: err.append("embed/hidden dims must be multiples of 8")
        if not self.use_rel_pos: err.append("use_rel_pos=False is not built (the reference factories always enable it)")
        if self.H != 2 * self.W: err.append("img_size must be (2W, W) (patchify asserts H == 2W, models_painter.py:361)")
        if self.merge_idx >= self.depth or len(set(self.taps)) != 4 or sorted(self.taps) != self.taps or self.taps[-1] != self.depth - 1:
            err.append("depth %d is incompatible with the feature taps %s (four increasing blocks, the last one the last block; the reference's "
                       "hard-coded [5, 11, 17, 23] only fits depth 24 -- pass feature_taps=..., e.g. depth/4*k - 1)" % (self.depth, self.taps))
        if self.merge_idx in self.taps or min(self.taps) < self.merge_idx:
            # the backward handles a block that is a feature tap OR the stream merge, and taps are taken on the merged stream
            err.append("depth %d puts a feature tap at or before the stream merge (block %d)" % (self.depth, self.merge_idx))
        if err:
            raise NotImplementedError("painter_amd HIP path: " + "; ".join(err))


class _Saved:
    pass


class HotPath:
    """Owns the per-module device constants and runs forward / backward."""

    def __init__(self, cfg: HotPathConfig, compute_dtype):
        self.cfg = cfg
        self.T = compute_dtype
        self._M = None
        self._wcache = {}
        self._pcache = {}
        self._rcache = {}
        self._side = {}
        self._lnws = {}                # per device: ring of workspaces for the deferred LayerNorm-backward reductions (ln_workspace)
        # parameter-gradient kernels (dW = dY^T.X, bias column sums) are off the backward's critical path: they go to a second HIP
        # stream (+3.5 % at B=8; PAINTER_AMD_SIDE_STREAM=0 turns it off).  This mode exposed two things, both fixed: a cross-stream
        # allocator hazard on the gradient buckets, and SLP-packed fp32 VALU code mis-computing beside another kernel's MFMA
        # workgroups (build.py: -fno-slp-vectorize); DESIGN.md section 6.
        self.use_side_stream = _SIDE_STREAM
        _configure_library()

    def side_stream(self, device):
        s = self._side.get(device)
        if s is None:
            # PAINTER_AMD_SIDE_PRIORITY: HIP stream priority of the parameter-gradient stream (0 = default; what the runtime accepts
            # is clamped by torch / HIP; tools/prio_ab.py measures it against a high-priority main stream)
            s = torch.cuda.Stream(device=device, priority=getattr(self, "side_priority", _SIDE_PRIORITY))
            self._side[device] = s
        return s

    LN_RING = 6

    def ln_workspace(self, dev, nbytes, main, side):
        """A workspace for one deferred LayerNorm-backward reduction (ops.layernorm_bwd(defer=True, ws=...)), from a ring of LN_RING
        buffers per device that is reused across blocks and steps -- every call used to allocate 12.6 MB that the allocator could not
        hand out again before the side stream had caught up (up to ~0.6 GB of extra cached memory per step when it lagged).  Ordering:
        the main-stream kernel that refills a buffer waits for the event recorded behind the side-stream reduction that last read it
        (six launches earlier: practically never a stall).  -> (buffer, done) -- call done() after enqueueing the reduction."""
        ring = self._lnws.setdefault(dev, {"bufs": [None] * self.LN_RING, "events": [None] * self.LN_RING, "i": 0})
        k = ring["i"] % self.LN_RING
        ring["i"] += 1
        buf = ring["bufs"][k]
        if buf is None or buf.numel() < nbytes:
            if buf is not None and side is not None:
                buf.record_stream(side)            # the replaced buffer may still be read by a reduction in flight
            buf = ring["bufs"][k] = torch.empty((int(nbytes),), dtype=torch.uint8, device=dev)
        ev = ring["events"][k]
        if ev is not None and side is not None:
            main.wait_event(ev)

        def done():
            if side is not None:
                e = ring["events"][k] or torch.cuda.Event()
                e.record(side)
                ring["events"][k] = e
        return buf, done

    # ------------------------------------------------------------------ constants / casts
    def pos_operator(self, device):
        """-> (sparse rows of M, sparse rows of M^T) on the device: the constant bicubic resize of the pre-training position grid."""
        if self._M is None or self._M[0][0].device != device:
            c = self.cfg
            M = hostmath.abs_pos_operator(c.src, c.Hp, c.Wp)
            fwd, bwd = hostmath.sparse_rows(M), hostmath.sparse_rows(M.T)
            dev = lambda t: (torch.from_numpy(t[0]).to(device), torch.from_numpy(t[1]).to(device))
            self._M = (dev(fwd), dev(bwd))
        return self._M

    def w(self, name, P):
        """T-typed copy of a weight matrix (K9: cast once per optimizer step, keyed on the tensor version)."""
        t = P[name]
        if self.T == torch.float32:
            return t.reshape(t.shape[0], -1)
        key = (name, t.data_ptr())
        ent = self._wcache.get(key)
        if ent is None or ent[0] != t._version or ent[1].device != t.device:
            buf = ent[1] if ent is not None and ent[1].device == t.device else torch.empty((t.shape[0], t.numel() // t.shape[0]), dtype=self.T, device=t.device)
            ops.cast_bf16(t, buf)
            self._wcache[key] = (t._version, buf)
            return buf
        return ent[1]

    def w_patch(self, P):
        """The patch-embed conv weight as the T [D, Kp] operand pa_patch_embed_fwd takes (Kp = 3*P*P rounded up to 8, zero padded).
        For P % 8 == 0 that is the ordinary T copy; otherwise (P = 14) a packed copy, cached on the parameter version like w()."""
        c = self.cfg
        if c.P % 8 == 0:
            return self.w("patch_embed.proj.weight", P)
        t = P["patch_embed.proj.weight"]
        key = ("patch_embed.proj.weight#packed", t.data_ptr())
        ent = self._pcache.get(key)
        if ent is None or ent[0] != t._version or ent[1].device != t.device:
            buf = ops.patch_weight_pack(t, self.T, c.P, out=ent[1] if ent is not None and ent[1].device == t.device else None)
            self._pcache = {key: (t._version, buf)}
            return buf
        return ent[1]

    def relpos(self, pre, P, transposed):
        """Rcat / Rcat^T operand of block `pre`.  Every block's pair is packed by ONE launch (pa_relpos_pack_batch, round 6) whenever the
        version of the asked-for block's tables has moved -- i.e. once per optimizer step in plain training (rounds 1 - 5: one launch per
        block and orientation, 48 per step; cached per parameter version since round 4, which only helped loops that leave the parameters
        alone: forward + backward timing, evaluation, accumulation micro-steps).  Keyed on addresses and tensor versions like every cache
        here: a write that bumps no version (p.data.copy_, an in-place collective) needs HotPath.invalidate().  The first call (and any call after
        the tables moved in memory) uploads the 2 * depth addresses: run one eager forward before capturing a hipGraph, as every caller here does."""
        c = self.cfg
        i = int(pre.split(".")[1])
        rh, rw = P[pre + "attn.rel_pos_h"], P[pre + "attn.rel_pos_w"]
        st = self._rcache
        if _RELPOS_PACK == "per_block":            # rounds 4 - 5 (A/B: tools/step_engine_ab.py): one launch per block and orientation
            key, ver = (pre, transposed), (rh.data_ptr(), rw.data_ptr(), rh._version, rw._version)
            if key not in st or st[key][0] != ver:
                st[key] = (ver, (ops.relpos_pack_t if transposed else ops.relpos_pack)(rh, rw, c.Hp, c.Wp, self.T))
            return st[key][1]
        if st and st["dev"] == rh.device and st["ptr"][i] == (rh.data_ptr(), rw.data_ptr()) and st["ver"][i] == (rh._version, rw._version):
            return (st["rcatT"] if transposed else st["rcat"])[i]
        hs = [P["blocks.%d.attn.rel_pos_h" % k] for k in range(c.depth)]
        ws = [P["blocks.%d.attn.rel_pos_w" % k] for k in range(c.depth)]
        hd = rh.shape[1]
        for t in hs + ws:
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == rh.device and t.shape[1] == hd
        ptr = [(h.data_ptr(), w.data_ptr()) for h, w in zip(hs, ws)]
        same = bool(st) and st["dev"] == rh.device and st["ptr"] == ptr
        tabs = st["tabs"] if same else torch.tensor([q[0] for q in ptr] + [q[1] for q in ptr], dtype=torch.int64).to(rh.device)
        rcat, rcatT = ops.relpos_pack_batch(tabs, c.depth, c.Hp, c.Wp, hd, self.T, rcat=st["rcat"] if same else None, rcatT=st["rcatT"] if same else None)
        st.clear()
        st.update(dev=rh.device, ptr=ptr, ver=[(h._version, w._version) for h, w in zip(hs, ws)], tabs=tabs, rcat=rcat, rcatT=rcatT)
        return (rcatT if transposed else rcat)[i]

    def relpos_stale(self):
        """Mark the packed rel-pos tables stale, as an optimizer update of the tables does through their version counters (buffers and the
        address table stay): bench.py calls it before every timed step so that forward + backward is timed as training runs it."""
        st = self._rcache
        if _RELPOS_PACK == "per_block":
            st.clear()
        elif st:
            st["ver"] = [None] * len(st["ver"])

    def invalidate(self):
        """Drop every cached operand copy (bf16 weights, packed patch weight, Rcat / Rcat^T): call after writing parameters in a way
        that leaves their version counters alone."""
        self._wcache.clear()
        self._pcache.clear()
        self._rcache.clear()

    def shadow_buffers(self):
        """{parameter data_ptr: cached bf16 copy} -- lets painter_amd.optim.AdamW refresh the copies inside its update pass."""
        return {key[1]: ent[1] for key, ent in self._wcache.items()}

    def mark_fresh(self, params):
        """The optimizer has rewritten these parameters AND their bf16 copies: adopt the new versions."""
        by_ptr = {p.data_ptr(): p for p in params}
        for key, ent in list(self._wcache.items()):
            p = by_ptr.get(key[1])
            if p is not None:
                self._wcache[key] = (p._version, ent[1])

    # ------------------------------------------------------------------ forward
    def forward(self, P, imgs, tgts, mask_u8, valid, seg_type=None, merge_between_batch=-1, drop_scales=None, need_grad=True):
        c, T = self.cfg, self.T
        dev = imgs.device
        B = imgs.shape[0]
        L, D = c.L, c.D
        S = _Saved()
        S.B, S.need_grad = B, need_grad
        S.imgs, S.tgts, S.mask, S.valid = imgs, tgts, mask_u8, valid
        S.drop = drop_scales
        S.seg_type = seg_type if c.seggpt else None
        pe = P["pos_embed"][0, c.cls:]
        pos = ops.pos_fwd(self.pos_operator(dev)[0], pe, L, D)
        tok_args = (P["patch_embed.proj.bias"], P["mask_token"], P["segment_token_x"], P["segment_token_y"], pos, mask_u8,
                    P.get("type_token_cls") if c.seggpt else None, P.get("type_token_ins") if c.seggpt else None,
                    seg_type if c.seggpt else None)
        S.cols = None
        if ops.patch_cols_ok(T, B, L, c.P, D):           # bf16, P % 8 == 0: materialised im2col operand + the 256 x 256 GEMM (kept for the weight gradient)
            cols = ops.patch_im2col(imgs, tgts, B, c.Hp, c.Wp, c.P)
            x = ops.patch_embed_fwd_cols(cols, self.w_patch(P), *tok_args, B, L, D)
            if need_grad:
                S.cols = cols
        else:
            x = ops.patch_embed_fwd(T, imgs, tgts, self.w_patch(P), *tok_args, B, c.Hp, c.Wp, c.P, D)
        concat = torch.empty((B * L, 4 * D), dtype=T, device=dev)
        S.blocks, S.taps = [], []
        Bc = 2 * B
        for i in range(c.depth):
            pre = "blocks.%d." % i
            R = Bc * L
            merge = 0
            if c.seggpt and merge_between_batch >= 0 and i >= merge_between_batch:
                merge = 1 if c.merge_idx >= i else 2
            ds_a, ds_m = (None, None) if drop_scales is None else drop_scales[i]
            ln1, mean1, rstd1 = ops.layernorm_fwd(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], c.ln_eps, T)
            qkv = ops.linear_fwd(ln1, self.w(pre + "attn.qkv.weight", P), P[pre + "attn.qkv.bias"], EPI_BIAS)
            rcat = self.relpos(pre, P, False)
            ao, lse, atab = ops.attn_fwd(qkv, rcat, Bc, L, c.heads, c.Hp, c.Wp, c.scale, need_tables=True) if need_grad else \
                ops.attn_fwd(qkv, rcat, Bc, L, c.heads, c.Hp, c.Wp, c.scale) + (None,)
            group = 0
            if merge > 0:
                # x1 = x0 + s_a * ens(proj(...)) (models_seggpt.py:207-238).  Differentiable like the reference's Block.forward: mean over
                # the group + broadcast is its own adjoint, the backward applies the same operator to the (DropPath-scaled) branch
                # gradient.  The reference itself only runs the ensemble under @torch.no_grad (seggpt_engine.py:26); with DropPath
                # factors (train mode) the scaled form is composed from the same kernels: ens alone, the row-scale kernel, a plain add
                # (= the ensemble kernel with groups of one).
                a = ops.linear_fwd(ao, self.w(pre + "attn.proj.weight", P), P[pre + "attn.proj.bias"], EPI_BIAS_F32)
                group = Bc // 2 if merge == 1 else Bc
                if ds_a is None:
                    x1 = ops.ensemble_resid(x, a, Bc, group, L, D)
                else:
                    e = ops.ensemble_resid(torch.zeros_like(x), a, Bc, group, L, D)
                    x1 = ops.ensemble_resid(x, ops.scale_cast(torch.float32, e, ds_a, L), Bc, 1, L, D)
                    del e
            else:
                x1 = ops.linear_fwd(ao, self.w(pre + "attn.proj.weight", P), P[pre + "attn.proj.bias"], EPI_BIAS_RESID,
                                    resid=x, rowscale=ds_a, rows_per_sample=L)
            ln2, mean2, rstd2 = ops.layernorm_fwd(x1, P[pre + "norm2.weight"], P[pre + "norm2.bias"], c.ln_eps, T)
            act, gaux = ops.linear_gelu(ln2, self.w(pre + "mlp.fc1.weight", P), P[pre + "mlp.fc1.bias"], need_aux=need_grad)
            x2 = ops.linear_fwd(act, self.w(pre + "mlp.fc2.weight", P), P[pre + "mlp.fc2.bias"], EPI_BIAS_RESID,
                                resid=x1, rowscale=ds_m, rows_per_sample=L)
            if need_grad:
                S.blocks.append((x, mean1, rstd1, ln1, qkv, rcat, ao, lse, x1, mean2, rstd2, ln2, gaux, act, Bc, atab, group))
            x = x2
            if i == c.merge_idx:
                Bc = B
                x = ops.merge_fwd(x, B * L, D)
            if i in c.taps:
                k = c.taps.index(i)
                _, mt, rt = ops.layernorm_fwd(x, P["norm.weight"], P["norm.bias"], c.ln_eps, T, out=concat[:, k * D:(k + 1) * D])
                if need_grad:
                    S.taps.append((x, mt, rt))
        E = ops.linear_pixshuf(concat, self.w("decoder_embed.weight", P), P["decoder_embed.bias"], B, c.Hp, c.Wp, c.P, c.dec)
        w3r, wf = ops.conv3x3_pack(P["decoder_pred.0.weight"], T)
        w1 = P["decoder_pred.3.weight"].reshape(3, c.dec)
        pred, y3 = ops.decoder_tail_fwd(E, w3r, P["decoder_pred.0.bias"], P["decoder_pred.1.weight"], P["decoder_pred.1.bias"],
                                        w1, P["decoder_pred.3.bias"], 1e-6, save_y3=need_grad)
        loss_out = ops.loss_fwd(pred, tgts, valid, mask_u8, c.P, ignore_rule=not c.seggpt,
                                eps_den=0.0 if c.seggpt else 1e-2, kind=c.loss_func)
        pred_patch = ops.patchify(pred, c.Hp, c.Wp, c.P)
        if need_grad:
            S.concat, S.E, S.y3, S.pred, S.loss_out, S.wf = concat, E, y3, pred, loss_out, wf
        return loss_out, pred, pred_patch, S

    # ------------------------------------------------------------------ backward
    def backward(self, P, S, dloss, sync=None):
        """-> {param name: fp32 grad}.  dloss: 0-d / [1] fp32 device tensor (may carry a GradScaler factor).
        sync: optional painter_amd.parallel.GradSync; buckets are handed over as soon as they are enqueued.

        Two HIP streams: the data-gradient chain (dgrad GEMMs, attention backward, LayerNorm backward) runs on the caller's
        stream; every weight/bias gradient of an nn.Linear (wgrad GEMM + slab reduction + column sum) is enqueued on a side
        stream behind an event, because nothing downstream in the backward consumes it."""
        c, T = self.cfg, self.T
        B, L, D = S.B, c.L, c.D
        dev = S.imgs.device
        G = {}
        main = torch.cuda.current_stream(dev)
        side = self.side_stream(dev) if self.use_side_stream else None
        keep = []
        self.trace = []

        def tr(name, t):
            if _DBG_TRACE:
                self.trace.append((name, t.detach().double().abs().sum()))

        filt = getattr(self, "side_filter", None)          # diagnostics: {"dec","fc2","fc1","proj","qkv"} subsets, "nocolsum", "nowgrad"

        def param_grads(wname, bname, dy, x, bout=None):
            """G[wname] = dy^T.x, G[bname] = colsum(dy) (into `bout`, a slice of the block's flat small-gradient buffer, when given)
            -- on the side stream when enabled."""
            tag = "dec" if wname.startswith("decoder") else wname.split(".")[-2]
            if side is None or (filt is not None and tag not in filt):
                G[wname] = ops.linear_wgrad(dy, x)
                if bname not in G:                 # (fc2 / proj biases: already summed by the LayerNorm backward that produced dy)
                    G[bname] = ops.colsum(dy, out=bout)
                return
            if filt is not None and "nocolsum" in filt:
                G[bname] = ops.colsum(dy, out=bout)
            if filt is not None and "nowgrad" in filt:
                G[wname] = ops.linear_wgrad(dy, x)
            side.wait_stream(main)                 # dy (and x) are enqueued on main
            with torch.cuda.stream(side):
                if wname not in G:
                    G[wname] = ops.linear_wgrad(dy, x)
                if bname not in G:
                    G[bname] = ops.colsum(dy, out=bout)
            dy.record_stream(side)                 # the allocator must not hand these out again before the side stream is done
            x.record_stream(side)
            if _DBG_KEEP:
                keep.extend([dy, x])
            if _DBG_SERIAL:
                main.wait_stream(side)

        def on_side(fn, *inputs):
            """Run a parameter-gradient computation that nothing downstream consumes on the side stream (inputs: main-stream tensors)."""
            if side is None or not _SIDE_EXTRA:
                return fn()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                r = fn()
            for t in inputs:
                t.record_stream(side)
            if _DBG_SERIAL:
                main.wait_stream(side)
            return r

        def ready(names, flat=None):
            """flat: one contiguous fp32 buffer that already holds every small gradient of `names` (they are views of it): it is
            exchanged as ONE message in place -- no flattening copy on the side stream."""
            if sync is None:
                return
            if side is None:
                sync.ready(G, names, flat=flat)
                return
            side.wait_stream(main)                 # the bucket's gradients come from both streams
            for n in names:                        # main-stream allocations (LayerNorm / rel-pos / tail gradients) are read by the
                G[n].record_stream(side)           # side stream's exchange: keep the allocator from recycling them early
            with torch.cuda.stream(side):
                sync.ready(G, names, flat=flat)

        dpred = ops.loss_bwd(S.pred, S.tgts, S.valid, S.mask, dloss, S.loss_out, c.P, c.loss_func)
        w1 = P["decoder_pred.3.weight"].reshape(3, c.dec)
        dy3, tg = ops.decoder_tail_bwd_pointwise(dpred, S.y3, P["decoder_pred.1.weight"], P["decoder_pred.1.bias"], w1, 1e-6)
        G["decoder_pred.1.weight"] = tg[0:64]
        G["decoder_pred.1.bias"] = tg[64:128]
        G["decoder_pred.3.weight"] = tg[128:320].reshape(3, c.dec, 1, 1)
        G["decoder_pred.3.bias"] = tg[320:323]
        npix = B * c.H * c.W
        G["decoder_pred.0.weight"], G["decoder_pred.0.bias"] = on_side(
            lambda: (ops.conv3x3_wgrad(dy3, S.E), ops.colsum(dy3.view(npix, c.dec))), dy3, S.E)
        dE = ops.conv3x3_dgrad_unshuffle(dy3, S.wf, B, c.Hp, c.Wp, c.P)
        del dy3
        param_grads("decoder_embed.weight", "decoder_embed.bias", dE, S.concat)
        dconcat = ops.linear_dgrad(dE, self.w("decoder_embed.weight", P))
        tr("dE", dE); tr("dconcat", dconcat)
        del dE
        ready(["decoder_embed.weight", "decoder_embed.bias"])
        ready([n for n in G if n.startswith("decoder_pred.")])
        dnorm = None
        dx = None
        dyT_next = None
        flats = {}

        def block_flat(i_):
            """The flat small-gradient buffer of block i_ (allocated on first use: block i_ + 1's norm1 backward already writes block
            i_'s fc2 bias gradient into it)."""
            if i_ not in flats:
                nrp_, hd_ = rc_shape
                sizes = [2 * D, 2 * D, 3 * D, D, c.hidden, D, nrp_ * hd_]
                fb = torch.empty((sum(sizes),), dtype=torch.float32, device=dev)
                if side is not None:
                    fb.record_stream(side)
                flats[i_] = (fb, dict(zip(("n1", "n2", "qkv", "proj", "fc1", "fc2", "rel"), torch.split(fb, sizes))))
            return flats[i_]

        rc_shape = tuple(S.blocks[-1][5].shape)            # Rcat [NRP, head_dim]: the same for every block
        for i in reversed(range(c.depth)):
            pre = "blocks.%d." % i
            x0, mean1, rstd1, ln1, qkv, rcat, ao, lse, x1, mean2, rstd2, ln2, gaux, act, Bc, atab, ens_group = S.blocks[i]
            S.blocks[i] = None
            R = Bc * L
            ds_a, ds_m = (None, None) if S.drop is None else S.drop[i]
            # the block's small gradients (LayerNorm affine, the four biases, the rel-pos tables) live in ONE flat buffer so that the
            # gradient exchange sends them as one message without a flattening copy: [norm1 g,b | norm2 g,b | qkv.b | proj.b | fc1.b | fc2.b | d rcat]
            nrp, hd = rcat.shape
            flat, fl = block_flat(i)
            del flats[i]
            # dyT = bf16(ds_m * dx) is emitted by the kernel that produces the final dx of this block's output: the tap
            # LayerNorm backward, block i+1's norm1 backward (dyT_next), or the stream-merge backward
            if i in c.taps:
                k = c.taps.index(i)
                xt, mt, rt = S.taps[k]
                dyT = torch.empty((R, D), dtype=T, device=dev)
                # (the LayerNorm backward kernels also sum the columns of the dxT they emit: dxT is the dY of fc2 / proj, so that
                # sum is the layer's bias gradient -- no separate pass over dxT)
                dx, gb = ops.layernorm_bwd(dconcat[:, k * D:(k + 1) * D], xt, mt, rt, P["norm.weight"], dres=dx, dx=dx, dxT=dyT,
                                           rowscale=ds_m, rows_per_sample=L, dxT_colsum=fl["fc2"])
                G[pre + "mlp.fc2.bias"] = fl["fc2"]
                dnorm = gb if dnorm is None else _add_(dnorm, gb)
            elif i == c.merge_idx:
                dx, dyT = ops.merge_bwd(T, dx, ds_m, L, B * L, D)
            else:
                dyT = dyT_next
            # ---- MLP branch: x2 = x1 + s_m * fc2(gelu(fc1(LN2(x1))))
            param_grads(pre + "mlp.fc2.weight", pre + "mlp.fc2.bias", dyT, act, fl["fc2"])
            tr("%d.dyT" % i, dyT)
            # (the GEMM's epilogue also sums the columns of the dpre it stores: fc1's bias gradient, no separate pass over [R, 4D])
            if _FC1_COLSUM == "epilogue":
                dpre = ops.linear_dgrad(dyT, self.w(pre + "mlp.fc2.weight", P), gelu_aux=gaux, colsum_out=fl["fc1"])
                G[pre + "mlp.fc1.bias"] = fl["fc1"]
            else:                                  # A/B: param_grads below sums the columns of dpre on the side stream
                dpre = ops.linear_dgrad(dyT, self.w(pre + "mlp.fc2.weight", P), gelu_aux=gaux)
            tr("%d.dpre" % i, dpre)
            param_grads(pre + "mlp.fc1.weight", pre + "mlp.fc1.bias", dpre, ln2, fl["fc1"])
            dln2 = ops.linear_dgrad(dpre, self.w(pre + "mlp.fc1.weight", P))
            tr("%d.dln2" % i, dln2)
            del dpre
            # dyT may still be read by the side stream: the attention branch's dY gets its own buffer
            dyA = torch.empty_like(dyT) if side is not None else dyT
            # (the reduction of the LayerNorm parameter-gradient partials is a parameter gradient too: side stream)
            lnws, lndone = self.ln_workspace(dev, ops.layernorm_bwd_workspace_bytes(R, D), main, side if _SIDE_EXTRA else None)
            dx, fin = ops.layernorm_bwd(dln2, x1, mean2, rstd2, P[pre + "norm2.weight"], dres=dx, dx=dx, dxT=dyA,
                                        rowscale=ds_a, rows_per_sample=L, gb=fl["n2"].view(2, D), dxT_colsum=fl["proj"], defer=True, ws=lnws)
            gb = on_side(fin)
            lndone()
            G[pre + "attn.proj.bias"] = fl["proj"]
            del dyT
            tr("%d.dx_ln2" % i, dx); tr("%d.dyA" % i, dyA); tr("%d.gb2" % i, gb)
            G[pre + "norm2.weight"], G[pre + "norm2.bias"] = gb[0], gb[1]
            # ---- attention branch: x1 = x0 + s_a * proj(attn(LN1(x0)))
            if ens_group > 0:
                # SegGPT feature ensemble (forward: x1 = x0 + s_a * ens(proj(...))): the branch gradient is ens applied to s_a * dx (fp32: the
                # kernel's own type), re-rounded to the operand type.  Column sums are unchanged by a mean + broadcast over samples, so
                # the proj bias gradient the LayerNorm backward already summed (of s_a * dx) stands.
                da = ops.ensemble_resid(torch.zeros_like(dx), dx if ds_a is None else ops.scale_cast(torch.float32, dx, ds_a, L), Bc, ens_group, L, D)
                dyA = da if T == torch.float32 else ops.cast_bf16(da, out=dyA)
                del da
            param_grads(pre + "attn.proj.weight", pre + "attn.proj.bias", dyA, ao, fl["proj"])
            dao = ops.linear_dgrad(dyA, self.w(pre + "attn.proj.weight", P), out=dln2)
            tr("%d.dao" % i, dao)
            del dyA
            rcatT = self.relpos(pre, P, True)
            dqkv, dG = ops.attn_bwd_core(qkv, rcat, rcatT, ao, dao, lse, Bc, L, c.heads, c.Hp, c.Wp, c.scale, tables=atab, prep=_ATTN_PREP)
            drcat = on_side(lambda: ops.attn_bwd_relpos(dG, qkv, nrp, Bc, L, c.heads, c.Hp, c.Wp, out=fl["rel"].view(nrp, hd)), dG, qkv)
            del dG
            tr("%d.dqkv" % i, dqkv)
            nh, nw = 2 * c.Hp - 1, 2 * c.Wp - 1
            G[pre + "attn.rel_pos_h"] = drcat[:nh]
            G[pre + "attn.rel_pos_w"] = drcat[nh:nh + nw]
            param_grads(pre + "attn.qkv.weight", pre + "attn.qkv.bias", dqkv, ln1, fl["qkv"])
            dln1 = ops.linear_dgrad(dqkv, self.w(pre + "attn.qkv.weight", P), out=dao)
            del dqkv
            nxt = i - 1
            dyT_next = None
            if nxt >= 0 and nxt not in c.taps and nxt != c.merge_idx:
                dyT_next = torch.empty((R, D), dtype=T, device=dev)
            ds_next = None if (S.drop is None or nxt < 0) else S.drop[nxt][1]
            cs_next = None
            if dyT_next is not None:               # dyT_next is block nxt's fc2 dY: its column sum goes into block nxt's flat buffer
                cs_next = block_flat(nxt)[1]["fc2"]
                G["blocks.%d.mlp.fc2.bias" % nxt] = cs_next
            lnws, lndone = self.ln_workspace(dev, ops.layernorm_bwd_workspace_bytes(R, D), main, side if _SIDE_EXTRA else None)
            dx, fin = ops.layernorm_bwd(dln1, x0, mean1, rstd1, P[pre + "norm1.weight"], dres=dx, dx=dx, dxT=dyT_next,
                                        rowscale=ds_next if dyT_next is not None else None, rows_per_sample=L, gb=fl["n1"].view(2, D),
                                        dxT_colsum=cs_next, defer=True, ws=lnws)
            gb = on_side(fin)
            lndone()
            G[pre + "norm1.weight"], G[pre + "norm1.bias"] = gb[0], gb[1]
            tr("%d.dx_ln1" % i, dx)
            del x0, ln1, qkv, ao, x1, ln2, gaux, act, atab
            ready([n for n in G if n.startswith(pre)], flat=flat)
        G["norm.weight"], G["norm.bias"] = dnorm[0], dnorm[1]
        # ---- token assembly + patch embed
        dpe, sums = ops.tokens_bwd(T, dx, S.mask, B, L, D)
        if S.cols is not None:
            G["patch_embed.proj.weight"] = ops.linear_wgrad(dpe, S.cols).view(D, 3, c.P, c.P)
        else:
            G["patch_embed.proj.weight"] = ops.patch_embed_wgrad(dpe, S.imgs, S.tgts, B, c.Hp, c.Wp, c.P, D).view(D, 3, c.P, c.P)
        G["patch_embed.proj.bias"] = ops.colsum(dpe)
        dposemb = torch.zeros_like(P["pos_embed"])
        ops.pos_bwd(self.pos_operator(dev)[1], sums[0], sums[1], dposemb[0, c.cls:], c.src * c.src, D)
        G["pos_embed"] = dposemb
        G["segment_token_x"] = ops.colsum(sums[0]).view(1, 1, 1, D)
        G["segment_token_y"] = ops.colsum(sums[1]).view(1, 1, 1, D)
        G["mask_token"] = ops.colsum(sums[2]).view(1, 1, 1, D)
        small_tail = []
        if c.seggpt and S.seg_type is not None:
            # SegGPT's two segmentation-type tokens (models_seggpt.py:415-420: added to every token of both streams of the samples of their
            # type): gradient = sum of dx over those samples' rows -- per-sample row weights (1 where the type matches) through the
            # row-scale kernel, then a column sum.  Only reached when a SegGPT module is differentiated (the reference never does).
            # A token no sample of the batch uses gets NO gradient (None), as under the reference's autograd -- a weight-decaying optimizer
            # then leaves it alone instead of decaying it towards zero.  (One host read of the [B] type vector at the very end of the
            # backward.  With a gradient exchange installed every rank must take part in the same collectives, so zeros are produced
            # there: the reference's DDP wrapper would refuse the unused parameter outright.)
            types_present = set(S.seg_type.reshape(-1).tolist())
            for t_, nm in ((0.0, "type_token_cls"), (1.0, "type_token_ins")):
                if t_ not in types_present and sync is None:
                    continue
                w = (S.seg_type.reshape(-1) == t_).to(torch.float32)
                sel = ops.scale_cast(torch.float32, dx, torch.cat((w, w)).contiguous(), L)
                G[nm] = ops.colsum(sel).view(1, 1, 1, D)
                small_tail.append(nm)
        ready(["norm.weight", "norm.bias", "patch_embed.proj.weight", "patch_embed.proj.bias", "pos_embed",
               "segment_token_x", "segment_token_y", "mask_token"] + small_tail)
        if side is not None:
            if getattr(self, "tail_probe", None) is not None:      # diagnostics (tools/step_tail.py): when each stream ran dry
                self.tail_probe[0].record(main)
                self.tail_probe[1].record(side)
            main.wait_stream(side)                 # every gradient is ordered before whatever the caller enqueues next
        if keep:
            torch.cuda.synchronize()
            keep.clear()
        if sync is not None:
            sync.finish()
        return G


def _add_(a, b):
    """a += b for two small fp32 device tensors through the slab reducer (keeps arithmetic inside the library)."""
    from ._lib import check, lib
    check(lib.pa_slab_reduce(b.data_ptr(), a.data_ptr(), a.numel(), 1, a.numel(), 1, ops.stream()), "pa_slab_reduce")
    return a
