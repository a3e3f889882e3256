"""Data-parallel gradient exchange for the hot path (replaces the DDP reducer installed at
Painter/main_train.py:340).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL on ROCm, xGMI between the 8 GPUs of a node).  The
path shards by samples, so the only exchange is the gradient all-reduce: `GradSync` is called by the engine's
backward as soon as a bucket of parameter gradients has been ENQUEUED (decoder head first, then blocks 23 -> 0, then
the patch/token parameters) and starts asynchronous all-reduce(AVG)s: the weight matrices in place (no flattening copy), the
bucket's small tensors flattened into one message.  RCCL runs it on its own stream, ordered after the producing kernels, so the
exchange of bucket k overlaps the backward kernels of bucket k+1.  `finish()` makes the compute stream wait for all buckets and
hands the averaged gradients back (in place, or as views into the flat message).

Bucket = one transformer block (~12.6 M params = 50 MB fp32) -- large messages because a ring/tree over
point-to-point xGMI links is per-link bandwidth bound, not latency bound; decoder_embed's 268 MB gradient is its own
bucket so it is on the wire while the blocks are still being differentiated.

Gradient accumulation: the exchange runs on EVERY micro-step, exactly like the reference (its DDP model is never put under
no_sync(), engine_train.py:85-90).  What is reduced is the micro-step's own gradient, before autograd adds it to `p.grad`; the sum
of averaged micro-gradients equals the average of the summed ones, so replicas stay identical.  (There is deliberately no switch
to skip micro-steps: skipping would leave the skipped micro-gradients rank-local and replicas would drift apart.)

Replica start: DistributedDataParallel broadcasts rank 0's parameters when it wraps a module; a run that installs `GradSync`
instead of the wrapper must call `broadcast_parameters(model)` once (the reference seeds every rank differently, main_train.py:190).
"""
import os
import warnings

import torch
import torch.distributed as dist


# PAINTER_AMD_DDP_SELFTEST=1: run the exchange even in a 1-rank group (exercises RCCL init, AVG all-reduce, the stream ordering
# and the in-place / flattened paths on a single-GPU box; the result must equal the local gradient)
_SELFTEST = os.environ.get("PAINTER_AMD_DDP_SELFTEST", "0") == "1"


class GradSync:
    """mode "per_block" (default): every ready() call starts its collectives at once -- one message per weight matrix, one flat message
    per transformer block (~125 asynchronous collectives per step, the finest overlap).  mode "coarse": ready() only collects; the
    collected gradients are exchanged as ONE coalesced collective (ncclGroupStart / End around the per-tensor all-reduces: one RCCL
    launch) whenever `bucket_bytes` have accumulated (default 400 MB -> four launches per ViT-L step: decoder_embed + blocks 23-21,
    blocks 20-13, 12-5, the rest) -- fewer, larger launches for the case where ~125 small ones cost more host time and RCCL channel
    set-up than their finer overlap buys.  mode "rs_ag" (round 6): per_block's messages, each exchanged as reduce_scatter_tensor +
    all_gather_into_tensor instead of one all_reduce -- SURVEY.md 8(e)'s all-peer pattern spelled out: every rank reduces 1/W of the message
    from all W - 1 peers at once (over all seven xGMI links of the node) and then hands its shard to all of them, where a ring all-reduce
    walks one link.  RCCL may pick the same schedule inside all_reduce by itself; bench.py --gpus N times all arrangements, so the first
    8-GPU run says whether it does.  Messages whose length W does not divide fall back to all_reduce.  Same arithmetic up to fp32
    summation order (a shard is summed on its owner instead of around a ring); replicas stay bit-identical, every rank receives the same
    shards.  bench.py --gpus N times all three (and the plain DDP wrapper)."""

    def __init__(self, process_group=None, average=True, mode="per_block", bucket_bytes=400 << 20):
        assert mode in ("per_block", "coarse", "rs_ag"), mode
        self.group = process_group
        self.average = average
        self.mode = mode
        self.bucket_bytes = int(bucket_bytes)
        self._pending = []
        self._held, self._held_bytes = [], 0
        self.launches = 0              # collectives (per_block) / coalesced launches (coarse) started since construction: diagnostics

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    BIG = 1 << 20          # gradients of at least this many elements are exchanged in place, without a flattening copy

    def _reduce(self, t):
        if self.mode == "coarse":                  # collected; _flush() starts the exchange
            self._held.append(t)
            self._held_bytes += t.numel() * t.element_size()
            return None, None
        backend = dist.get_backend(self.group)
        W = self.world_size
        if self.mode == "rs_ag" and t.is_contiguous() and t.numel() % W == 0 and (W > 1 or _SELFTEST):      # (self-test: the two RCCL calls on a 1-rank group)
            return self._reduce_scatter_gather(t, backend, W)
        self.launches += 1
        if self.average and backend == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None
        work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work, ((1.0 / self.world_size) if self.average else None)    # gloo (CPU tests) has no AVG

    def _reduce_scatter_gather(self, t, backend, W):
        """In-place all-reduce of `t` as reduce-scatter + all-gather.  Both collectives are enqueued back to back on the group's own stream
        (ProcessGroupNCCL orders a communicator's collectives), so the all-gather reads the shard the reduce-scatter wrote without a host
        wait; the returned handle is the all-gather's.  gloo (the CPU / two-ranks-on-one-GPU tests) has no reduce-scatter: W reduce() calls,
        shard r to rank r -- the same data movement, one message per shard."""
        flat = t.view(-1)
        n = flat.numel() // W
        rank = dist.get_rank(self.group)
        self.launches += 2
        if backend == "nccl":
            shard = torch.empty((n,), dtype=flat.dtype, device=flat.device)
            op = dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM
            dist.reduce_scatter_tensor(shard, flat, op=op, group=self.group, async_op=True)
            return dist.all_gather_into_tensor(flat, shard, group=self.group, async_op=True), None
        works = [dist.reduce(flat[r * n:(r + 1) * n], dst=dist.get_global_rank(self.group, r) if self.group is not None else r, op=dist.ReduceOp.SUM,
                             group=self.group, async_op=True) for r in range(W)]
        for w in works:
            w.wait()
        shard = flat[rank * n:(rank + 1) * n].clone()
        return dist.all_gather_into_tensor(flat, shard, group=self.group, async_op=True), ((1.0 / W) if self.average else None)

    def _flush(self):
        """coarse mode: exchange everything collected so far as one coalesced collective."""
        if not self._held:
            return
        ts, self._held, self._held_bytes = self._held, [], 0
        self.launches += 1
        backend = dist.get_backend(self.group)
        avg = self.average and backend == "nccl"
        post = (1.0 / self.world_size) if (self.average and not avg) else None
        if backend == "gloo" and ts[0].is_cuda:
            # gloo's coalesced all-reduce takes host tensors only (the two-ranks-on-one-GPU debug arrangement): per-tensor collectives,
            # still issued together at the bucket boundary
            for t in ts:
                self._pending.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), t, post))
            return
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", FutureWarning)          # (announced deprecation of the coalesced entry point; it is what ProcessGroupNCCL batches)
            try:
                fut = dist.all_reduce_coalesced(ts, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
            except (RuntimeError, ValueError):
                if not avg:
                    raise
                # a backend build whose coalesced path refuses AVG (argument check at call time, nothing has been enqueued): SUM + scale
                post = 1.0 / self.world_size
                fut = dist.all_reduce_coalesced(ts, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((fut, ts, post))

    def ready(self, G, names, flat=None):
        """Gradients `names` of dict G are enqueued on the current stream: start their all-reduce.  Weight matrices (>= 1 M
        elements: 4-268 MB messages, large enough to run at link bandwidth) are reduced in place; the bucket's small tensors
        (biases, LayerNorm, rel-pos tables) go as ONE message: `flat` when the producer wrote them into one contiguous buffer
        (the engine does, per transformer block: they are views of it, nothing is copied), otherwise they are flattened into a
        fresh buffer here and replaced by views of it."""
        if self.world_size == 1 and not _SELFTEST:
            return
        small = []
        for n in names:
            t = G[n]
            if t.numel() >= self.BIG and t.is_contiguous():
                work, post = self._reduce(t)
                self._pending.append((work, t, post))
            else:
                small.append(n)
        if small and flat is not None:
            # the tensors that are views of `flat` travel inside it; anything else that is small here (the weight matrices of a
            # down-sized test model, say) takes the flattening path below
            lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
            inside = [n for n in small if lo <= G[n].data_ptr() and G[n].data_ptr() + G[n].numel() * G[n].element_size() <= hi]
            assert inside, "flat holds none of the bucket's gradients"
            work, post = self._reduce(flat)
            self._pending.append((work, flat, post))
            small = [n for n in small if n not in inside]
        if small:
            ts = [G[n] for n in small]
            flat = torch.cat([t.reshape(-1) for t in ts])
            work, post = self._reduce(flat)
            off = 0
            for n, t in zip(small, ts):
                G[n] = flat[off:off + t.numel()].view(t.shape)
                off += t.numel()
            self._pending.append((work, flat, post))
        if self.mode == "coarse":
            self._pending = [e for e in self._pending if e[0] is not None]      # (collected tensors carry no work handle yet)
            if self._held_bytes >= self.bucket_bytes:
                self._flush()

    def finish(self):
        if self.mode == "coarse":
            self._flush()
        for work, flat, post in self._pending:
            work.wait()
            if post is not None:
                for t in (flat if isinstance(flat, (list, tuple)) else (flat,)):
                    t.mul_(post)
        self._pending = []


def broadcast_parameters(module, src=0, process_group=None):
    """Every rank adopts rank `src`'s parameters and buffers -- what the DDP wrapper does at construction
    (Painter/main_train.py:340); needed once when GradSync replaces the wrapper."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            # through a staging tensor and Tensor.copy_: a write through `.data` (or by the collective itself) does not bump the
            # parameter's version counter, and the engine's cached bf16 weight copies are keyed on it (engine.HotPath.w) -- a
            # replica that had already run a forward would keep multiplying by its OLD weights.
            buf = t.detach().clone()
            dist.broadcast(buf, src=src, group=process_group)
            t.copy_(buf)
    hot = getattr(module, "_hot", None)
    if hot is not None:
        for cache in (hot._wcache, hot._pcache, hot._rcache):
            cache.clear()                          # belt and braces: the T-typed operand copies are rebuilt on the next forward


def _ipc_env():
    """The host driver of the MI355X boxes only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's P2P set-up fails
    with `hipIpcGetMemHandle: invalid argument`.  HSA reads the variable when the runtime initialises, so it has to be in the
    environment before the first HIP call of the process (painter_amd/__init__.py sets it at import as well)."""
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY was unset and the HIP runtime is already initialised: export "
                          "HSA_ENABLE_IPC_MODE_LEGACY=0 before starting multi-GPU runs (INTEGRATION.md section 7)")


def init_distributed(backend=None):
    """env:// rendezvous as torchrun sets it up (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT);
    mirrors util/misc.py:217-249 without the SLURM/OMPI parsing.  -> (rank, local_rank, world_size)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not _SELFTEST:
        return 0, 0, 1
    _ipc_env()
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    return rank, local, world
