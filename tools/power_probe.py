"""Is the training step clock / power limited?  Samples the GPU's power draw and shader clock (amdgpu hwmon / pp_dpm files, falling back
to `rocm-smi`) every 100 ms while a workload loops for a few seconds each:
    step      the ViT-L B = 8 training step (bench.py's loop)                       [with PA_G256_ILV = 0 and 2 when the experiment library is loaded]
    gemm      the fc1 + GELU forward GEMM alone, back to back
    attn      the attention forward + backward alone, back to back
    ln        the LayerNorm backward alone (HBM-bound)
    idle      nothing
Prints mean / max power, mean shader clock and the work rate per workload.  A step that sits at the board's power cap with the clock
far below 2.4 GHz is energy-bound: a kernel that saves cycles without saving energy gives the cycles back as a lower clock."""
import glob
import os
import statistics
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from painter_amd import models_painter, ops  # noqa: E402
from painter_amd._lib import lib  # noqa: E402


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


class Sampler:
    """Samples every amdgpu hwmon node (a box may expose the nodes of GPUs that are not ours): the card whose power moves with the load
    is the one the process runs on; it is picked after the first loaded measurement."""

    def __init__(self):
        self.nodes = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.pfile = {n: next((os.path.join(n, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(n, f))), None) for n in self.nodes}
        self.ffile = {n: os.path.join(n, "freq1_input") for n in self.nodes}
        self.cap = {n: _read(os.path.join(n, "power1_cap")) for n in self.nodes}
        self.samples, self.alive, self.on, self.card = [], True, False, None

    def once(self):
        out = {}
        for n in self.nodes:
            pw, fq = _read(self.pfile[n]) if self.pfile[n] else None, _read(self.ffile[n])
            out[n] = (float(pw) / 1e6 if pw else None, float(fq) / 1e6 if fq else None)
        return out

    def loop(self):
        while self.alive:
            if self.on:
                self.samples.append(self.once())
            time.sleep(0.1)

    def start(self):
        threading.Thread(target=self.loop, daemon=True).start()

    def measure(self, fn, seconds, unit_work):
        fn()
        torch.cuda.synchronize()
        self.samples, self.on = [], True
        t0, n = time.time(), 0
        while time.time() - t0 < seconds:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            n += 4
        dt = time.time() - t0
        self.on = False
        per = {}
        for node in self.nodes:
            pw = [s_[node][0] for s_ in self.samples if s_[node][0] is not None]
            ck = [s_[node][1] for s_ in self.samples if s_[node][1] is not None]
            per[node] = (statistics.mean(pw) if pw else 0.0, max(pw) if pw else 0.0, statistics.mean(ck) if ck else 0.0, min(ck) if ck else 0.0)
        return {"rate": n * unit_work / dt, "ms": dt / n * 1e3, "per": per, "n": len(self.samples)}


def main():
    dev = torch.device("cuda")
    s = Sampler()
    print("hwmon nodes:", len(s.nodes), "power caps (uW):", sorted(set(s.cap.values())), flush=True)
    s.start()
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).train()
    c = m._cfg
    inp = bench.synthetic_inputs(8, c.H, c.W, c.L, 1234, dev)

    def step():
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()

    g = torch.Generator().manual_seed(0)
    T = torch.bfloat16
    rnd = lambda *sh: (torch.rand(sh, generator=g) * 2 - 1).to(T).to(dev)
    x, w, b = rnd(12544, 1024), rnd(4096, 1024) * 0.05, torch.zeros(4096, device=dev)
    L, H, Hp, Wp = 1568, 16, 56, 28
    qkv, dout = torch.randn(8 * L, 3 * H * 64, generator=g).to(T).to(dev), torch.randn(8 * L, H * 64, generator=g).to(T).to(dev)
    rel_h, rel_w = (torch.randn(111, 64, generator=g) * 0.05).to(dev), (torch.randn(55, 64, generator=g) * 0.05).to(dev)
    rcat, rcatT = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T), ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)

    def attn():
        out, lse, tab = ops.attn_fwd(qkv, rcat, 8, L, H, Hp, Wp, 0.125, need_tables=True)
        ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, 8, L, H, Hp, Wp, 0.125, tables=tab)

    xr = torch.randn(12544, 1024, generator=g).to(dev)
    gam = torch.ones(1024, device=dev)
    ln, mean, rstd = ops.layernorm_fwd(xr, gam, torch.zeros(1024, device=dev), 1e-6, T)
    dyl, dres = rnd(12544, 1024), torch.randn(12544, 1024, generator=g).to(dev)
    dxT = torch.empty(12544, 1024, dtype=T, device=dev)

    work = [("idle", lambda: time.sleep(0.05), 0.0)]
    ilvs = [0, 2] if "ilv" in os.environ.get("PAINTER_AMD_LIB", "") else [None]
    for i in ilvs:
        def f(i=i):
            if i is not None:
                lib.pa_debug_set(5, 1 + i)
            step()
        work.append(("step" + ("" if i is None else " ILV%d" % i), f, 8 * 4.769e12))
    for i in ilvs:
        def f(i=i):
            if i is not None:
                lib.pa_debug_set(5, 1 + i)
            ops.linear_gelu(x, w, b)
        work.append(("gemm fc1+gelu" + ("" if i is None else " ILV%d" % i), f, 2.0 * 12544 * 4096 * 1024))
    work.append(("attention fwd+bwd", attn, 3.5 * 4.0 * 8 * H * L * L * 64))
    work.append(("layernorm bwd", lambda: ops.layernorm_bwd(dyl, xr, mean, rstd, gam, dres=dres, dx=dres, dxT=dxT), 203e6))
    base = None
    for name, fn, unit in work:
        r = s.measure(fn, 4.0 if name != "idle" else 1.5, unit)
        if name == "idle":
            base = r["per"]
            print("idle: per-node mean power", ["%.0f" % v[0] for v in r["per"].values()], flush=True)
            continue
        if s.card is None:          # our GPU = the node whose power rose most over idle
            s.card = max(s.nodes, key=lambda n_: r["per"][n_][0] - base[n_][0])
            print("our GPU is", s.card, "(+%.0f W over idle; cap %s uW)" % (r["per"][s.card][0] - base[s.card][0], s.cap[s.card]), flush=True)
        pm, px, cm, cmin = r["per"][s.card]
        print("%-22s %8.3f ms/iter  %8.1f T(FLOP|B)/s   power mean %4.0f W  max %4.0f W   sclk mean %4.0f MHz  min %4.0f MHz   (%d samples)"
              % (name, r["ms"], r["rate"] / 1e12, pm, px, cm, cmin, r["n"]), flush=True)
    lib.pa_debug_set(5, 0)
    s.alive = False


if __name__ == "__main__":
    main()
