#!/usr/bin/env python3
"""Benchmark of the PSSR2 hot path on MI355X: HR tiles/s of ResUNet training, through ``pssr2_amd.train.train_paired``.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): ResUNet 1-ch 4xSR, 128^2 -> 512^2, bf16 storage / f32 accumulate,
batch 32 per GPU, SSIMLoss(mix=.8) (MS-SSIM + L1), AdamW, 4096 synthetic-EM uint8 HR tiles per GPU resident in HBM
(``DeviceTileDataset``).  A step = one iteration of ``train_paired``'s loop: device-side pair generation (crop / rot90 / flip
gather, Pillow-exact 4x reduction, AdditiveGaussian(13), round/clip), forward, loss, backward, (gradient all-reduce),
optimizer update -- replayed as one hipGraph by the driver itself (pssr2_amd/fastpath.py).  The timed region is bracketed
from inside the loop by a callback: barrier + synchronize after step W and after step W+K.

After the headline region rank 0 (N = 1) also times, each with its own warm-up: c2 inference through ``predict_images``
(batch 128), the c5 sheet through ``predict_sheet``, the exact-f32 training step, and the CPU oracle.  ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GFLOP_PER_TILE = 189.91      # 31.652 GMAC fwd x 2 FLOP x 3 (fwd + dgrad + wgrad), c2
FWD_GFLOP_PER_TILE = 63.30
PEAK_BF16_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_F32_TFLOPS = 157.3            # MI355X_MICROARCH.md: f32-input MFMA = vector rate
PEAK_HBM_GBS = 8000.0
# layer-wise roofline time per tile, sum over layers of max(FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) -> the bound on
# tiles/s per GPU; `layerwise_bound.frac` = measured / bound is the survey's single headline fraction
LAYERWISE_BOUND = {"c2_train": 10.6e3, "c2_infer": 31.8e3, "c3_train": 6.0e3, "c3_infer": 18.1e3, "c4_train": 2.6e3, "c4_infer": 7.9e3}
# the 3x3 loops for Cout > 64: 16x16-pixel x 128-channel tiles (conv_v3), 8x16 x 128 and -- where those would leave one workgroup per CU -- 8x16 x 64
DOMINANT_SYMBOLS = ("conv_igemm_kernelIDF16bLi128ELi0ELi9E", "conv_v3_kernelIDF16bLi128E", "conv_igemm_kernelIDF16bLi64ELi0ELi9E")


from pssr2_amd.distributed import CooperativeStop  # noqa: E402  (imports torch.distributed only; no HIP call)


class _Done(CooperativeStop):
    """Raised by the timing callback after the last timed step: callback exceptions abort the driver loop (as upstream).  Every rank
    raises it at the same step, so the multi-rank failure watch must not treat it as a rank failure (distributed.CooperativeStop)."""


def layerwise(key, tiles_per_s_per_gpu):
    b = LAYERWISE_BOUND[key]
    return {"tiles_per_s": b, "frac": round(tiles_per_s_per_gpu / b, 4),
            "is": "SURVEY.md 8(d): sum over layers of max(FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) per tile; frac = measured / bound"}


def traffic_table(mode):
    """profiles/traffic.json[mode] (tools/summarize_profile.py from the committed rocprofv3 --pmc passes), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)[mode]
    except (OSError, KeyError, ValueError):
        return None


def kernel_traffic(mode, symbol):
    """Counter HBM bytes per launch of the kernel whose (mangled) name contains ``symbol``, launch-weighted over its entries."""
    t = traffic_table(mode)
    if not t:
        return None
    tot_b = tot_n = 0
    for name, rec in t["kernels"].items():
        if symbol in name:
            tot_b += rec["hbm_bytes_per_launch"] * rec.get("launches", 1)
            tot_n += rec.get("launches", 1)
    return round(tot_b / tot_n) if tot_n else None


def pass_traffic(mode):
    """Counter HBM bytes of one whole pass (every kernel of a training step / an inference batch) from the same table, or None.  The
    number of passes the counters saw = the launches of a once-per-pass kernel (AdamW for training, the input im2col for inference)."""
    t = traffic_table(mode)
    if not t:
        return None
    once = "adamw_kernel" if mode.endswith("train") else "input_im2col_kernel"
    passes = sum(r.get("launches", 0) for name, r in t["kernels"].items() if once in name)
    if not passes:
        return None
    return round(sum(r["hbm_bytes_per_launch"] * r.get("launches", 1) for r in t["kernels"].values()) / passes)


def whole_pass_roofline(tflops, scope, traffic=None):
    """roofline object of an extra leg: the whole pass against the dense MFMA peak (the per-kernel breakdown of the inference pass is
    profiles/r04_infer_*; its dominant kernels are the training forward's).  ``traffic``: counter HBM bytes of one pass (one step / one
    batch) summed over all its kernels, from profiles/traffic.json."""
    return {"bound": "mfma", "achieved": tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": "HBM bytes per pass (one step / one batch), all kernels", "scope": scope}


def pmc_traffic(mode):
    """HBM bytes per launch of the dominant kernel set from the committed rocprofv3 --pmc passes (profiles/traffic.json, written
    by tools/summarize_profile.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            data = json.load(f)[mode]
        tot_b, tot_n = 0.0, 0
        for name, rec in data["kernels"].items():
            if any(s in name for s in DOMINANT_SYMBOLS):
                tot_b += rec["hbm_bytes_per_launch"] * rec.get("launches", 1)
                tot_n += rec.get("launches", 1)
        if tot_n:
            return tot_b / tot_n, data["source"]
    except (OSError, KeyError, ValueError):
        pass
    return None, None


class ConvTimer:
    """HIP-event timing of the dominant kernel set (3x3 conv forward + dgrad with Cout > 64, 16-bit storage) on the launch stream,
    with the algorithmic FLOPs and bytes of every launch (input read once, output written once, + the weights)."""

    def __init__(self):
        self.events = []
        self.orig = None

    def install(self):
        from pssr2_amd import ops, _lib as L_
        self.orig = ops.conv2d
        timer = self

        def timed(x, cin0, w0, out, cout, **kw):
            w = kw["w"]
            dominant = cout > 64 and w > 8 and w0.dtype in (1, 2) and w0.taps == 9       # 16-bit storage (bf16 = 1, f16 = 2)
            if not dominant:
                return timer.orig(x, cin0, w0, out, cout, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = timer.orig(x, cin0, w0, out, cout, **kw)
            e1.record()
            npix = kw["n"] * kw["h"] * w
            k0, k1 = min(cin0, w0.k_pad), kw.get("cin1", 0)
            flops = 2.0 * npix * cout * (w0.taps * k0 + k1)
            nbytes = 2.0 * (npix * (k0 + k1 + cout) + cout * (w0.taps * k0 + k1))
            if kw.get("aux") is not None:
                nbytes += 2.0 * npix * cout
            # Reconstruction.pre's forward with Reconstruction.conv's forward in its epilogue (FLAG_HEADQ / EPI_HEADQ): the launch also does
            # the head's 2 x 9 x 64 FLOP per high-resolution pixel and writes the nine f32 tap planes (36 B per high-resolution pixel);
            # in eval mode (EPI_HEADQ) it does not write its own activation
            fused = bool(kw.get("flags", 0) & L_.FLAG_HEADQ) or kw.get("epilogue", 0) == L_.EPI_HEADQ
            if fused:
                flops += 2.0 * 9 * npix * cout
                nbytes += 36.0 * npix * (cout // 64)
                if kw.get("epilogue", 0) == L_.EPI_HEADQ:
                    nbytes -= 2.0 * npix * cout
            timer.events.append((e0, e1, flops, nbytes, fused))
            return r
        ops.conv2d = timed
        import pssr2_amd.engine as E
        E.ops.conv2d = timed

    def remove(self):
        from pssr2_amd import ops
        import pssr2_amd.engine as E
        ops.conv2d = self.orig
        E.ops.conv2d = self.orig

    def summary(self):
        if not self.events:
            return None
        ms = [e[0].elapsed_time(e[1]) for e in self.events]
        tot_ms, tot_fl, tot_b = sum(ms), sum(e[2] for e in self.events), sum(e[3] for e in self.events)
        n = len(ms)
        out = dict(launches=n, avg_us=1e3 * tot_ms / n, tflops=tot_fl / (tot_ms * 1e-3) / 1e12, gflop_per_launch=tot_fl / n / 1e9,
                   bytes_per_launch=tot_b / n, fused_head=None)
        fz = [(m, e) for m, e in zip(ms, self.events) if e[4]]
        if fz and len(fz) < n:
            f_ms, f_fl = sum(m for m, _ in fz), sum(e[2] for _, e in fz)
            out["fused_head"] = dict(launches=len(fz), avg_us=1e3 * f_ms / len(fz), tflops=f_fl / (f_ms * 1e-3) / 1e12,
                                     others_tflops=(tot_fl - f_fl) / ((tot_ms - f_ms) * 1e-3) / 1e12, others_avg_us=1e3 * (tot_ms - f_ms) / (n - len(fz)))
        return out


class HbmTimer:
    """HIP-event timing of the HBM-bound kernels north_star asks a GB/s figure for: the reconstruction head (``Reconstruction.conv`` on the
    pixel-shuffled 64-channel HR tensor: forward, and its one-pass backward) and the fused AdamW update.  Algorithmic bytes: the head's
    activation read once (+ its gradient written once in the backward) plus the f32 HR images; AdamW 28 B per parameter (p, g, m, v
    read; p, m, v written)."""

    def __init__(self):
        self.rec = {}
        self.orig = {}

    def _wrap(self, name, fn, nbytes):
        timer = self

        def timed(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            timer.rec.setdefault(name, []).append((e0, e1, float(nbytes(*a, **kw))))
            return r
        return timed

    def install(self):
        from pssr2_amd import ops
        esz = lambda code: 4 if code == 0 else 2
        # head_conv_fwd(act, blk, weight, bias, out, n, h, w, cin, cout, out_scale, out_shift, dtype, ...): h, w are HR sizes
        fwd_b = lambda act, blk, wt, b, out, n, h, w, cin, cout, osc, osh, dt, **kw: n * h * w * (cin * esz(dt) + 4 * cout)
        # head_conv_bwd_rows(g, g_scale, weight, act, dact, blk, dw_rows, bias_rows, n, h, w, cin, cout, dtype)
        bwd_b = lambda g, gs, wt, act, dact, blk, dwr, br, n, h, w, cin, cout, dt: n * h * w * (2 * cin * esz(dt) + 4 * cout)
        adam_b = lambda p, *a, **kw: 28 * p.numel()
        # head_q_gather(q, bias, out, n, hh, ww, scale, shift): nine f32 planes read, the f32 image written
        gat_b = lambda q, b, out, n, h, w, *a, **kw: 40 * n * h * w * 16
        for name, nb in (("head_conv_fwd", fwd_b), ("head_q_gather", gat_b), ("head_conv_bwd_rows", bwd_b), ("adamw_step_dev", adam_b), ("adamw_step", adam_b)):
            self.orig[name] = getattr(ops, name)
            setattr(ops, name, self._wrap(name, self.orig[name], nb))

    def remove(self):
        from pssr2_amd import ops
        for name, fn in self.orig.items():
            setattr(ops, name, fn)

    def objects(self, mode="train"):
        out = []
        sym = {"head_conv_fwd": "head_fwd_kernel", "head_q_gather": "head_q_gather_kernel", "head_conv_bwd_rows": "head_bwd_kernel",
               "adamw_step_dev": "adamw_kernel", "adamw_step": "adamw_kernel"}
        label = {"head_conv_fwd": "head_fwd_kernel (Reconstruction.conv forward on the pixel-shuffled HR tensor)",
                 "head_q_gather": "head_q_gather_kernel (sum of the nine tap planes Reconstruction.pre's epilogue wrote: what is left of Reconstruction.conv's forward)",
                 "head_conv_bwd_rows": "head_bwd_kernel (its data + weight gradient + bias sums in one pass)",
                 "adamw_step_dev": "adamw_kernel (fused AdamW over the flat parameter buffer)", "adamw_step": "adamw_kernel (fused AdamW over the flat parameter buffer)"}
        for name, ev in self.rec.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in ev)
            nb = sum(e[2] for e in ev)
            if ms <= 0:
                continue
            gbs = nb / (ms * 1e-3) / 1e9
            out.append({"kernel": label[name], "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "launches": len(ev), "avg_launch_us": round(1e3 * ms / len(ev), 1),
                        "algorithmic_bytes_per_launch": round(nb / len(ev)), "traffic": kernel_traffic(mode, sym[name]),
                        "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc passes, profiles/traffic.json)"})
        return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """The oracle (CPU restatement of the reference path, torch fp32, oneDNN) on the host cores: one warm-up + 3 timed training
    steps (fwd + MS-SSIM/L1 + bwd + AdamW) at the c1 shape (batch 4, 64^2 -> 256^2) and the c2 shape (batch 4, 128^2 -> 512^2), and
    3 timed eval forwards at the c2 shape.  Bounded: a few tens of seconds."""
    from oracle import loss_ref, model_ref
    torch.manual_seed(0)
    # 16 threads: measured fastest for this step on the MI355X host (2 x EPYC 9575F, 256 hardware threads) -- 8 / 16 / 32 / 64 / 128
    # threads gave 2.6 / 3.8 / 3.3 / 1.5 / 0.7 tiles/s at the c2 shape, and larger batches (8 - 32) were slower per tile at every count
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    sd = model_ref.make_state_dict(seed=1, randomize_bn=False)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=1e-3)

    def train_steps(batch, lr_res, n):
        lr = torch.rand(batch, 1, lr_res, lr_res) * 255
        hr = torch.rand(batch, 1, lr_res * 4, lr_res * 4) * 255
        ts = []
        for i in range(n + 1):
            t0 = time.perf_counter()
            y, _ = model_ref.resunet_forward(lr, params, 5, 3, 4, train=True)
            loss = loss_ref.ssim_loss(y / 255, hr / 255, mix=0.8)
            loss.backward()
            opt.step()
            opt.zero_grad()
            if i:
                ts.append(time.perf_counter() - t0)
        return batch / (sum(ts) / len(ts)), sum(ts)

    def infer_steps(batch, lr_res, n):
        lr = torch.rand(batch, 1, lr_res, lr_res) * 255
        ts = []
        with torch.no_grad():
            for i in range(n + 1):
                t0 = time.perf_counter()
                model_ref.resunet_forward(lr, params, 5, 3, 4, train=False)
                if i:
                    ts.append(time.perf_counter() - t0)
        return batch / (sum(ts) / len(ts)), sum(ts)

    def pairgen(n):
        """Pair generation as the reference's _gen_pair does it for a 512^2 uint8 tile (pssr/data.py:471-495): Pillow BILINEAR 512^2 -> 128^2
        (the reference's own third-party call; the oracle's numpy restatement of it when Pillow is absent), AdditiveGaussian(13) from numpy's
        generator, np.round + clip -- one host thread, as one DataLoader worker runs it.  BASELINE.md section 4's pair-generation row."""
        import numpy as np
        from oracle import pairs_ref
        rng = np.random.default_rng(0)
        tiles = rng.integers(0, 256, size=(8, 512, 512), dtype=np.uint8)
        try:
            from PIL import Image
            reduce_ = lambda a: np.asarray(Image.fromarray(a).resize((128, 128), Image.Resampling.BILINEAR))
            how = "Pillow BILINEAR"
        except ImportError:
            reduce_ = lambda a: pairs_ref.pil_bilinear_u8(a[None], 128, 128)[0]
            how = "oracle restatement of Pillow BILINEAR (numpy)"
        t0 = time.perf_counter()
        for i in range(n):
            lr = reduce_(tiles[i % 8]).astype(np.float32)
            lr = pairs_ref.round_clip(pairs_ref.additive_gaussian(lr, rng.normal(0, 13, lr.shape))).astype(np.float32)
        return n / (time.perf_counter() - t0), how

    c1, t1 = train_steps(4, 64, 3)
    c2, t2 = train_steps(4, 128, 3)
    inf, t3 = infer_steps(4, 128, 3)
    pg, pg_how = pairgen(1500)
    return dict(pairgen_tiles_per_s=round(pg, 1), pairgen_cores=1,
                pairgen_sample=f"1500 x (512^2 uint8 -> {pg_how} 128^2 -> + N(0, 13) -> np.round, clip) on ONE host thread "
                               "(the reference runs it per item inside a DataLoader worker; BASELINE.md section 4 row 'pair generation per tile')",
                **dict(value=c2, unit="HR tiles/s", cores=threads, kind="port", cpu=cpu_model(),
                sample=f"CPU oracle, torch fp32, {threads} threads (fastest of 8-128 on this host; {os.cpu_count()} hardware threads present), 1 warm-up + 3 timed steps each: train step (fwd + MS-SSIM/L1 + bwd + AdamW) "
                       f"batch 4 at 128^2->512^2 = value ({t2:.1f} s); c1 shape 64^2->256^2 batch 4: {c1:.2f} tiles/s of 256^2 ({t1:.1f} s); "
                       f"eval forward batch 4 at 128^2->512^2: {inf:.2f} tiles/s ({t3:.1f} s)",
                c1_train_tiles_per_s=round(c1, 3), c2_infer_tiles_per_s=round(inf, 3)))


def make_tiles(n, res, channels, rank, max_workers=16):
    """n seeded synthetic-EM uint8 tiles [n, channels, res, res], generated by a process pool before the GPU is touched."""
    import numpy as np
    from concurrent.futures import ProcessPoolExecutor
    from functools import partial
    from pssr2_amd.data import synthetic_em_tile
    workers = max(1, min(max_workers, (os.cpu_count() or 8) // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    idx = [rank * 1000003 + i for i in range(n)]
    if workers == 1 or n < 64:
        tiles = [synthetic_em_tile(i, res, channels) for i in idx]
    else:
        with ProcessPoolExecutor(workers) as ex:
            tiles = list(ex.map(partial(synthetic_em_tile, res=res, channels=channels), idx, chunksize=16))
    return np.stack(tiles)


class StepClock:
    """Callback of the driver loop: untimed warm-up steps, then exactly ``steps`` steps between two barrier + synchronize points."""

    def __init__(self, warmup, steps, barrier):
        self.warmup, self.steps, self.barrier = warmup, steps, barrier
        self.count, self.t0, self.elapsed = 0, None, None

    def __call__(self):
        self.count += 1
        if self.count == self.warmup:
            self.barrier()
            self.t0 = time.perf_counter()
        elif self.count == self.warmup + self.steps:
            self.barrier()
            self.elapsed = time.perf_counter() - self.t0
            raise _Done()


def spawn_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher around it: start the N ranks as a CHILD ``torch.distributed.run`` (one process per
    GPU, RCCL) before this process has made any HIP call -- counting devices does not initialise the GPU, and a process that has touched
    the GPU must never re-launch itself -- pass rank 0's JSON line through, and return the launcher's exit code (non-zero when any rank
    failed).  Fewer than N devices is an error, not a 1-GPU run: the line would otherwise claim ``n_gpus`` it never used."""
    import socket
    import subprocess
    rehearsal = os.environ.get("PSSR_BENCH_FORCE_DEVICE") is not None        # several gloo ranks on ONE card (tests / tools only)
    have = torch.cuda.device_count()
    if have < n and not rehearsal:
        print(f"bench: --gpus {n} needs {n} devices, this node shows {have}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step")
    ap.add_argument("--lr-res", type=int, default=128)
    ap.add_argument("--tiles", type=int, default=4096, help="resident HR tiles per GPU (SURVEY.md §8d)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "f32"])
    ap.add_argument("--channels", type=int, default=1, help="image channels / frames (BASELINE config 4 uses 3)")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "sheet"],
                    help="infer: predict_images at --batch; sheet: BASELINE config 5 (4096^2 LR sheet, 128^2 tiles, overlap 32, batch 128)")
    ap.add_argument("--model", default="resunet", choices=["resunet", "rdresunet"])
    ap.add_argument("--crappifier", default="gaussian", choices=["gaussian", "poisson"])
    ap.add_argument("--tile-workers", type=int, default=16, help="processes generating the synthetic tiles (1 under rocprofv3: no forks behind the profiler)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the infer / sheet / f32 legs that follow the headline region")
    ap.add_argument("--no-graph", action="store_true", help="PSSR_GRAPH=0: every kernel launched from Python by the drivers")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} ranks")
    if args.no_graph:
        os.environ["PSSR_GRAPH"] = "0"
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ["PSSR_COMM_STATS"] = "1"
    if args.warmup < 3 and not args.no_graph:
        args.warmup = 3                     # two eager steps + the capture come first (pssr2_amd/fastpath.py)

    rank_env, world_env = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    hr_res = args.lr_res * 4
    n_tiles = args.tiles if args.mode != "sheet" else 0
    need = (args.warmup + args.steps) * args.batch
    tiles_np = make_tiles(n_tiles, hr_res, args.channels, rank_env, args.tile_workers) if n_tiles else None      # before CUDA: the pool forks
    extras = (rank_env == 0 and world_env == 1 and args.mode == "train" and not args.no_extras and args.model == "resunet" and args.channels == 1
              and args.lr_res == 128)
    # BASELINE config 4: 3-frame 1024^2 HR tiles.  96 of them: 86 training tiles = 10 full batches of 8, so that the 3 + 5 steps of the leg
    # stay clear of the epoch's partial last batch (one eager step on a new engine plan: 30 ms once per run, not a step of the replayed graph)
    tiles_c4 = make_tiles(96, 1024, 3, 77, args.tile_workers) if extras else None

    from pssr2_amd import distributed as D
    force_dev = os.environ.get("PSSR_BENCH_FORCE_DEVICE")       # rehearsal knob: several gloo ranks on ONE card
    if force_dev is not None:
        torch.cuda.set_device(int(force_dev))
    rank, world, local = D.init_from_env(os.environ.get("PSSR_BENCH_BACKEND"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if force_dev is not None:
        local = int(force_dev)
    torch.cuda.set_device(local if world > 1 else 0)
    dev = torch.device("cuda", local if world > 1 else 0)

    from pssr2_amd.crappifiers import AdditiveGaussian, Poisson
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.models import RDResUNet, ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.predict import predict_images, predict_sheet
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    import numpy as np

    dtypes = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}

    def make_model(dtype):
        torch.manual_seed(0)
        m = (ResUNet(channels=args.channels) if args.model == "resunet" else RDResUNet(channels=args.channels)).to(dev)
        m.compute_dtype = dtypes[dtype]
        return m

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    crap = AdditiveGaussian(13, 0, 0) if args.crappifier == "gaussian" else Poisson()
    loss_fn = SSIMLoss(channels=args.channels, mix=0.8)
    quiet = open(os.devnull, "w")

    def run_train(model, dataset, batch, warmup, steps, dataloader_kwargs=None, loss=None):
        """train_paired until the clock has seen warmup + steps steps; returns seconds for the timed steps (max over ranks)."""
        loss = loss_fn if loss is None else loss
        opt = FusedAdamW(model.parameters(), lr=1e-3)
        clock = StepClock(warmup, steps, barrier)
        per_epoch = (len(dataset) - len(dataset.val_idx)) // world // batch
        epochs = max(1, -(-(warmup + steps) // max(per_epoch, 1)))
        old = sys.stdout
        sys.stdout = quiet                               # the driver prints per-epoch lines (as upstream); the bench prints ONE line
        try:
            train_paired(model, dataset, batch, loss, opt, epochs, device=dev, callbacks=[clock], dataloader_kwargs=dataloader_kwargs)
        except _Done:
            pass
        finally:
            sys.stdout = old
        if clock.elapsed is None:
            raise SystemExit(f"bench: the dataset holds fewer than {warmup + steps} batches of {batch}")
        el = clock.elapsed
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = t.item()
        return el, opt

    def run_infer(model, dataset, batch, reps=2):
        """predict_images twice (the first call captures the graph); returns seconds of the second call."""
        model.eval()
        for r in range(reps):
            barrier()
            t0 = time.perf_counter()
            out = predict_images(model, dataset, device=dev, batch_size=batch, out_dir=None)
            barrier()
            el = time.perf_counter() - t0
        assert len(out) == len(dataset.val_idx)
        return el

    res = {}
    timer = ConvTimer()
    tiles_dev = torch.from_numpy(tiles_np).to(dev) if tiles_np is not None else None
    model = make_model(args.dtype)
    D.broadcast_module(model)
    per_step = args.batch

    if args.mode == "train":
        ds = DeviceTileDataset(tiles_dev, hr_res=hr_res, lr_scale=4, crappifier=crap, val_split=0.1, rotation=True, device=dev, seed=1234 + rank)
        elapsed, opt = run_train(model, ds, args.batch, args.warmup, args.steps)
        steps_done = args.steps
    elif args.mode == "infer":
        ds = DeviceTileDataset(tiles_dev, hr_res=hr_res, lr_scale=4, crappifier=crap, val_split=1.0, rotation=False, device=dev, seed=1234 + rank)
        ds.device_outputs = True
        elapsed = run_infer(model, ds, args.batch)
        steps_done = -(-len(ds.val_idx) // args.batch)
        args.steps, args.warmup = steps_done, steps_done
    else:
        model.eval()
        rng = np.random.default_rng(7 + rank)
        sheet = torch.from_numpy(rng.integers(0, 256, size=(args.channels, 4096, 4096), dtype=np.uint8)).to(dev)
        sheet_tiles = ((4096 - args.lr_res) // (args.lr_res - 32) + 1) ** 2
        args.batch = max(args.batch, 128)
        for _ in range(args.warmup):
            predict_sheet(model, sheet, tile_res=args.lr_res, overlap=32, margin=8, batch_size=args.batch, device=dev, to_numpy=False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            predict_sheet(model, sheet, tile_res=args.lr_res, overlap=32, margin=8, batch_size=args.batch, device=dev, to_numpy=False)
        barrier()
        elapsed = time.perf_counter() - t0
        per_step, steps_done = sheet_tiles, args.steps

    # ---- N > 1: the ranks' parameters must be identical after the timed steps (same averaged gradients, same update kernel): a checksum
    # of the flat parameter buffer (sum and sum of magnitudes, f64) is all-gathered and compared exactly, BEFORE anything rank-specific runs
    in_sync = None
    if world > 1 and args.mode == "train":
        pf = opt._flat[0]["flat"].double()
        chk = torch.stack([pf.sum(), pf.abs().sum()])
        every = [torch.zeros_like(chk) for _ in range(world)]
        torch.distributed.all_gather(every, chk)
        in_sync = all(torch.equal(every[0], e) for e in every[1:])
        del pf
    # ---- N > 1: what the gradient exchange costs and how much of it the backward pass hides
    res_comm = None
    if world > 1 and args.mode == "train":
        flat = model._engine._flat_grad
        probe = torch.zeros_like(flat)
        ts = []
        for i in range(4):
            barrier()
            t0 = time.perf_counter()
            D.sum_flat(probe)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        comm_ms = 1e3 * min(ts[1:])
        ranks = [None] * world
        torch.distributed.all_gather_object(ranks, (rank, local, os.uname().nodename))
        stp = getattr(model._engine, "last_train_stepper", None)
        exposed = stp.exposed_comm_ms(last=args.steps) if stp is not None else None
        if rank == 0:
            res_comm = {"weights_in_sync": in_sync, "weights_in_sync_is": "sum and sum of magnitudes (f64) of the flat parameter buffer after the timed "
                        "steps, all-gathered and compared exactly over the ranks",
                        "reduce": "reduce_scatter+all_gather" if D.use_reduce_scatter() else "all_reduce",
                        "allreduce_bytes_per_step": int(flat.numel() * 4), "comm_ms": round(comm_ms, 3),
                           "comm_ms_is": "one sum of the whole flat gradient buffer over the ranks (all-reduce, or reduce-scatter + all-gather with "
                                         "PSSR_DDP_RS=1), nothing else running (min of 3)",
                           "exposed_ms_per_step": None if exposed is None else round(exposed, 3),
                           "overlap_frac": None if exposed is None else round(max(0.0, 1.0 - exposed / comm_ms), 3),
                           "split_graph": bool(stp is not None and stp.graph2 is not None), "ranks_seen": sorted(ranks),
                           "bus_GBps": round(2 * (world - 1) / world * flat.numel() * 4 / (comm_ms * 1e-3) / 1e9, 1)}
    # ---- roofline of the dominant kernel set: the same kernels once more, one at a time on the launch stream, HIP events
    conv, hbm_objects = None, []
    if rank == 0 and args.dtype in ("bf16", "fp16") and args.mode != "sheet":
        eng = model._engine
        # (world > 1: only rank 0 runs these two extra optimizer steps -- its weights, optimizer state and BatchNorm buffers are put back
        # afterwards, so that the ranks leave the bench as they left the timed region: in sync)
        snap = None
        if world > 1 and args.mode == "train":
            st0 = opt._flat[0]
            snap = ([st0[k].clone() for k in ("flat", "m", "v")] + ([st0["dev"].clone()] if "dev" in st0 else []), st0["step"],
                    [b.clone() for b in model.buffers()])
        side, eng.side_wgrad = eng.side_wgrad, False
        eng.mark_weights_changed()
        timer.install()
        hbm = HbmTimer()
        hbm.install()
        x = torch.rand(args.batch, args.channels, args.lr_res, args.lr_res, device=dev) * 255
        tgt = torch.rand(args.batch, args.channels, hr_res, hr_res, device=dev)
        for _ in range(2):
            if args.mode == "train":
                model.train()
                loss_fn(model(x) / 255, tgt).backward()
                opt.step()
                opt.zero_grad()
            else:
                model.eval()
                with torch.no_grad():
                    model(x)
        torch.cuda.synchronize()
        timer.remove()
        hbm.remove()
        default_hbm = args.dtype == "bf16" and args.lr_res == 128 and args.channels == 1 and args.batch == (32 if args.mode == "train" else args.batch)
        hbm_objects = hbm.objects(args.mode if args.model == 'resunet' else f'{args.model}_{args.mode}') if default_hbm else hbm.objects('none')
        eng.side_wgrad = side
        if snap is not None:
            st0 = opt._flat[0]
            for k, t in zip(("flat", "m", "v"), snap[0]):
                st0[k].copy_(t)
            if "dev" in st0:
                st0["dev"].copy_(snap[0][3])
            st0["step"] = snap[1]
            for bdst, bsrc in zip(model.buffers(), snap[2]):
                bdst.copy_(bsrc)
        eng.mark_weights_changed()
        conv = timer.summary()

    if rank == 0:
        tiles_per_s = world * per_step * steps_done / elapsed
        gflop_tile = TRAIN_GFLOP_PER_TILE if args.mode == "train" else FWD_GFLOP_PER_TILE
        if args.model == "rdresunet":      # 53.575 GMAC fwd per tile
            gflop_tile = 321.45 if args.mode == "train" else 107.15
        if args.model == "resunet" and args.channels == 3:      # 129.109 GMAC fwd per 1024^2 tile = 64.555 GFLOP per 128^2 of LR
            gflop_tile = 774.65 / 4 if args.mode == "train" else 258.22 / 4
        scale = (args.lr_res / 128) ** 2
        peak = PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS
        name = "ResUNet" if args.model == "resunet" else "RDResUNet"
        res = {
            "metric": f"HR tiles/sec ({hr_res}^2 4xSR) {args.mode}",
            "value": round(tiles_per_s, 2), "unit": "HR tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / steps_done, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype if args.mode == "train" else {torch.bfloat16: "bf16", torch.float16: "fp16", torch.float32: "f32"}[model._engine.storage_dtype(False)],
            "data": "synthetic",
            "config": {"workload": f"{name} {args.channels}-ch 4xSR {args.lr_res}^2->{hr_res}^2 {args.mode}, batch {args.batch}/GPU, "
                                   f"{n_tiles if n_tiles else 'one 4096^2 sheet:'} {'synthetic-EM uint8 HR tiles resident in HBM' if n_tiles else ''}, "
                                   f"{'AdditiveGaussian(13)' if args.crappifier == 'gaussian' else 'Poisson()'} device crappifier, MS-SSIM+L1 (mix .8), FusedAdamW",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "driver": {"train": "pssr2_amd.train.train_paired", "infer": "pssr2_amd.predict.predict_images", "sheet": "pssr2_amd.predict.predict_sheet"}[args.mode],
                       "launch": "eager" if args.no_graph or args.mode == "sheet" else "hipGraph replay inside the driver"},
            "step_compute": {"algorithmic_tflops": round(tiles_per_s * gflop_tile * scale / 1e3 / world, 2),
                             "frac_of_peak": round(tiles_per_s * gflop_tile * scale / 1e3 / world / peak, 4),
                             "traffic": pass_traffic(args.mode if args.model == "resunet" else f"{args.model}_{args.mode}")
                             if (args.dtype == "bf16" and args.lr_res == 128 and args.channels == 1 and args.mode != "sheet") else None,
                             "traffic_unit": "HBM bytes per step / batch, all kernels (rocprofv3 --pmc passes, profiles/traffic.json)"},
        }
        lw_key = {("resunet", 1): "c2", ("rdresunet", 1): "c3", ("resunet", 3): "c4"}.get((args.model, args.channels))
        if lw_key is not None and args.lr_res == (256 if lw_key == "c4" else 128):
            res["layerwise_bound"] = layerwise(f"{lw_key}_{'train' if args.mode == 'train' else 'infer'}", tiles_per_s / world)
        default_workload = args.dtype == "bf16" and args.lr_res == 128 and args.channels == 1 and args.batch == (32 if args.mode == "train" else args.batch)
        traffic, traffic_src = pmc_traffic(args.mode if args.model == "resunet" else f"{args.model}_{args.mode}") if default_workload else (None, None)
        if conv:
            res["roofline"] = {"bound": "mfma", "kernel": "3x3 conv forward + dgrad, Cout > 64 (conv_v3_kernel 16x16-pixel x 128-channel tiles where they fill the chip, "
                                                          "conv_igemm_kernel 8x16-pixel tiles of 128 or 64 channels otherwise)",
                               "achieved": round(conv["tflops"], 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(conv["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                               "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(conv["bytes_per_launch"]),
                               "launches": conv["launches"] // 2, "instrumented_passes": 2, "avg_launch_us": round(conv["avg_us"], 1),
                               "algorithmic_gflop_per_launch": round(conv["gflop_per_launch"], 2),
                               "fused_head_launch": None if not conv.get("fused_head") else {
                                   "is": "Reconstruction.pre's forward, which since round 4 also evaluates Reconstruction.conv's forward in its epilogue "
                                         "(tap planes, PSSR_FLAG_HEADQ / PSSR_EPI_HEADQ): head_fwd_kernel's pass over the 1.07 GB tensor (290 us) is gone "
                                         "from the step, this launch is longer by the plane stores",
                                   "launches_per_pass": conv["fused_head"]["launches"] // 2, "avg_launch_us": round(conv["fused_head"]["avg_us"], 1),
                                   "achieved": round(conv["fused_head"]["tflops"], 2),
                                   "class_without_it": {"achieved": round(conv["fused_head"]["others_tflops"], 2),
                                                        "frac": round(conv["fused_head"]["others_tflops"] / PEAK_BF16_TFLOPS, 4),
                                                        "avg_launch_us": round(conv["fused_head"]["others_avg_us"], 1)}},
                               "measured_in": "instrumented eager passes after the timed region: the same kernels one at a time on the launch stream, HIP events "
                                              "(the timed region replays them from a hipGraph with the weight-gradient kernels overlapping on a second stream)"}
        if hbm_objects:
            res["roofline_hbm"] = hbm_objects

    if rank == 0 and res_comm is not None:
        res["comm"] = res_comm
    # ---- extra legs (rank 0 of a 1-GPU run): inference, sheet, exact-f32 training, CPU oracle
    if extras:
        try:
            infer_dt = {torch.bfloat16: "bf16", torch.float16: "fp16", torch.float32: "f32"}[model._engine.storage_dtype(False)]
            ids = DeviceTileDataset(tiles_dev, hr_res=hr_res, lr_scale=4, crappifier=crap, val_split=1.0, rotation=False, device=dev, seed=99)
            ids.device_outputs = True
            t = run_infer(model, ids, 128)
            res["infer"] = {"metric": f"HR tiles/sec ({hr_res}^2 4xSR) infer", "value": round(len(ids.val_idx) / t, 2), "unit": "HR tiles/s",
                            "dtype": infer_dt,
                            "config": f"predict_images over {len(ids.val_idx)} resident tiles, batch 128, uint8 predictions kept in HBM, {infer_dt} storage "
                                      f"(the default inference storage of a {args.dtype} model: Engine.storage_dtype)",
                            "algorithmic_tflops": round(len(ids.val_idx) / t * FWD_GFLOP_PER_TILE / 1e3, 2)}
            res["infer"]["roofline"] = whole_pass_roofline(res["infer"]["algorithmic_tflops"], "predict_images pass: every kernel of the forward, "
                                                           "uint8 clipping included; algorithmic FLOPs of the forward convolutions",
                                                           traffic=pass_traffic("infer"))
            res["infer"]["layerwise_bound"] = layerwise("c2_infer", res["infer"]["value"])
            del ids
            rng = np.random.default_rng(7)
            sheet = torch.from_numpy(rng.integers(0, 256, size=(1, 4096, 4096), dtype=np.uint8)).to(dev)
            sheet_tiles = ((4096 - 128) // 96 + 1) ** 2
            model.eval()
            ts = []
            for i in range(3):
                barrier()
                t0 = time.perf_counter()
                predict_sheet(model, sheet, tile_res=128, overlap=32, margin=8, batch_size=128, device=dev, to_numpy=False)
                barrier()
                ts.append(time.perf_counter() - t0)
            res["sheet"] = {"metric": "HR tiles/sec (512^2 4xSR) whole-sheet inference", "value": round(sheet_tiles / min(ts[1:]), 2), "unit": "HR tiles/s",
                            "config": f"predict_sheet: 4096^2 uint8 sheet -> {sheet_tiles} tiles of 128^2 (overlap 32), batch 128, device tiling + "
                                      f"overlap-averaged reassembly, {infer_dt} storage", "dtype": infer_dt, "seconds_per_sheet": round(min(ts[1:]), 4)}
            res["sheet"]["algorithmic_tflops"] = round(sheet_tiles / min(ts[1:]) * FWD_GFLOP_PER_TILE / 1e3, 2)
            res["sheet"]["layerwise_bound"] = layerwise("c2_infer", res["sheet"]["value"])
            res["sheet"]["roofline"] = whole_pass_roofline(res["sheet"]["algorithmic_tflops"], "predict_sheet: device tiling + forward + "
                                                           "overlap-averaged reassembly, host wall clock; algorithmic FLOPs of the forward convolutions")
            del sheet
            model32 = make_model("f32")
            ds32 = DeviceTileDataset(tiles_dev[:512], hr_res=hr_res, lr_scale=4, crappifier=crap, val_split=0.1, rotation=True, device=dev, seed=5)
            t32, _ = run_train(model32, ds32, args.batch, 3, 4)
            v32 = args.batch * 4 / t32
            res["f32_train"] = {"metric": f"HR tiles/sec ({hr_res}^2 4xSR) train", "value": round(v32, 2), "unit": "HR tiles/s", "dtype": "f32",
                                "config": "same workload with compute_dtype = float32 (exact-f32 MFMA: the path the 1e-3 dB PSNR parity tests pin), 4 timed steps",
                                "algorithmic_tflops": round(v32 * TRAIN_GFLOP_PER_TILE / 1e3, 2),
                                "frac_of_f32_mfma_peak": round(v32 * TRAIN_GFLOP_PER_TILE / 1e3 / PEAK_F32_TFLOPS, 4)}
            del model32, ds32
            # ---- c3: RDResUNet, Poisson crappifier, bf16, batch 32 per GPU (321.45 GFLOP per tile)
            torch.manual_seed(0)
            model_rd = RDResUNet().to(dev)
            model_rd.compute_dtype = torch.bfloat16
            ds_rd = DeviceTileDataset(tiles_dev[:1024], hr_res=hr_res, lr_scale=4, crappifier=Poisson(), val_split=0.1, rotation=True, device=dev, seed=8)
            rd_steps = 16
            t_rd, _ = run_train(model_rd, ds_rd, 32, 4, rd_steps)
            v_rd = 32 * rd_steps / t_rd
            res["rd_train"] = {"metric": f"HR tiles/sec ({hr_res}^2 4xSR) train", "value": round(v_rd, 2), "unit": "HR tiles/s", "dtype": "bf16",
                               "ms_per_step": round(1e3 * t_rd / rd_steps, 3),
                               "config": "BASELINE config 3 on one GPU: RDResUNet 1-ch 128^2->512^2, Poisson() device crappifier, MS-SSIM+L1, FusedAdamW, "
                                         f"batch 32, train_paired (hipGraph replay), {rd_steps} timed steps",
                               "roofline": whole_pass_roofline(round(v_rd * 321.45 / 1e3, 2), "whole training step (pair generation, forward, loss, backward, "
                                                               "optimizer); algorithmic FLOPs of the convolutions (SURVEY.md 8d: 321.45 GFLOP per tile)",
                                                               traffic=pass_traffic("rdresunet_train")),
                               "layerwise_bound": layerwise("c3_train", v_rd)}
            del model_rd, ds_rd
            # ---- c4: ResUNet 3-ch multiframe, 256^2 -> 1024^2, MS-SSIM+L1, fp16 storage + dynamic loss scaling, batch 8 per GPU
            torch.manual_seed(0)
            model_c4 = ResUNet(channels=3).to(dev)
            model_c4.compute_dtype = torch.float16
            ds_c4 = DeviceTileDataset(torch.from_numpy(tiles_c4).to(dev), hr_res=1024, lr_scale=4, crappifier=crap, val_split=0.1, rotation=True, device=dev,
                                      seed=9)
            t_c4, _ = run_train(model_c4, ds_c4, 8, 3, 5, loss=SSIMLoss(channels=3, mix=0.8))
            v_c4 = 8 * 5 / t_c4
            res["c4_train"] = {"metric": "HR tiles/sec (1024^2 4xSR) train", "value": round(v_c4, 2), "unit": "HR tiles/s", "dtype": "fp16",
                               "ms_per_step": round(1e3 * t_c4 / 5, 3),
                               "config": "BASELINE config 4 on one GPU: ResUNet 3-ch (3 frames) 256^2->1024^2, AdditiveGaussian(13), MS-SSIM+L1 over 3 channels, "
                                         "fp16 storage + dynamic loss scaling, FusedAdamW, batch 8, train_paired (hipGraph replay), 5 timed steps",
                               "roofline": whole_pass_roofline(round(v_c4 * 774.65 / 1e3, 2), "whole training step; algorithmic FLOPs of the convolutions "
                                                               "(SURVEY.md 8d: 774.65 GFLOP per 1024^2 tile)", traffic=pass_traffic("c4_train")),
                               "layerwise_bound": layerwise("c4_train", v_c4)}
            del model_c4, ds_c4
            # ---- the drop-in API fed by a HOST dataset (what a user of the reference passes: Pillow reduction + numpy crappifier on the CPU,
            # DataLoader workers): PCIe-inclusive, never the headline value
            from pssr2_amd.data import ArrayDataset
            workers = max(2, min(12, (os.cpu_count() or 8) - 4))
            hds = ArrayDataset(tiles_np[:1024], hr_res=hr_res, lr_scale=4, crappifier=crap, val_split=0.05, rotation=True)
            model_h = make_model(args.dtype)
            hw, hs = 6, 20          # one epoch of 30 batches: no epoch boundary (validation pass, loader restart) inside the timed steps
            hs = min(hs, (len(hds) - len(hds.val_idx)) // args.batch - hw)
            if hs < 4:
                raise RuntimeError(f"api_host leg needs at least {(hw + 4) * args.batch} training tiles")
            t_h, _ = run_train(model_h, hds, args.batch, hw, hs, dataloader_kwargs=dict(num_workers=workers, pin_memory=True, persistent_workers=True,
                                                                                   prefetch_factor=4, multiprocessing_context="spawn"))
            v_h = args.batch * hs / t_h
            res["api_host"] = {"metric": f"HR tiles/sec ({hr_res}^2 4xSR) train", "value": round(v_h, 2), "unit": "HR tiles/s", "dtype": args.dtype,
                               "ms_per_step": round(1e3 * t_h / hs, 3), "frac_of_headline": round(v_h / tiles_per_s, 3),
                               "config": f"train_paired(model, ArrayDataset, ...) with dataloader_kwargs num_workers={workers}, pin_memory: pairs made on the "
                                         f"host CPU (Pillow reduction + numpy AdditiveGaussian, as the reference's ImageDataset), one pinned host-to-device "
                                         f"copy (uint8 items, converted on the device) + one hipGraph replay per batch; PCIe-inclusive; {os.cpu_count()} hardware threads present, "
                                         f"{hs} timed steps"}
            del model_h, hds
        except Exception as e:                                     # an extra leg must not take the headline line with it
            res["extras_error"] = f"{type(e).__name__}: {e}"
            import traceback
            traceback.print_exc(file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "resunet":
        res["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()
    if in_sync is False:
        print("bench: the ranks' parameters differ after the timed steps (comm.weights_in_sync = false)", file=sys.stderr)
        raise SystemExit(4)


if __name__ == "__main__":
    main()
