#!/usr/bin/env python3
"""Benchmark of the PSSR2 hot path on MI355X: HR tiles/s of one ResUNet training step.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): ResUNet 1-ch 4xSR, 128^2 -> 512^2, bf16 storage / f32 accumulate,
batch 32 per GPU, SSIMLoss(mix=.8) (MS-SSIM + L1), AdamW.  A step = device-side pair generation
(Pillow-exact 4x reduction + AdditiveGaussian(13) + round/clip) from uint8 HR tiles already resident in
HBM, forward, loss, backward, (gradient all-reduce), optimizer update.  Prints ONE JSON line on rank 0.
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
PEAK_HBM_GBS = 8000.0
DOMINANT_SYMBOL = "conv_igemm_kernelIDF16bLi128ELi0ELi9E"   # conv_igemm_kernel<bf16, BN=128, GEO=0 (8x16 tile), 9 taps>


def pmc_traffic(mode):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/traffic.json, written by tools/summarize_profile.py; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when no such profile has been committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            data = json.load(f)[mode]
        for name, rec in data["kernels"].items():
            if DOMINANT_SYMBOL in name:
                return rec["hbm_bytes_per_launch"], data["source"]
    except (OSError, KeyError, ValueError):
        pass
    return None, None


class ConvTimer:
    """HIP-event timing of the dominant kernel (conv_igemm<bf16,128,geo0>) on the launch stream."""

    def __init__(self):
        self.events = []
        self.orig = None

    def install(self):
        from pssr2_amd import ops
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
            flops = 2.0 * kw["n"] * kw["h"] * w * cout * (w0.taps * min(cin0, w0.k_pad) + kw.get("cin1", 0))
            timer.events.append((e0, e1, flops))
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
        ms = [a.elapsed_time(b) for a, b, _ in self.events]
        fl = [f for _, _, f in self.events]
        tot_ms, tot_fl = sum(ms), sum(fl)
        return dict(launches=len(ms), avg_us=1e3 * tot_ms / len(ms), tflops=tot_fl / (tot_ms * 1e-3) / 1e12,
                    gflop_per_launch=tot_fl / len(ms) / 1e9)


def cpu_baseline(batch=4, lr_res=128):
    """The oracle (CPU restatement, torch fp32) doing the same training step on a bounded sample."""
    from oracle import loss_ref, model_ref
    torch.manual_seed(0)
    sd = model_ref.make_state_dict(seed=1, randomize_bn=False)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=1e-3)
    lr = torch.rand(batch, 1, lr_res, lr_res) * 255
    hr = torch.rand(batch, 1, lr_res * 4, lr_res * 4) * 255
    t0 = time.time()
    y, _ = model_ref.resunet_forward(lr, params, 5, 3, 4, train=True)
    loss = loss_ref.ssim_loss(y / 255, hr / 255, mix=0.8)
    loss.backward()
    opt.step()
    dt = time.time() - t0
    return dict(value=batch / dt, unit="HR tiles/s", cores=torch.get_num_threads(), kind="port",
                sample=f"1 training step (fwd + MS-SSIM/L1 + bwd + AdamW) of the CPU oracle on {batch} tiles {lr_res}^2->{lr_res * 4}^2, fp32, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step")
    ap.add_argument("--lr-res", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "f32"],
                    help="fp16: a static loss scale of 1024 stands in for train_paired's dynamic LossScaler (no host sync in the captured step)")
    ap.add_argument("--channels", type=int, default=1, help="image channels / frames (BASELINE config 4 uses 3)")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "sheet"],
                    help="sheet: BASELINE config 5 end to end on the device (4096^2 LR sheet -> 128^2 tiles, overlap 32 -> predict -> "
                         "uint8 -> overlap-averaged 16384^2-class sheet), one step = one sheet")
    ap.add_argument("--model", default="resunet", choices=["resunet", "rdresunet"],
                    help="resunet = BASELINE.json configs[1] (default, the metric's config); rdresunet = configs[2] (RDNet encoder)")
    ap.add_argument("--crappifier", default="gaussian", choices=["gaussian", "poisson"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a captured hipGraph")
    args = ap.parse_args()

    from pssr2_amd import distributed as D
    # rehearsal knobs (NOT used by the driver): run several ranks on ONE card over gloo to exercise the data-parallel code
    # path on a 1-GPU box, e.g.  PSSR_BENCH_BACKEND=gloo PSSR_BENCH_FORCE_DEVICE=0 python -m torch.distributed.run ...
    force_dev = os.environ.get("PSSR_BENCH_FORCE_DEVICE")
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
    from pssr2_amd.data import DevicePairGenerator, synthetic_em_tile
    from pssr2_amd.models import RDResUNet, ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.util import SSIMLoss
    import numpy as np

    torch.manual_seed(0)
    model = (ResUNet(channels=args.channels) if args.model == "resunet" else RDResUNet(channels=args.channels)).to(dev)
    model.compute_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[args.dtype]
    loss_scale = 1024.0 if args.dtype == "fp16" else 1.0
    D.broadcast_module(model)
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    loss_fn = SSIMLoss(channels=args.channels, mix=0.8)
    hr_res = args.lr_res * 4
    pool_n = 8                                           # distinct synthetic EM tiles per rank (tiled to the batch)
    pool = np.stack([synthetic_em_tile(rank * 100003 + i, hr_res, channels=args.channels) for i in range(pool_n)])
    pool = torch.from_numpy(pool).to(dev)                # uint8 [pool, 1, HR, HR], resident in HBM
    from pssr2_amd import ops
    use_graph = not args.no_graph
    # device-resident counters: nothing that changes from step to step is a kernel argument, so the
    # whole step can be captured once into a hipGraph and replayed (the ~1000 launches per step
    # otherwise cost more host time than the GPU needs to execute them)
    tile_counter = torch.full((1,), rank * 10 ** 9, dtype=torch.int64, device=dev)
    step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    ar = torch.arange(args.batch, device=dev)
    crap = AdditiveGaussian(13, 0, 0) if args.crappifier == "gaussian" else Poisson()
    gen = DevicePairGenerator(4, crap, seed=1234, tile_counter=tile_counter)
    opt.device_state = True

    def next_batch():
        idx = (ar + step_dev) % pool_n
        hr, lr = gen(pool[idx])
        ops.counter_add(tile_counter, args.batch)
        step_dev.add_(1)
        return hr, lr

    def fwd_bwd():
        hr, lr = next_batch()
        hr_hat = model(lr)
        loss = loss_fn(hr_hat / 255, hr / 255)
        (loss * loss_scale if loss_scale != 1.0 else loss).backward()
        return loss

    def reduce_and_update(zero=True):
        if world > 1:
            flat = model._engine._flat_grad
            torch.distributed.all_reduce(flat)        # SUM; the mean's 1/world is folded into the optimizer's gradient scale
        opt.step(grad_scale=1.0 / (loss_scale * world))
        if zero:
            opt.zero_grad()

    def infer_body():
        _, lr = next_batch()
        with torch.no_grad():
            y = model(lr)
            out = torch.empty(y.shape, dtype=torch.uint8, device=dev)
            ops.clip_u8(y, out)
        return out

    graphs = {}

    def capture_split(eng):
        """fwd + loss + first part of the backward | rest of the backward, as two hipGraphs sharing one memory pool.  The engine's
        backward is driven directly (d loss / d output from autograd.grad) so that the capture switches graphs on this thread."""
        def body(cb):
            opt.zero_grad()
            hr, lr = next_batch()
            hr_hat = model(lr)
            loss = loss_fn(hr_hat / 255, hr / 255)
            (dout,) = torch.autograd.grad(loss * loss_scale if loss_scale != 1.0 else loss, hr_hat)
            eng.backward(dout, split_cb=cb)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                body(lambda: None)
                if world > 1:
                    torch.distributed.all_reduce(eng._flat_grad)     # a real data-parallel step: the ranks must stay identical
                opt.step(grad_scale=1.0 / (loss_scale * world))      # the weights must be stale at capture time so that the
        torch.cuda.current_stream().wait_stream(side)                # forward's re-pack of every conv weight is part of the graph
        torch.cuda.synchronize()
        stale = [c for c in eng._convs.values() for key in c.packed if c.version.get(key) != c.m.weight._version]
        if not stale:
            raise RuntimeError("weights not stale before capture: the packed-weight refresh would be missing from the graph")
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        state = {"g": None}

        def switch():
            g1.capture_end(); state["g"] = None
            g2.capture_begin(pool=pool); state["g"] = g2
        with torch.cuda.stream(cap):
            try:
                g1.capture_begin(pool=pool); state["g"] = g1
                body(switch)
                g2.capture_end(); state["g"] = None
            except Exception:
                if state["g"] is not None:          # leave no stream in capture mode behind: the caller falls back to one graph
                    try:
                        state["g"].capture_end()
                    except Exception:
                        pass
                raise
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        return g1, g2, eng.grad_split_offset()

    def capture(name, body):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        graphs[name] = g

    if args.mode == "train":
        model.train()
        split_graph = world > 1 or os.environ.get("PSSR_BENCH_SPLIT_GRAPH") == "1"      # rehearsal of the N>1 structure on one rank
        if use_graph and not split_graph:
            def whole():
                fwd_bwd()
                reduce_and_update()
            capture("step", whole)
            fn = lambda s: graphs["step"].replay()
        elif use_graph:
            # data-parallel: the gradient all-reduce stays outside the captured region
            def eager_step():
                fwd_bwd()
                reduce_and_update()
            for _ in range(2):
                eager_step()
            def fwd_bwd_fresh():
                opt.zero_grad()               # Python-only (sets .grad = None): the engine then publishes views of its flat buffer
                fwd_bwd()
            eng = model._engine
            split = None
            if os.environ.get("PSSR_BENCH_OVERLAP", "1") != "0":
                # Two graphs split where ~85 % of the gradient bytes (reconstruction, decoder, deepest encoder block) are final: their
                # all-reduce is launched between the two replays and runs on RCCL's stream under the rest of the backward pass.
                try:
                    split = capture_split(eng)
                except Exception as e:                       # any capture problem: the single-graph path below
                    if rank == 0:
                        print(f"[bench] split capture unavailable ({type(e).__name__}: {e}); all-reduce after the backward graph", file=sys.stderr)
                    split = None
                    torch.cuda.synchronize()
            if split is None:
                capture("fwd_bwd", fwd_bwd_fresh)
            # Replays run no Python, so the .grad views published during capture must stay in place: no zero_grad()
            # between steps (the captured backward zeroes the flat buffer itself), and FusedAdamW consumes the flat
            # buffer the all-reduce just averaged.
            assert all(p.grad is not None and p.grad._base is eng._flat_grad for p in model.parameters())

            if split is not None:
                g1, g2, a0 = split
                flat = eng._flat_grad

                def fn(s):
                    g1.replay()
                    h1 = torch.distributed.all_reduce(flat[a0:], async_op=True) if world > 1 else None    # waits for g1 on RCCL's stream, runs under g2
                    g2.replay()
                    if world > 1:
                        h2 = torch.distributed.all_reduce(flat[:a0], async_op=True)
                        h1.wait(), h2.wait()
                    opt.step(grad_scale=1.0 / (loss_scale * world))
            else:
                def fn(s):
                    graphs["fwd_bwd"].replay()
                    reduce_and_update(zero=False)
        else:
            if world > 1:
                model._engine.attach_reducer()

            def fn(s):
                fwd_bwd()
                if world > 1:
                    opt.step(grad_scale=1.0 / loss_scale), opt.zero_grad()
                else:
                    reduce_and_update()
    elif args.mode == "sheet":
        from pssr2_amd.predict import predict_sheet
        model.eval()
        rng = np.random.default_rng(7 + rank)
        sheet = torch.from_numpy(rng.integers(0, 256, size=(args.channels, 4096, 4096), dtype=np.uint8)).to(dev)
        sheet_tiles = ((4096 - args.lr_res) // (args.lr_res - 32) + 1) ** 2
        fn = lambda s: predict_sheet(model, sheet, tile_res=args.lr_res, overlap=32, margin=8, batch_size=max(args.batch, 128), device=dev, to_numpy=False)
        use_graph = False
    else:
        model.eval()
        if use_graph:
            capture("infer", infer_body)
            fn = lambda s: graphs["infer"].replay()
        else:
            fn = lambda s: infer_body()

    for s in range(args.warmup):
        fn(s)
    timer = ConvTimer()
    if rank == 0 and args.dtype == "bf16" and not use_graph:
        timer.install()       # eager mode: HIP events around every launch of the dominant kernel, in the timed region

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        fn(args.warmup + s)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()

    measured_in = "timed region"
    if rank == 0 and use_graph and args.dtype in ("bf16", "fp16"):      # both 16-bit MFMA paths have the same dense peak
        # the timed region replayed a hipGraph; time the same kernels once more in an instrumented eager pass
        timer.install()
        eager = (lambda: (fwd_bwd(), opt.zero_grad())) if args.mode == "train" else (infer_body if args.mode == "infer" else (lambda: fn(0)))
        # The roofline figure is about the kernel itself: in this pass the weight-gradient launches stay on the launch stream
        # (in the timed region they run on a second stream and share the chip with the kernel being timed, which would
        # stretch its HIP-event duration by whatever they take from it).
        eng = getattr(model, "_engine", None)
        side = getattr(eng, "side_wgrad", False)
        if eng is not None:
            eng.side_wgrad = False
        for _ in range(2):
            eager()
        torch.cuda.synchronize()
        if eng is not None:
            eng.side_wgrad = side
        timer.remove()
        measured_in = ("instrumented eager pass after the timed region, kernels one at a time on the launch stream (the timed region replays "
                       "the same kernels from a hipGraph, with the weight-gradient kernels overlapping on a second stream)")
    if os.environ.get("PSSR_BENCH_CHECKSUM") == "1":        # rehearsal aid: compare launch structures by their effect on the weights
        cs = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
        print(f"[bench] rank {rank} param checksum {cs:.6f}", file=sys.stderr)
    if rank == 0:
        conv = timer.summary()
        per_step = sheet_tiles if args.mode == "sheet" else args.batch
        tiles_per_s = world * per_step * args.steps / elapsed
        gflop_tile = TRAIN_GFLOP_PER_TILE if args.mode == "train" else FWD_GFLOP_PER_TILE
        if args.mode == "sheet":
            args.batch = max(args.batch, 128)
        if args.model == "rdresunet":      # 53.575 GMAC fwd per tile
            gflop_tile = 321.45 if args.mode == "train" else 107.15
        scale = (args.lr_res / 128) ** 2
        res = {
            "metric": f"HR tiles/sec (512^2 4xSR) {args.mode}",
            "value": round(tiles_per_s, 2), "unit": "HR tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{'ResUNet' if args.model == 'resunet' else 'RDResUNet'} {args.channels}-ch 4xSR {args.lr_res}^2->{hr_res}^2 {args.mode}, batch {args.batch}/GPU, "
                                   f"{'AdditiveGaussian(13)' if args.crappifier == 'gaussian' else 'Poisson()'} device crappifier, MS-SSIM+L1 (mix .8), AdamW",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "launch": "hipGraph replay" if use_graph else "eager"},
            "step_compute": {"algorithmic_tflops": round(tiles_per_s * gflop_tile * scale / 1e3 / world, 2),
                             "frac_of_bf16_peak": round(tiles_per_s * gflop_tile * scale / 1e3 / world / PEAK_BF16_TFLOPS, 4)},
        }
        traffic, traffic_src = pmc_traffic(args.mode if args.model == 'resunet' else f'{args.model}_{args.mode}')
        if args.dtype != "bf16" or args.lr_res != 128 or args.channels != 1 or args.batch != 32:
            traffic, traffic_src = None, None          # the committed counter passes were taken on the default workload only
        if conv:
            res["roofline"] = {"bound": "mfma", "kernel": f"conv_igemm_kernel<{'f16' if args.dtype == 'fp16' else 'bf16'},BN=128,8x16 tile,9 taps> (3x3 conv fwd+dgrad, Cout>64)",
                               "achieved": round(conv["tflops"], 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(conv["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                               "traffic_source": traffic_src,
                               "launches": conv["launches"], "avg_launch_us": round(conv["avg_us"], 1), "measured_in": measured_in,
                               "algorithmic_gflop_per_launch": round(conv["gflop_per_launch"], 2)}
        if not args.no_cpu_baseline and world == 1 and args.model == "resunet":
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
