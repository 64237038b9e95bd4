"""hipGraph replay behind ``train_paired`` / ``predict_images`` (no reference counterpart: the reference issues every op from Python).

A training step of the MI355X path is 300-700 kernel launches; issued one by one through ctypes they cost more host time
than the GPU needs to run them (c2: 33 ms vs 13 ms per step).  When the dataset can produce its batches on the device from
device-resident inputs only (``DeviceTileDataset.device_batch``) nothing that changes from step to step is a kernel
argument -- the batch's gather table is read through a device cursor, the Philox tile counter, the AdamW step count and the
learning rate live in HBM -- so the drivers run the first two batches of the first epoch eagerly (real steps: they are also
the warm-up), capture the third into a ``torch.cuda.CUDAGraph`` and replay it for every further full batch of every epoch.
Partial last batches, other datasets, ``extra`` losses and unsupported models take the ordinary loop.

What a replay does not do is run Python: with the optimizer inside the graph ``.grad`` is ``None`` between steps, as after the
reference's ``zero_grad()`` (the captured backward zeroes the engine's flat gradient buffer itself), parameter ``_version``
counters do not move
(``Engine.mark_weights_changed`` invalidates the packed-weight caches when control returns to eager code), and a learning-rate
scheduler reaches the captured optimizer through ``FusedAdamW.sync_device_lr``.
"""
from __future__ import annotations

import os

import torch

from . import distributed as D
from .util import SSIMLoss as _SSIMLoss


# stream capture in thread-local mode: a DataLoader's pin-memory thread allocates pinned host memory while the main thread captures
# (in the default global mode any such call from ANY thread invalidates the capture)
_CAPTURE_MODE = "thread_local"


def enabled():
    return os.environ.get("PSSR_GRAPH", "1") != "0"


def supports(model, dataset, device):
    """The replay path needs an engine-backed model on the GPU and a dataset that batches on the device.  ``model.sync_bn`` with more
    than one rank stays on the eager loop: its BatchNorm statistics are all-reduced in the middle of the forward and backward passes,
    which a captured graph cannot hold with gloo (host-synchronising) and must not hold with RCCL (captured collectives would share
    the communicator with the un-captured gradient all-reduce issued between the two replays: per-rank ordering is not guaranteed)."""
    if getattr(model, "sync_bn", False) and D.rank_world()[1] > 1:
        return False
    return (enabled() and getattr(model, "_engine", None) is not None and hasattr(dataset, "device_batch") and hasattr(dataset, "draw_items")
            and torch.device(device).type == "cuda" and getattr(dataset, "extra_hr_files", None) is None)


def supports_host(model, dataset, device):
    """Replay for HOST-fed batches: any dataset that goes through a DataLoader (the reference's ImageDataset / SlidingDataset, their
    crappifiers running on the CPU, optionally in worker processes) with an engine-backed model on the GPU.  The step is captured once
    over static input buffers; a batch then costs one asynchronous host-to-device copy and one graph replay instead of ~400 ctypes
    launches (pssr/train.py:75-103, pssr/predict.py:49-61).  PSSR_HOST_GRAPH=0 keeps the launch-by-launch loop."""
    if getattr(model, "sync_bn", False) and D.rank_world()[1] > 1:
        return False
    return (enabled() and os.environ.get("PSSR_HOST_GRAPH", "1") != "0" and getattr(model, "_engine", None) is not None
            and torch.device(device).type == "cuda" and getattr(dataset, "extra_hr_files", None) is None)


class _capture:
    """``torch.cuda.graph`` with the collector switched off for the duration: the backward pass of a captured step runs on autograd's
    worker thread, and a collection there that happens to free an older captured graph (its private memory pool) while THIS capture is
    under way asserts inside torch's allocator and takes the process down (seen when a failed test had left steppers behind)."""

    def __init__(self, graph):
        self.ctx = torch.cuda.graph(graph, capture_error_mode=_CAPTURE_MODE)

    def __enter__(self):
        import gc
        gc.collect()
        self.was_on = gc.isenabled()
        gc.disable()
        try:
            return self.ctx.__enter__()
        except BaseException:
            if self.was_on:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was_on:
                gc.enable()


def _all_ranks_ok(ok, device):
    """True when ``ok`` holds on every rank (MIN all-reduce of a flag; ``device`` None = single process): every rank must take the same
    path -- replay or launch by launch -- because the collectives of a step are issued from it."""
    if device is None or not D.is_distributed():
        return ok
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    return float(flag) >= 1.0


def _feedable(t):
    """A host batch as the feed takes it: float32, or uint8 as it is (a dataset of this package in compact mode: converted on the device)."""
    return t if t.dtype in (torch.float32, torch.uint8) else t.float()


class _HostFeed:
    """Static device inputs of a captured graph, refilled from host batches: the host-to-device copy of batch i + 1 runs on its own
    stream into one of two staging buffers while the graph of batch i is still running; a device-to-device copy (~15 us for a c2
    batch) moves it into the graph's buffers in stream order.  Pinned host tensors (DataLoader(pin_memory=True)) copy without a
    host-side staging pass."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(self.device)
        self.static, self.stage, self.free_ev, self.keep, self.k = None, [None, None], [None, None], [None, None], 0

    def matches(self, tensors):
        return (self.static is not None and len(tensors) == len(self.static)
                and all(t.shape == s.shape and t.dtype == g.dtype for t, s, g in zip(tensors, self.static, self.stage[0])))

    def allocate(self, tensors):
        # the staging buffers take the batch as it arrives (float32, or the uint8 of a dataset in compact mode: a quarter of the bytes over
        # PCIe); the copy into the graph's float32 buffers converts on the device
        mk = lambda f32: [torch.empty(t.shape, dtype=torch.float32 if f32 else t.dtype, device=self.device) for t in tensors]
        self.static, self.stage = mk(True), [mk(False), mk(False)]

    def push(self, tensors):
        k = self.k
        self.k ^= 1
        main = torch.cuda.current_stream()
        if any(t.is_cuda for t in tensors):
            # a dataset that yields DEVICE tensors (DeviceTileDataset through a DataLoader, a user dataset on the GPU): the batch was made
            # and collated on the launch stream; the side-stream copy must not read it before it is written
            self.copy_stream.wait_stream(main)
        with torch.cuda.stream(self.copy_stream):
            if self.free_ev[k] is not None:
                self.copy_stream.wait_event(self.free_ev[k])           # the previous content of this staging buffer has been consumed
            for dst, src in zip(self.stage[k], tensors):
                dst.copy_(src, non_blocking=True)
                if src.is_cuda:
                    src.record_stream(self.copy_stream)            # its block is not handed out again before this copy has run
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        main.wait_event(ready)
        for dst, src in zip(self.static, self.stage[k]):
            dst.copy_(src)
        done = torch.cuda.Event()
        done.record(main)
        self.free_ev[k], self.keep[k] = done, tensors                  # (the host tensors stay alive until their copy has run)
        return self.static


class _Cursor:
    """Gather table of an epoch (device int64 [n, 3]) read ``batch`` rows at a time through a device-side cursor."""

    def __init__(self, dataset, batch, capacity, device):
        self.dataset, self.batch = dataset, batch
        self.table = torch.zeros(max(capacity, batch), 3, dtype=torch.int64, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=device)
        self.ar = torch.arange(batch, device=device)

    def load(self, order):
        rows = self.dataset.draw_items(order)
        if rows.shape[0] > self.table.shape[0]:
            raise RuntimeError("epoch longer than the captured gather table")
        self.table[:rows.shape[0]].copy_(rows)
        self.cursor.zero_()
        return rows.shape[0]

    def next_rows(self):
        rows = self.table[self.ar + self.cursor]
        self.cursor.add_(self.batch)
        return rows

    def tail_rows(self, start, count):
        return self.table[start:start + count]


class TrainStepper:
    """One training step of ``train_paired`` (pssr/train.py:86-103) as a replayed graph."""

    WARM = 2

    def __init__(self, model, dataset, batch_size, loss_fn, optim, clamp, image_range, scaler, capacity, device, host=False):
        from .optim import FusedAdamW
        self.model, self.dataset, self.batch, self.loss_fn, self.optim = model, dataset, batch_size, loss_fn, optim
        self.clamp, self.image_range, self.scaler = clamp, image_range, scaler
        self.engine = model._engine
        self.rank, self.world = D.rank_world()
        self.fused = isinstance(optim, FusedAdamW)
        # fp16 storage: the loss scaler's policy runs on the device (LossScaler.step_dev), so the optimizer stays inside the graph
        self.amp_dev = scaler is not None and self.fused and self.world == 1 and os.environ.get("PSSR_AMP_DEVICE", "1") != "0"
        self.in_graph_optim = self.fused and self.world == 1 and (scaler is None or self.amp_dev)
        if self.fused:
            optim.device_state = True
        self.scale_dev = (scaler.to_device(device).scale_dev if self.amp_dev else torch.ones(1, device=device)) if scaler is not None else None
        self.host = host               # batches arrive from a DataLoader (host tensors) instead of being made on the device
        self.cur = _Cursor(dataset, batch_size, capacity, device) if not host else None
        self.feed = _HostFeed(device) if host else None
        self.device = torch.device(device)
        self.graph, self.outs, self.eager_done = None, None, 0
        self.eager_only = False        # set when the step could not be captured (see step): ordinary launches from then on
        self._reduced = False
        # data-parallel: the step is captured as TWO graphs split where the gradients of the reconstruction head, the decoder and
        # the deepest encoder block (the tail of the engine's flat buffer, ~85 % of a ResUNet's bytes) are final; their all-reduce is
        # launched between the two replays and runs on RCCL's stream under the rest of the backward pass (PSSR_OVERLAP=0: one graph,
        # one all-reduce after it)
        # PSSR_DDP_SPLIT=0 (older name: PSSR_OVERLAP=0) is the fallback should the two-graph schedule misbehave on a real RCCL node
        self.split = self.world > 1 and os.environ.get("PSSR_DDP_SPLIT", os.environ.get("PSSR_OVERLAP", "1")) != "0"
        self.graph2, self.split_at = None, 0
        self.comm_events = [] if os.environ.get("PSSR_COMM_STATS") == "1" else None     # (after graph 2, after the all-reduces) per step
        self.engine.last_train_stepper = self       # the stepper of the model's most recent train_paired call (bench.py / tests read its statistics)

    def begin_epoch(self, order):
        self.n = self.cur.load(order)
        self.pos = 0
        return (self.n + self.batch - 1) // self.batch

    # ---- the step itself (what the slow loop of train_paired does, on device-made batches)
    def _inputs(self, rows=None):
        """(hr, lr) of the step being issued: made on the device from the next rows of the gather table, or the static buffers a
        host batch was copied into."""
        if self.host:
            return self.feed.static if rows is None else rows
        return self.dataset.device_batch(self.cur.next_rows() if rows is None else rows)

    def _fwd_bwd(self, rows):
        hr, lr = self._inputs(rows)
        hr_hat = self.model(lr)
        if self.clamp:
            hr_hat = torch.clamp(hr_hat, 0, self.image_range)
        if isinstance(self.loss_fn, _SSIMLoss) and os.environ.get("PSSR_FUSED_DIV", "1") != "0":
            loss = self.loss_fn.forward_divided(hr_hat, hr, self.image_range)       # same values, three passes over 33 MB tensors fewer
        else:
            loss = self.loss_fn(hr_hat / self.image_range, hr / self.image_range)
        (loss * self.scale_dev if self.scale_dev is not None else loss).backward()
        return hr, lr, hr_hat, loss

    def _body(self, rows=None):
        out = self._fwd_bwd(rows)
        if self.in_graph_optim:
            if self.amp_dev:
                self.scaler.step_dev(self.optim, self.engine._flat_grad)
            else:
                self.optim.step()
            self.optim.zero_grad()      # Python only (.grad = None): the next backward publishes views of the flat buffer again
        return out

    def _after(self):
        """Everything of a step that stays outside the graph: (all-reduce,) optimizer, zero_grad -- the reference's order."""
        eng, opt = self.engine, self.optim
        if self.in_graph_optim:
            return
        if self.world > 1 and not self._reduced:
            D.sum_flat(eng._flat_grad)
        self._reduced = False
        eng.publish_grads()             # a replayed backward wrote the flat buffer but ran no Python
        if self.scaler is not None:
            if self.world > 1:
                eng._flat_grad.mul_(1.0 / self.world)
            self.scaler.step(opt, list(self.model.parameters()))
            self.scale_dev.fill_(self.scaler.scale_value)
        elif self.fused:
            opt.step(grad_scale=1.0 / self.world)
        else:
            if self.world > 1:
                eng._flat_grad.mul_(1.0 / self.world)
            opt.step()
        opt.zero_grad()

    def step(self, batch=None):
        """Next batch of the epoch (``batch``: the DataLoader's (hr, lr) host tensors in host mode).  Returns (hr, lr, hr_hat, loss)
        device tensors (static buffers once the graph is captured)."""
        if self.host:
            hr, lr = (_feedable(t) for t in batch)
            if self.feed.static is None and hr.shape[0] == self.batch:
                self.feed.allocate((hr, lr))
            if hr.shape[0] != self.batch or not self.feed.matches((hr, lr)):      # partial last batch / another tile size: ordinary launches
                self._leave_graph()
                out = self._body((hr.to(self.device).float(), lr.to(self.device).float()))
                self._after()
                self._leave_graph()
                return out
            self.feed.push((hr, lr))
        else:
            left = self.n - self.pos
            if left < self.batch:                               # partial last batch: ordinary launches, its own engine plan
                self._leave_graph()
                out = self._body(self.cur.tail_rows(self.pos, left))
                self._after()
                self.pos += left
                self._leave_graph()
                return out
            self.pos += self.batch
        if self.graph is None and self.eager_done < self.WARM:
            if self.scaler is not None and not self.amp_dev:
                self.scale_dev.fill_(self.scaler.scale_value)
            out = self._body()                              # real steps that double as warm-up (weights end up stale: the capture
            self._after()                                   # below then contains the packed-weight refresh)
            self.eager_done += 1
            return out
        if self.graph is None and not self.eager_only:
            torch.cuda.synchronize()
            if self.split:
                err = None
                try:
                    self._capture_split()
                except Exception as e:                          # any capture problem: one graph + one all-reduce after it
                    err = e
                # every rank must issue the same collectives per step: the split is kept only if EVERY rank captured it
                ok = torch.tensor([0.0 if err is not None else 1.0], device=self.engine._flat_grad.device)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
                if float(ok) < 1.0:
                    if self.rank == 0 or err is not None:
                        why = f"{type(err).__name__}: {err}" if err is not None else "another rank could not capture it"
                        print(f"[pssr2_amd] rank {self.rank}: split capture unavailable ({why}); all-reduce after the backward graph", flush=True)
                    self.split, self.graph, self.graph2 = False, None, None
                    self.engine.reset_backward_state()          # an aborted capture may have left the side stream mid-backward
                    torch.cuda.synchronize()
            if self.graph is None:
                g = torch.cuda.CUDAGraph()
                err = None
                try:
                    with _capture(g):
                        self.outs = self._body()
                except Exception as e:
                    # a loss_fn that synchronises with the host (.item(), data-dependent Python control flow) or an op that cannot be
                    # captured: the reference accepts any nn.Module loss (pssr/train.py:19), so run this stepper launch by launch
                    err = e
                if _all_ranks_ok(err is None, self.engine._flat_grad.device if self.world > 1 else None):
                    self.graph = g
                else:
                    self._capture_failed(err)
        if self.eager_only:
            out = self._body()
            self._after()
            self._leave_graph()
            return out
        if self.graph2 is not None:
            flat, a0 = self.engine._flat_grad, self.split_at
            self.graph.replay()
            h1 = D.sum_flat(flat[a0:], async_op=True)       # waits for graph 1 on RCCL's stream, runs under graph 2
            self.graph2.replay()
            h2 = D.sum_flat(flat[:a0], async_op=True)
            if self.comm_events is not None:
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
            h1.wait(), h2.wait()
            if self.comm_events is not None:
                eb.record()
                self.comm_events.append((ea, eb))
            self._reduced = True
        else:
            self.graph.replay()
        self._after()
        return self.outs

    def _capture_split(self):
        """forward + loss + the first part of the backward | the rest of the backward, as two hipGraphs sharing one memory pool.  The
        engine's backward is driven directly (d loss / d output from autograd.grad) so that the capture can switch graphs inside it."""
        eng = self.engine
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        state = {"g": None}

        def switch():
            g1.capture_end()
            state["g"] = None
            g2.capture_begin(pool=pool, capture_error_mode=_CAPTURE_MODE)
            state["g"] = g2
        import gc
        gc.collect()
        gc_on = gc.isenabled()
        gc.disable()
        with torch.cuda.stream(cap):
            try:
                g1.capture_begin(pool=pool, capture_error_mode=_CAPTURE_MODE)
                state["g"] = g1
                hr, lr = self._inputs()
                raw = self.model(lr)
                hr_hat = torch.clamp(raw, 0, self.image_range) if self.clamp else raw     # (the engine gets d loss / d raw: through the clamp)
                loss = self.loss_fn(hr_hat / self.image_range, hr / self.image_range)
                (dout,) = torch.autograd.grad(loss * self.scale_dev if self.scale_dev is not None else loss, raw)
                eng.backward(dout, split_cb=switch)
                if state["g"] is not g2:
                    raise RuntimeError("the engine's backward did not reach its split point")
                g2.capture_end()
                state["g"] = None
            except Exception:
                if state["g"] is not None:          # leave no stream in capture mode behind
                    try:
                        state["g"].capture_end()
                    except Exception:
                        pass
                raise
            finally:
                if gc_on:
                    gc.enable()
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        self.graph, self.graph2, self.split_at = g1, g2, eng.grad_split_offset()
        self.outs = (hr, lr, hr_hat.detach(), loss.detach())

    def _capture_failed(self, err):
        """The step cannot be replayed: forget the aborted capture and keep going launch by launch (what PSSR_HOST_GRAPH=0 /
        PSSR_GRAPH=0 select up front).  The batch whose capture failed is then run eagerly by ``step``."""
        why = f"{type(err).__name__}: {err}" if err is not None else "another rank could not capture it"
        if self.rank == 0 or err is not None:
            print(f"[pssr2_amd] rank {self.rank}: the training step could not be captured into a hipGraph ({why}); "
                  "running it launch by launch (PSSR_GRAPH=0 selects this from the start)", flush=True)
        self.graph = self.graph2 = None
        self.split, self.eager_only = False, True
        self.engine.reset_backward_state()
        self.engine.mark_weights_changed()      # the aborted capture marked packed weights / folded BatchNorms fresh without running their kernels
        for p in self.model.parameters():
            p.grad = None
        torch.cuda.synchronize()

    def exposed_comm_ms(self, last=None):
        """Mean time per step the launch stream sat waiting for the gradient all-reduces after the backward graph had finished
        (PSSR_COMM_STATS=1): the part of the communication that the backward pass did not hide."""
        ev = self.comm_events or []
        ev = ev[-last:] if last else ev
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None

    def _leave_graph(self):
        """Control goes back to eager code (or to another graph): its caches must not trust parameter versions."""
        self.engine.mark_weights_changed()

    def finish(self):
        self._leave_graph()
        if self.amp_dev:
            self.scaler.pull()          # scale / good / skipped counters back on the host attributes (one synchronisation per epoch)


class EvalStepper:
    """No-grad forward (+ loss) over device-made batches as a replayed graph: the validation loop of ``train_paired``
    (pssr/train.py:122-148) and the prediction loop of ``predict_images`` (pssr/predict.py:52-60)."""

    def __init__(self, model, dataset, batch_size, device, loss_fn=None, clamp=False, image_range=255, to_u8=False, weights_move=True, host=False):
        self.model, self.dataset, self.batch, self.loss_fn = model, dataset, batch_size, loss_fn
        self.host, self.device = host, torch.device(device)
        self.feed = _HostFeed(device) if host else None
        self.clamp, self.image_range, self.to_u8 = clamp, image_range, to_u8
        self.engine = model._engine
        self.weights_move = weights_move           # True: the weights change between uses (validation inside training)
        # any order the drivers may ask for (validation split, a val_idx enlarged later to predict every image) fits a table of
        # len(dataset) rows
        self.cur = _Cursor(dataset, batch_size, max(len(dataset), len(dataset.val_idx), batch_size), device) if not host else None
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=device)
        self.graph, self.outs, self.eager_done = None, None, 0
        self.eager_only = False
        self.sig = None

    def _signature(self):
        """Everything a captured eval forward depends on besides its inputs (weights_move = False: the graph holds no
        packed-weight refresh, so a change has to be seen here and answered by one eager batch, which refreshes the shared buffers)."""
        m = self.model
        return (self.engine._wepoch[0], tuple(p._version for p in m.parameters()), tuple(b._version for b in m.buffers()))

    def begin(self, order=None):
        if not self.host:
            self.n = self.cur.load(order)
            self.pos = 0
        self.loss_sum.zero_()
        self.count = 0
        if self.weights_move:
            self.engine.mark_weights_changed()
        elif self._signature() != self.sig:
            self.eager_done = 0             # the first full batch of this pass runs eagerly: packed weights / folded BatchNorm are refreshed
        return (self.n + self.batch - 1) // self.batch if not self.host else None

    def _run(self, rows):
        from . import ops
        with torch.no_grad():
            if self.host:
                item = self.feed.static if rows is None else rows
                hr, lr = (None, item[0]) if len(item) == 1 else item
            else:
                item = self.dataset.device_batch(self.cur.next_rows() if rows is None else rows)
                hr, lr = (None, item) if self.dataset.is_lr else item
            hr_hat = self.model(lr)
            if self.clamp:
                hr_hat = torch.clamp(hr_hat, 0, self.image_range)
            loss = None
            if self.loss_fn is not None:
                loss = self.loss_fn(hr_hat / self.image_range, hr / self.image_range)
                self.loss_sum.add_(loss.detach().float().reshape(1))
            u8 = None
            if self.to_u8:
                y = hr_hat.detach().contiguous().float()
                u8 = torch.empty(y.shape, dtype=torch.uint8, device=y.device)
                ops.clip_u8(y, u8)
        return hr, lr, hr_hat, loss, u8

    def step(self, batch=None):
        """``batch`` (host mode): the DataLoader's item -- (hr, lr) host tensors, or (lr,) for an LR-only dataset."""
        if self.host:
            batch = tuple(_feedable(t) for t in batch)
            if self.feed.static is None and batch[-1].shape[0] == self.batch:
                self.feed.allocate(batch)
            self.count += 1
            if batch[-1].shape[0] != self.batch or not self.feed.matches(batch):
                if self.weights_move:
                    self.engine.mark_weights_changed()
                return self._run(tuple(t.to(self.device).float() for t in batch))
            self.feed.push(batch)
        else:
            left = self.n - self.pos
            if left < self.batch:
                if self.weights_move:
                    self.engine.mark_weights_changed()
                out = self._run(self.cur.tail_rows(self.pos, left))
                self.pos += left
                self.count += 1
                return out
            self.pos += self.batch
            self.count += 1
        if self.eager_done < 1 or (self.eager_only and self.graph is None):
            self.eager_done += 1
            out = self._run(None)
            self.sig = self._signature()
            return out
        if self.graph is None:
            torch.cuda.synchronize()
            if self.weights_move:
                self.engine.mark_weights_changed()          # capture the packed-weight refresh and the BatchNorm folds too
            g = torch.cuda.CUDAGraph()
            try:
                with _capture(g):
                    self.outs = self._run(None)
                self.graph = g
            except Exception as e:          # e.g. a loss_fn that synchronises with the host: ordinary launches for this stepper
                print(f"[pssr2_amd] the evaluation pass could not be captured into a hipGraph ({type(e).__name__}: {e}); "
                      "running it launch by launch", flush=True)
                self.eager_only = True
                self.engine.mark_weights_changed()      # (the aborted capture marked caches fresh without running their kernels)
                torch.cuda.synchronize()
        if self.eager_only:
            if self.weights_move:
                self.engine.mark_weights_changed()
            return self._run(None)
        self.graph.replay()
        return self.outs

    def mean_loss_stat(self):
        """[sum of batch losses, number of batches] on the device."""
        return torch.stack([self.loss_sum[0], torch.tensor(float(self.count), device=self.loss_sum.device)])
