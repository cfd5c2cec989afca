"""Data-parallel gradient averaging for one 8-GPU MI355X node (replaces the DistributedDataParallel wrap of the
reference, glow_tts_train/__main__.py:83-88,268-271).

One process per GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI on ROCm).  The reference leaves
bucketing to DDP's default 25 MB buckets; here the buckets are *slices of the optimizer's flat gradient buffer*
cut at flow-block boundaries (one [ActNorm, InvConvNear, CouplingBlock] block ~ 1.79 M parameters = 7.1 MB at the
default width; TWO blocks per bucket, see default_bucket_key), so

  * nothing is copied into or out of bucket storage — RCCL reduces the gradient memory in place;
  * a block's all-reduce is issued the moment its last gradient has been accumulated, i.e. while the backward of
    the blocks below it (and then of the encoder) is still running; RCCL runs on its own stream, so the
    collective overlaps the remaining flow backward;
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): six 14.3 MB decoder messages + four encoder messages (28.8 MB
    together) keep each ring step large enough to be bandwidth- rather than latency-bound without delaying the first
    launch, and leave only the encoder head's bucket in flight when backward ends (10 collectives per step).

Semantics match DDP: gradients are averaged over ranks (each rank normalises its loss by its own mask sums,
utils.py:19-21,27) and parameters are broadcast from rank 0 once at start (which is also what makes rank 0's
data-dependent ActNorm initialisation win).
"""
from __future__ import annotations

import contextlib
import os
import queue
import re
import threading
import typing

import torch
import torch.distributed as dist

from . import _hip, convops

_FLOW_RE = re.compile(r"^(?:module\.)?decoder\.flows\.(\d+)\.")
_ENC_FFN_RE = re.compile(r"^(?:module\.)?encoder\.encoder\.ffn_layers\.(\d+)\.")


def default_bucket_key(name: str) -> str:
    """TWO flow blocks per bucket (14.3 MB) and four buckets for the text encoder (28.8 MB), cut where its parameters are
    contiguous in construction order: embedding + prenet + attention stack, FFN layers 0-2, FFN layers 3-5, tail
    (projections, duration predictor) — 6 + 4 = 10 collectives per step at the default size.  Fewer, larger messages are
    the right shape for xGMI (a ring step is per-link bound, ~153 GB/s: a 14 MB bucket is ~0.17 ms on the wire at 8 ranks
    while each launch costs ~50 us of host time on the backward thread); the decoder buckets complete in reverse block
    order during the backward, and only the encoder head is still in flight when it ends."""
    m = _FLOW_RE.match(name)
    if m:
        return f"dec{int(m.group(1)) // 6:03d}"
    m = _ENC_FFN_RE.match(name)
    if m:
        return f"enc.ffn{int(m.group(1)) // 3}"
    core = name[len("module."):] if name.startswith("module.") else name
    if core.startswith("encoder."):
        if core.startswith("encoder.emb.") or core.startswith("encoder.pre.") \
                or core.startswith("encoder.encoder.attn_layers.") or core.startswith("encoder.encoder.norm_layers_1."):
            return "enc.head"
        return "enc.tail"
    return "misc"


class Bucket(typing.NamedTuple):
    key: str
    lo: int        # element range [lo, hi) inside the flat gradient buffer
    hi: int
    n_params: int


class FlowBlockReducer:
    """Bucketed, backward-overlapped gradient all-reduce over the flat gradient buffer of `optimize.FlatAdam`."""

    def __init__(self, model: torch.nn.Module, optimizer, process_group=None,
                 bucket_key: typing.Callable[[str], str] = default_bucket_key, force: bool = False,
                 measure: bool = False, comm_thread: typing.Optional[bool] = None):
        """`force`: hook the gradients and issue every bucket's collective even in a group of ONE rank (the collective is
        then the identity; it exercises the RCCL launch path, its streams and `finish()` on a single GPU).
        `measure`: record HIP events around the wait in `finish()` — `exposed_comm_ms()` is the time the compute stream
        spent waiting for collectives that backward did not hide — and, per bucket, three events (`bucket_timings()`):
        `ready` on the stream whose announcement completed the bucket, `start` on the launch stream once every producer stream
        has been waited for, `end` on the launch stream after the collective (RCCL only: the launch stream is made to wait for
        the work, which does not block the host).  ready -> start is what the collective waits for OTHER streams' queued work
        (every bucket waits on all side streams, the encoder's included), start -> end its queueing behind earlier
        collectives plus its time on the wire.
        `comm_thread` (default: GLOWTTS_DP_COMM_THREAD, on): the collectives are issued by a launcher thread of this reducer, not by
        the thread whose gradient announcement completed the bucket.  That thread — autograd's backward thread, which queues the
        whole backward — only records one event per producer stream and hands (bucket, events) over; the launcher makes the
        collective's launch stream wait for those events and calls `all_reduce` (~50-80 us of host time per collective inside
        the process-group machinery, which releases the interpreter lock).  Buckets are handed over and launched in the same
        order on every rank (a FIFO; the backward's node order is the same everywhere), as the communicator requires."""
        flat = getattr(optimizer, "_optim", optimizer)
        if not hasattr(flat, "flat_g"):
            raise TypeError("FlowBlockReducer needs the flat-buffer optimizer (glow_tts_train.optimize.Adam)")
        self.flat = flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self.backend = backend
        self._use_avg = backend == "nccl"
        self._active = self.world > 1 or (bool(force) and dist.is_initialized())
        self._measure = bool(measure)
        self._exposed: typing.List[typing.Tuple[typing.Any, typing.Any]] = []
        self._bucket_events: typing.List[typing.Tuple[int, bool, typing.Any, typing.Any, typing.Any]] = []
        self._in_finish = False
        self.launched_in_backward = 0          # buckets whose collective was issued before finish() in the last step
        if comm_thread is None:
            comm_thread = os.environ.get("GLOWTTS_DP_COMM_THREAD", "1") != "0"
        self._thread_mode = bool(comm_thread) and self._active and flat.flat_g.is_cuda
        self._q: typing.Optional[queue.SimpleQueue] = None
        self._thr: typing.Optional[threading.Thread] = None
        self._thr_error: typing.Optional[BaseException] = None
        named = list(model.named_parameters())
        by_id = {id(p): (o, p.numel()) for p, o in zip(flat._params, flat.offsets)}
        self.buckets: typing.List[Bucket] = []
        self._bucket_of: typing.Dict[int, int] = {}
        cur_key, lo, hi, cnt = None, 0, 0, 0
        last_end = None
        for name, p in named:
            if id(p) not in by_id:
                raise RuntimeError(f"parameter {name} is not managed by the optimizer")
            o, n = by_id[id(p)]
            if last_end is not None and o < last_end:
                raise RuntimeError("model.named_parameters() order differs from the optimizer's flat layout")
            key = bucket_key(name)
            if key != cur_key:
                if cur_key is not None:
                    self.buckets.append(Bucket(cur_key, lo, hi, cnt))
                cur_key, lo, cnt = key, o, 0
            hi = o + n
            last_end = hi
            cnt += 1
            self._bucket_of[id(p)] = len(self.buckets)
        if cur_key is not None:
            self.buckets.append(Bucket(cur_key, lo, hi, cnt))
        self._pending = [b.n_params for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works: typing.List[typing.Any] = []
        self._hooks = []
        self._hook_of: typing.Dict[int, typing.Any] = {}
        self._announced: typing.Set[int] = set()     # parameters whose gradient an operator announced itself this step
        self._seen: typing.Set[int] = set()
        if self._active:
            for _, p in named:
                if p.requires_grad:
                    h = p.register_post_accumulate_grad_hook(self._on_hook)
                    self._hooks.append(h)
                    self._hook_of[id(p)] = h
            # gradients the conv operators write straight into .grad are announced by the operators themselves
            convops.add_grad_ready_listener(self._on_announce)
            if self._thread_mode:
                self._q = queue.SimpleQueue()
                self._thr = threading.Thread(target=self._comm_loop, args=(flat.flat_g.device,), name="glowtts-dp-comm", daemon=True)
                self._thr.start()

    # -- collectives ------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src: int = 0):
        """One broadcast of the whole flat parameter buffer (DDP's construction-time sync, __main__.py:269-271)."""
        if self._active:
            dist.broadcast(self.flat.flat_p, src=src, group=self.group)

    def _comm_loop(self, device):
        """The launcher thread: for every (bucket, producer events, ...) handed over, in order, make the launch stream wait for the
        events and issue the collective.  A `threading.Event` in the queue is a rendezvous (finish() waits for it); None ends the
        thread."""
        torch.cuda.set_device(device)
        comm = _hip.side_stream(device, "comm")
        while True:
            item = self._q.get()
            if item is None:
                return
            if isinstance(item, threading.Event):
                item.set()
                continue
            i, waits, ev, early = item
            try:
                b = self.buckets[i]
                view = self.flat.flat_g[b.lo:b.hi]
                # (issuing the collective from the stream that completed the bucket instead of a launch stream of its own —
                #  one stream fewer — was measured 0.3-0.5 ms per step SLOWER: its waits for the other producers' events stall
                #  that stream's own weight-gradient kernels; tools/dp_probe.py, round 5)
                with torch.cuda.stream(comm):
                    for e in waits:
                        comm.wait_event(e)
                    if ev is not None:
                        ev[1].record(comm)
                    if self._use_avg:
                        work = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                    else:
                        view.div_(self.world)
                        work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    if self.backend == "nccl":
                        work.wait()                  # the LAUNCH stream waits for the collective (no host block): `end` is its end
                    if ev is not None:
                        ev[2].record(comm)
                        self._bucket_events.append((i, early, *ev))
                        self._trim_events()
                self._works.append(work)
            except BaseException as exc:               # surfaced by finish() on the training thread
                self._thr_error = exc

    def _launch(self, i: int):
        b = self.buckets[i]
        view = self.flat.flat_g[b.lo:b.hi]
        if self._thread_mode:
            dev = view.device
            comm = _hip.side_stream(dev, "comm")
            cur = torch.cuda.current_stream(dev)
            ev = None
            if self._measure:
                ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
                ev[0].record(cur)
            waits = []
            for s in [cur] + [s for s in _hip.all_side_streams(dev) if s is not comm and s is not cur]:
                e = torch.cuda.Event()
                e.record(s)
                waits.append(e)
            self._q.put((i, waits, ev, not self._in_finish))
            self._launched[i] = True
            return
        if view.is_cuda:
            # A bucket's gradients come from several streams (dx chain, weight-gradient stream, encoder stream) and the
            # announcement that completes it arrives in the context of only ONE of them.  The collective is therefore issued
            # from a launch stream that first waits for all of them — the compute streams themselves never wait.
            comm = _hip.side_stream(view.device, "comm")
            ev = None
            if self._measure:
                ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
                ev[0].record(torch.cuda.current_stream(view.device))
            comm.wait_stream(torch.cuda.current_stream(view.device))
            for s in _hip.all_side_streams(view.device):
                if s is not comm:
                    comm.wait_stream(s)
            ctx = torch.cuda.stream(comm)
        else:
            ctx = contextlib.nullcontext()
            ev = None
        with ctx:
            if ev is not None:
                ev[1].record(comm)
            if self._use_avg:
                work = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                view.div_(self.world)
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if self.backend == "nccl" and view.is_cuda:
                work.wait()                          # the LAUNCH stream waits for the collective (no host block; the same with and
                                                     # without `measure`): `end` is its end
            if ev is not None:
                ev[2].record(comm)
                self._bucket_events.append((i, not self._in_finish, *ev))
                self._trim_events()
        self._works.append(work)
        self._launched[i] = True

    def _trim_events(self, keep_steps: int = 64):
        """`measure=True` keeps the timing events of the last `keep_steps` steps only: a long run that never reads
        `bucket_timings()` must not grow without bound."""
        cap = keep_steps * len(self.buckets)
        if len(self._bucket_events) > 2 * cap:
            del self._bucket_events[:-cap]
        if len(self._exposed) > 2 * keep_steps:
            del self._exposed[:-keep_steps]

    def _on_hook(self, p: torch.Tensor):
        # AccumulateGrad hooks also fire for parameters whose gradient an operator writes in place (it hands autograd None),
        # and they fire when that operator's backward RETURNS — before a deferred un-packing on another stream has run.
        # For those parameters (convops._mark_direct) only the operator's own announcement counts.
        if getattr(p, "_glowtts_direct", False):
            return
        self._on_grad(p)

    def _on_announce(self, params):
        """convops._notify: the gradients of `params` (a list) are complete on the current stream."""
        bucket_of, seen, pending = self._bucket_of, self._seen, self._pending
        ready = []
        for p in params:
            pid = id(p)
            self._announced.add(pid)
            i = bucket_of.get(pid)
            if i is None or pid in seen:
                continue
            seen.add(pid)
            pending[i] -= 1
            if pending[i] == 0 and not self._launched[i]:
                ready.append(i)
        for i in ready:
            self._launch(i)

    def _on_grad(self, p: torch.Tensor):
        if id(p) not in self._bucket_of or id(p) in self._seen:
            return
        self._seen.add(id(p))
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0 and not self._launched[i]:
            self._launch(i)

    def finish(self):
        """Call after backward(), before clipping: reduces buckets whose parameters got no gradient this step (their
        slice is the zeros left by zero_grad), then makes the current stream wait for every collective."""
        if self._active:
            self.launched_in_backward = sum(self._launched)
            self._in_finish = True
            for i in range(len(self.buckets)):
                if not self._launched[i]:
                    self._launch(i)
            self._in_finish = False
            if self._thread_mode:                        # every bucket handed over so far has been launched once this returns
                done = threading.Event()
                self._q.put(done)
                done.wait()
                if self._thr_error is not None:
                    exc, self._thr_error = self._thr_error, None
                    raise RuntimeError("FlowBlockReducer: the collective launcher thread failed") from exc
            cuda = self.flat.flat_g.is_cuda
            if cuda and self._measure:
                cur = torch.cuda.current_stream(self.flat.flat_g.device)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(cur)
            for w in self._works:
                w.wait()
            if cuda:                                     # (the non-AVG path divides on the launch stream before reducing)
                torch.cuda.current_stream(self.flat.flat_g.device).wait_stream(_hip.side_stream(self.flat.flat_g.device, "comm"))
                if self._measure:
                    t1.record(cur)
                    self._exposed.append((t0, t1))
            # Parameters whose gradient the operators announce themselves need no autograd hook: ~440 of 519 at the default
            # model, each a C++ -> Python call on the backward thread (~1.5 ms per step together).  A parameter that stops
            # being announced (another gradient mode) only loses its EARLY launch: finish() reduces whatever is left.
            for pid in self._announced:
                h = self._hook_of.pop(pid, None)
                if h is not None:
                    h.remove()
        self._announced.clear()
        self._works.clear()
        self._seen.clear()
        self._pending = [b.n_params for b in self.buckets]
        self._launched = [False] * len(self.buckets)

    def exposed_comm_ms(self, reset: bool = True) -> typing.List[float]:
        """Per step since the last call: milliseconds the compute stream waited in `finish()` for gradient collectives
        (`measure=True`; synchronises to read the events).  What backward hid does not show up here."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._exposed]
        if reset:
            self._exposed.clear()
        return out

    def bucket_timings(self, last_steps: typing.Optional[int] = None, reset: bool = True) -> typing.List[dict]:
        """Per bucket, means over the recorded steps (`measure=True`; synchronises): MB, the fraction of steps in which it was
        launched during backward, `ready_to_start_ms` and `start_to_end_ms` (None on a backend whose work cannot be waited for
        on a stream: gloo).  See __init__ for what the three events bracket."""
        torch.cuda.synchronize()
        ev = self._bucket_events
        if last_steps is not None:
            ev = ev[-last_steps * len(self.buckets):]
        rows = []
        for i, b in enumerate(self.buckets):
            mine = [e for e in ev if e[0] == i]
            if not mine:
                continue
            r2s = [e[2].elapsed_time(e[3]) for e in mine]
            s2e = [e[3].elapsed_time(e[4]) for e in mine] if self.backend == "nccl" else None
            rows.append({"key": b.key, "MB": round(4e-6 * (b.hi - b.lo), 2),
                         "launched_during_backward": round(sum(1 for e in mine if e[1]) / len(mine), 2),
                         "ready_to_start_ms": round(sum(r2s) / len(r2s), 3),
                         "start_to_end_ms": None if s2e is None else round(sum(s2e) / len(s2e), 3)})
        if reset:
            self._bucket_events.clear()
        return rows

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        convops.remove_grad_ready_listener(self._on_announce)
        if self._thr is not None:
            self._q.put(None)
            self._thr.join(timeout=10)
            self._thr, self._thread_mode = None, False
