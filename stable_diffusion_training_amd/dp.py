"""Data parallelism for train_step: one process per GPU, gradients averaged with a bucketed all-reduce over
RCCL (xGMI) that overlaps the remaining backward.

The reference gets the same semantics implicitly: a (N,1) mesh, batch sharded on "data_parallel", everything else
replicated (training_utils.py:35-37, 446-483, 835-932) so GSPMD inserts the gradient all-reduce for the global-batch
mean (:709).  Here the flat fp32 gradient buffer of each ParamStore is cut into contiguous buckets; a bucket is
launched on a side stream as soon as the backward has enqueued the weight-gradient kernels of all its leaves
(ParamStore leaves are laid out in forward order, so buckets complete back to front).  Clipping needs the REDUCED
gradient, so the optimizer sweep waits for the last bucket (reducer.finish()).

Works with any torch.distributed backend: "nccl" (= RCCL) on GPUs, "gloo" on CPU tensors for the world_size-2 tests.

Sharded optimizer (shard=True; SURVEY.md §8(e) "better-than-reference variant", same results): the gradients of the QUANTISED
segments - the Dense / conv kernels, > 96 % of the parameters, which the forward reads only through their bf16 mirrors - are
reduce-scattered instead of all-reduced: rank r receives the mean of slice r of every bucket, adds its slices' squared norm to
a double that one scalar all-reduce completes (clip_by_global_norm needs the global norm), runs clip + Lion-8bit + EMA on its
slices only, and the bf16 mirror slices are all-gathered for the next forward.  Wire bytes per GPU drop from 2(N-1)/N x 4 B to
(N-1)/N x (4 + 2) B per parameter, the optimizer sweep to 1/N; every xGMI link carries traffic in both phases.  The small
non-quantised segments (biases, norm parameters, embeddings: read from the fp32 master by the kernels) stay replicated: plain
all-reduce, identical sweep on every rank.  fp32 masters / momentum codes / EMA of a quantised slice are current only on its
owner: GradReducer.gather_state() - a collective EVERY rank calls - makes them whole before a checkpoint / export (exports of a
store that is not whole raise instead of starting a collective from one rank).  The all-gather runs on the communication
stream and the next step waits for it where it first reads trained weights (wait_gathered: behind its VAE encode).  The in-place
RCCL reduce_scatter_tensor / all_gather_into_tensor forms have only run on a one-rank group (no multi-GPU node was available to
the build); the two-rank tests go through gloo's all-reduce / all-gather stand-ins, the 8-way slicing is tested on virtual ranks,
and inplace_collectives_ok() checks the two forms at start-up before bench.py relies on them.

Captured steps (training_utils._GraphedStep): RCCL collectives are NOT captured (capturing them crashes on this stack, and
a graph that embeds a communicator is hard to reason about); instead the step becomes graph A (forward + backward, with an
event-record NODE where each bucket completes) | the eager exchange on the communication stream, each all-reduce
behind its bucket's event, overlapping the rest of graph A | graph B (clip + optimizer + EMA).  ExchangePlan is what the
capture records and the replay walks.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib


_SKIP_SELF = os.environ.get("SDT_DP_SKIP_SELF") == "1"  # developer probe (tools/dp_graph_probe.sh): see GradReducer._reduce


class _Done:
    def wait(self):
        return True


class ExchangePlan:
    """Recorded while a step is captured: the (event, gradient view) pairs in completion order and the loss hand-off."""

    def __init__(self, device):
        self.items = []
        self.events = os.environ.get("SDT_DP_EVENTS", "1") != "0"  # 0: exchange after graph A has finished (no overlap)
        self.loss_src = None
        self.loss_out = torch.zeros(1, dtype=torch.float32, device=device)
        self.split = None  # set by the capturer: ends graph A and begins graph B

    def new_event(self):
        ev = ctypes.c_void_p()
        _lib.call("sdt_event_create", ctypes.byref(ev))
        return ev

    def close(self):
        for ev, *_ in self.items:
            if ev is not None:
                _lib.call("sdt_event_destroy", ev)
        self.items = []


class GradReducer:
    def __init__(self, stores, process_group=None, bucket_bytes=96 << 20, overlap=True, force=False, shard=False, skip_self=False):
        """force: keep the exchange machinery on in a one-rank group (exercises the RCCL / stream logic on a one-GPU box).
        skip_self (with force, one rank): keep the launch structure - two graphs, event nodes, stream waits - but issue no collective:
        a one-rank all-reduce is the identity, and RCCL's one-rank kernels (a 3.9 GB copy at ~1.2 TB/s beside the backward) say
        nothing about a real exchange; what is left is the cost of the structure itself.
        shard: sharded optimizer (module docstring); needs a world size that divides 8."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.stores = list(stores)
        self.overlap = overlap
        self.shard = bool(shard) and self.active
        self.skip_self = bool(skip_self or _SKIP_SELF) and self.world == 1
        self.buckets = []  # dict(store, a, b, need, pending, launched, scatter)
        self._owner = {}
        for si, st in enumerate(self.stores):
            if self.shard:
                flagged = st.shard_buckets(self.world, bucket_bytes)
                ranges, scatter = [(a, b) for a, b, q, d in flagged], [q for a, b, q, d in flagged]
                owners = _leaves_per_range(st, ranges)
                st.sharded = True
                st._prep = None  # the per-step conversion table depends on it (ParamStore._build_prep)
            else:
                ranges, owners = st.bucket_ranges(bucket_bytes)
                scatter = [False] * len(ranges)
            for (a, b), leaves, sc in zip(ranges, owners, scatter):
                bi = len(self.buckets)
                self.buckets.append(dict(store=st, a=a, b=b, need=len(leaves), pending=len(leaves), launched=False, scatter=sc))
                for p in leaves:
                    self._owner.setdefault((si, p), []).append(bi)
            st.grad_ready = self._make_cb(si)
        self._handles = []
        self._seen = set()
        self.cuda = self.stores[0].grad.is_cuda if self.stores else False
        # RCCL averages in the collective itself; gloo only sums (the mean is applied after the wait)
        self.native_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
        # high priority: HIP keeps a separate hardware-queue pool per priority, so the exchange can never be mapped onto the
        # compute stream's queue (where it would sit behind the whole backward), and the dispatcher favours it
        self.comm_stream = torch.cuda.Stream(priority=-1) if (self.cuda and overlap) else None
        self.capture = None  # an ExchangePlan while a step is being captured
        # sharded optimizer on the device: the all-gather of the bf16 mirrors runs on the communication stream and the NEXT step waits
        # for it only where it first reads trained weights (wait_gathered, behind its VAE encode) - not at the end of this step
        self.defer_gather = self.shard and self.cuda and self.comm_stream is not None and os.environ.get("SDT_DP_DEFER_GATHER", "1") != "0"
        self._gather_event = None
        self._gather_pending = False
        self.timing = None   # set to [] to collect (first all-reduce issued, last finished, compute stream at the join) events

    def _make_cb(self, si):
        def cb(path):
            key = (si, path)
            if key in self._seen or not self.active:
                return
            self._seen.add(key)
            for bi in self._owner.get(key, ()):
                bk = self.buckets[bi]
                bk["pending"] -= 1
                if bk["pending"] == 0 and self.overlap:
                    self._launch(bk)
        return cb

    def begin_step(self):
        self._seen.clear()
        self._handles.clear()
        for bk in self.buckets:
            bk["pending"] = bk["need"]
            bk["launched"] = False

    def _launch(self, bk):
        if bk["launched"]:
            return
        bk["launched"] = True
        view = bk["store"].grad_view(bk["a"], bk["b"])
        if self.capture is not None:  # the bucket is complete HERE in the captured stream: mark it, exchange at replay
            ev = None
            if self.capture.events:
                ev = self.capture.new_event()
                _lib.call("sdt_event_record", ev, 1, torch.cuda.current_stream().cuda_stream)
            self.capture.items.append((ev, view, bk))
            return
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            if not self._handles:
                self._stamp_begin()
            with torch.cuda.stream(self.comm_stream):
                self._handles.append((self._reduce(bk, view), view))
        else:
            self._handles.append((self._reduce(bk, view), view))

    def _slice(self, bk):
        """Element range of this rank's slice of a scattered bucket."""
        n = (bk["b"] - bk["a"]) // self.world
        return bk["a"] + self.rank * n, bk["a"] + (self.rank + 1) * n

    def _reduce(self, bk, view):
        """One bucket's gradient exchange: mean over the ranks into every rank (all-reduce), or - scattered buckets of the sharded
        optimizer - into the owning rank's slice only (reduce-scatter, in place).  gloo has no reduce-scatter: it all-reduces,
        which leaves the same values in the owner's slice (CPU / one-GPU tests)."""
        if self.skip_self:  # forced one-rank group: the launch structure without the (identity) collective
            return _Done()
        if bk["scatter"] and self.native_avg:
            sa, sb = self._slice(bk)
            return dist.reduce_scatter_tensor(bk["store"].grad_view(sa, sb), view, op=self.op, group=self.group, async_op=True)
        return dist.all_reduce(view, op=self.op, group=self.group, async_op=True)

    def finish(self):
        """Launch whatever is left (leaves that got no gradient this step), wait, and turn sums into means."""
        if not self.active:
            return
        for bk in self.buckets:
            self._launch(bk)
        if self.capture is not None:
            self.capture.split()
            return
        inv = 1.0 / self.world
        if self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                for h, view in self._handles:
                    h.wait()
                    if not self.native_avg:
                        view.mul_(inv)
            self._stamp_end()
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for h, view in self._handles:
                h.wait()
                if not self.native_avg:
                    view.mul_(inv)
        self._handles.clear()
        self._shard_norms()

    # ---- sharded optimizer: norm of the scattered part, the pieces each rank sweeps, the mirror all-gather, state gather
    def _shard_norms(self):
        """sum of g^2 over this rank's slices of the scattered buckets (double, per store), completed by ONE all-reduce per store;
        the replicated segments are added by ParamStore.optimizer_step.  Runs on the compute stream behind the exchange."""
        if not self.shard:
            return
        for st in self.stores:
            st.sqnorm.zero_()
        for bk in self.buckets:
            if bk["scatter"]:
                st = bk["store"]
                sa, sb = self._slice(bk)
                if self.cuda:
                    st.sqnorm_accumulate(sa, sb)
                else:  # host tensors: the gloo plumbing tests (the optimizer kernels themselves need the device)
                    st.sqnorm += st.grad_view(sa, sb).double().square().sum()
        for st in self.stores:
            dist.all_reduce(st.sqnorm, op=dist.ReduceOp.SUM, group=self.group)

    def shard_pieces(self, store):
        """optimizer_step(shard=...) argument for one of this reducer's stores (None when the optimizer is replicated)."""
        if not self.shard:
            return None
        pieces = []
        for bk in self.buckets:
            if bk["store"] is store:
                quant = bk["scatter"]
                a, b = self._slice(bk) if quant else (bk["a"], bk["b"])
                decay = next(d for (q, d, sa, sb) in store.segments if sa <= bk["a"] < sb)
                pieces.append((a, b, quant, decay))
        return pieces, True

    def after_optimizer(self):
        """All-gather the bf16 mirror slices the owners have just written, so that every rank's next forward reads current weights."""
        if not self.shard:
            return
        if self.capture is not None:
            self.capture.post = True  # replay: GradReducer.run_post after the optimizer graph
            return
        self._gather_mirrors()

    def _gather_mirrors(self):
        if not self.defer_gather:
            self._gather_buffers(lambda st: [(st.w, 1)])
            return
        cs = self.comm_stream
        cs.wait_stream(torch.cuda.current_stream())  # behind the optimizer sweep that wrote this rank's slices
        with torch.cuda.stream(cs):
            self._gather_buffers(lambda st: [(st.w, 1)])
        if self._gather_event is None:
            ev = ctypes.c_void_p()
            _lib.call("sdt_event_create", ctypes.byref(ev))
            self._gather_event = ev
        _lib.call("sdt_event_record", self._gather_event, 0, cs.cuda_stream)
        self._gather_pending = True

    def wait_gathered(self):
        """The current stream waits for the last all-gather of the weight mirrors.  train_step calls it in front of its first read of
        trained weights (captured steps: an event-wait node of graph A, so the gather runs beside the VAE encode of the next step);
        anything else that reads ParamStore.w / padded weights on a stream right after a step must call it too (a device-wide
        synchronize covers it as well)."""
        if self._gather_event is None:
            if self.defer_gather and self.capture is not None:  # capturing before the first gather: the node must exist from the start
                ev = ctypes.c_void_p()
                _lib.call("sdt_event_create", ctypes.byref(ev))
                self._gather_event = ev
            else:
                return
        if self.capture is not None:
            _lib.call("sdt_stream_wait_event_external", torch.cuda.current_stream().cuda_stream, self._gather_event)
        elif self._gather_pending:
            _lib.call("sdt_stream_wait_event", torch.cuda.current_stream().cuda_stream, self._gather_event)
            self._gather_pending = False

    def _gather_buffers(self, what):
        """what(store) -> [(flat buffer, elements of the parameter range per buffer element)]: all-gather every scattered bucket's slices."""
        for bk in self.buckets:
            if not bk["scatter"] or bk["store"] not in self.stores:
                continue
            st = bk["store"]
            n = (bk["b"] - bk["a"]) // self.world
            for buf, per in what(st):
                if buf is None:
                    continue
                a, m = bk["a"] // per, n // per
                whole = buf[a: a + m * self.world]
                if self.native_avg:
                    dist.all_gather_into_tensor(whole, whole[self.rank * m: (self.rank + 1) * m], group=self.group)
                else:
                    mine = whole[self.rank * m: (self.rank + 1) * m].clone()
                    dist.all_gather([whole[r * m: (r + 1) * m] for r in range(self.world)], mine, group=self.group)

    def gather_state(self):
        """COLLECTIVE - call on every rank, like checkpoint.gather_rng_states, BEFORE any `if rank == 0:` save / export block:
        all-gathers fp32 master, EMA, momentum codes and scales of the scattered buckets so that every rank holds the whole
        state (ParamStore.state_whole).  A no-op for the replicated optimizer."""
        if not self.shard:
            return
        self.wait_gathered()
        self._gather_buffers(lambda st: [(st.master, 1), (st.ema, 1), (st.codes, 1), (st.inv_scale, st.block_size)])
        for st in self.stores:
            st.state_whole = True

    def run_post(self, plan):
        """Replay-time counterpart of after_optimizer for a captured step: call right after the optimizer graph was launched."""
        if getattr(plan, "post", False):
            self._gather_mirrors()

    # ---- optional timing of the exchange (bench.py): HIP events on the communication / compute streams, nothing when timing is None
    def _stamp_begin(self):
        if self.timing is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(self.comm_stream)  # behind the first bucket's readiness wait: the first all-reduce can start here
            self._t0 = e

    def _stamp_end(self):
        if self.timing is not None and getattr(self, "_t0", None) is not None:
            e1, ej = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e1.record(self.comm_stream)           # last bucket reduced
            ej.record(torch.cuda.current_stream())  # compute stream has finished the backward and now waits for e1
            self.timing.append((self._t0, e1, ej))
            self._t0 = None

    def exchange_times_ms(self):
        """(span, exposed) medians over the recorded steps: first all-reduce issued -> last one finished, and how long the
        compute stream then still had to wait (0 when the exchange hid behind the backward).  Call after a synchronize."""
        if not self.timing:
            return None
        span = sorted(a.elapsed_time(b) for a, b, _ in self.timing)
        exposed = sorted(max(j.elapsed_time(b), 0.0) for _, b, j in self.timing)
        return span[len(span) // 2], exposed[len(exposed) // 2]

    def mean_scalar(self, t):
        if not self.active:
            return t
        if self.capture is not None:
            self.capture.loss_src = t
            return self.capture.loss_out
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world

    def run_exchange(self, plan):
        """Replay-time counterpart of _launch/finish/mean_scalar for a captured step: call right after graph A was launched."""
        cs = self.comm_stream if self.comm_stream is not None else torch.cuda.current_stream()
        handles = []
        if not plan.events:
            cs.wait_stream(torch.cuda.current_stream())
        for ev, view, bk in plan.items:
            if ev is not None:
                _lib.call("sdt_stream_wait_event", cs.cuda_stream, ev)
            if not handles and self.comm_stream is not None:
                self._stamp_begin()
            with torch.cuda.stream(cs):
                handles.append((self._reduce(bk, view), view))
        inv = 1.0 / self.world
        with torch.cuda.stream(cs):
            for h, view in handles:
                h.wait()
                if not self.native_avg:
                    view.mul_(inv)
        if self.comm_stream is not None:
            self._stamp_end()
        torch.cuda.current_stream().wait_stream(cs)
        self._shard_norms()
        if plan.loss_src is not None:
            plan.loss_out.copy_(plan.loss_src)
            dist.all_reduce(plan.loss_out, op=dist.ReduceOp.SUM, group=self.group)
            plan.loss_out.mul_(inv)


def _leaves_per_range(store, ranges):
    """For each [a, b) element range: the leaves whose gradients it holds (a leaf that straddles ranges is listed in each)."""
    owners = [[] for _ in ranges]
    starts = [a for a, _ in ranges]
    import bisect
    for p, lf in store.leaves.items():
        lo, hi = lf.offset, lf.offset + max(lf.numel, 1) - 1
        i = max(bisect.bisect_right(starts, lo) - 1, 0)
        while i < len(ranges) and ranges[i][0] <= hi:
            if ranges[i][1] > lo:
                owners[i].append(p)
            i += 1
    return owners


def shardable_world(world):
    """The sharded optimizer cuts every bucket into `world` block-aligned slices: possible when world x 256 divides the segment
    alignment of ParamStore (world sizes 1, 2, 4, 8); any other world size runs the all-reduce + replicated sweep."""
    from .params import SEG_ALIGN
    return world >= 1 and SEG_ALIGN % (world * 256) == 0


def inplace_collectives_ok(device, group=None, _fail_on_rank=None):
    """Self-test of the two in-place collective forms the sharded optimizer relies on (reduce_scatter_tensor into the caller's own
    slice of the input, all_gather_into_tensor from the caller's own slice of the output), on 8 KiB per rank with known values.
    Collective: every rank calls it; every rank gets the same answer (a failure on any rank turns it off everywhere).

    Every rank issues the SAME sequence of collectives whatever happens locally: reduce-scatter, all-gather, all-reduce(MIN) of the
    verdict.  A rank whose in-place call raises (a runtime that rejects the aliasing) issues the same collective again out of
    place on scratch buffers - collectives pair up by kind and size, not by buffer - so its peers, which may already be inside
    that collective, are never left waiting for a call that does not come.  (_fail_on_rank: test hook, raises in front of both
    in-place calls on that rank.)"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nccl = dist.get_backend(group) == "nccl"
    n = 2048
    ok = True

    def matched(inplace, scratch, what):
        nonlocal ok
        try:
            if _fail_on_rank is not None and rank == _fail_on_rank:
                raise RuntimeError("injected failure")
            inplace()
            return True
        except Exception as e:  # this rank refuses the in-place form: keep the collective sequence whole, report, fall back everywhere
            print(f"[sdt] in-place {what} self-test failed on rank {rank}: {type(e).__name__}: {e}", flush=True)
            ok = False
            try:
                scratch()
            except Exception as e2:  # pragma: no cover - nothing left to keep the ranks paired with
                print(f"[sdt] out-of-place {what} failed as well on rank {rank}: {type(e2).__name__}: {e2}", flush=True)
            return False

    pattern = torch.arange(world * n, device=device, dtype=torch.float32) % 97
    x = pattern * float(rank + 1)
    want = pattern * (world + 1) / 2.0  # mean over ranks of (rank + 1)
    if nccl:
        if matched(lambda: dist.reduce_scatter_tensor(x[rank * n: (rank + 1) * n], x, op=dist.ReduceOp.AVG, group=group),
                   lambda: dist.reduce_scatter_tensor(torch.empty(n, device=device), x.clone(), op=dist.ReduceOp.AVG, group=group), "reduce-scatter"):
            ok = ok and bool(torch.allclose(x[rank * n: (rank + 1) * n], want[rank * n: (rank + 1) * n], rtol=1e-6, atol=0))
    else:  # gloo stand-in (CPU / one-GPU rehearsals): all-reduce leaves the same values in the owner's slice
        if matched(lambda: dist.all_reduce(x, group=group), lambda: dist.all_reduce(x.clone(), group=group), "all-reduce"):
            ok = ok and bool(torch.allclose(x[rank * n: (rank + 1) * n] / world, want[rank * n: (rank + 1) * n], rtol=1e-6, atol=0))
    y = torch.full((world * n,), -1.0, device=device, dtype=torch.bfloat16)
    y[rank * n: (rank + 1) * n] = float(rank)
    mine = y[rank * n: (rank + 1) * n]
    if nccl:
        done = matched(lambda: dist.all_gather_into_tensor(y, mine, group=group),
                       lambda: dist.all_gather_into_tensor(torch.empty_like(y), mine.clone(), group=group), "all-gather")
    else:
        done = matched(lambda: dist.all_gather([y[r * n: (r + 1) * n] for r in range(world)], mine.clone(), group=group),
                       lambda: dist.all_gather([torch.empty(n, device=device, dtype=torch.bfloat16) for _ in range(world)], mine.clone(), group=group),
                       "all-gather")
    if done:
        ok = ok and bool(torch.equal(y.view(world, n)[:, 0].float(), torch.arange(world, device=device, dtype=torch.float32)))
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() == 1.0)


def rccl_group_options():
    """pg_options for init_process_group("nccl", ...): RCCL's own streams in the high-priority queue pool (see comm_stream)."""
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    return opts
