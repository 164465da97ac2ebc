"""Data parallelism for train_step: one process per GPU, gradients averaged with a bucketed all-reduce over
RCCL (xGMI) that overlaps the remaining backward.

The reference gets the same semantics implicitly: a (N,1) mesh, batch sharded on "data_parallel", everything else
replicated (training_utils.py:35-37, 446-483, 835-932) so GSPMD inserts the gradient all-reduce for the global-batch
mean (:709).  Here the flat fp32 gradient buffer of each ParamStore is cut into contiguous buckets; a bucket is
launched on a side stream as soon as the backward has enqueued the weight-gradient kernels of all its leaves
(ParamStore leaves are laid out in forward order, so buckets complete back to front).  Clipping needs the REDUCED
gradient, so the optimizer sweep waits for the last bucket (reducer.finish()).

Works with any torch.distributed backend: "nccl" (= RCCL) on GPUs, "gloo" on CPU tensors for the world_size-2 tests.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, stores, process_group=None, bucket_bytes=96 << 20, overlap=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.stores = list(stores)
        self.overlap = overlap
        self.buckets = []  # dict(store, a, b, need, pending, launched)
        self._owner = {}
        for si, st in enumerate(self.stores):
            ranges, owners = st.bucket_ranges(bucket_bytes)
            for (a, b), leaves in zip(ranges, owners):
                bi = len(self.buckets)
                self.buckets.append(dict(store=st, a=a, b=b, need=len(leaves), pending=len(leaves), launched=False))
                for p in leaves:
                    self._owner.setdefault((si, p), []).append(bi)
            st.grad_ready = self._make_cb(si)
        self._handles = []
        self._seen = set()
        self.cuda = self.stores[0].grad.is_cuda if self.stores else False
        # RCCL averages in the collective itself; gloo only sums (the mean is applied after the wait)
        self.native_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
        self.comm_stream = torch.cuda.Stream() if (self.cuda and overlap) else None

    def _make_cb(self, si):
        def cb(path):
            key = (si, path)
            if key in self._seen or self.world == 1:
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
        view = bk["store"].grad[bk["a"]: bk["b"]]
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self._handles.append((dist.all_reduce(view, op=self.op, group=self.group, async_op=True), view))
        else:
            self._handles.append((dist.all_reduce(view, op=self.op, group=self.group, async_op=True), view))

    def finish(self):
        """Launch whatever is left (leaves that got no gradient this step), wait, and turn sums into means."""
        if self.world == 1:
            return
        for bk in self.buckets:
            self._launch(bk)
        inv = 1.0 / self.world
        if self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                for h, view in self._handles:
                    h.wait()
                    if not self.native_avg:
                        view.mul_(inv)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for h, view in self._handles:
                h.wait()
                if not self.native_avg:
                    view.mul_(inv)
        self._handles.clear()

    def mean_scalar(self, t):
        if self.world == 1:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world
