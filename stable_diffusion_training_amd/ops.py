"""Host-side operators of the train_step path: thin torch.autograd.Function wrappers whose forward and
backward ONLY launch kernels of libsdtrain_hip.so through the C ABI (include/sdt.h) on the current HIP stream.
PyTorch supplies device memory, the stream and the autograd tape; none of its compute kernels are on the path.

Activations: bf16, NHWC / (rows, channels), channel counts multiples of 8.  Weight gradients are accumulated
straight into the owning ParamStore's flat fp32 gradient buffer (the all-reduce payload) as a side effect of
backward; `store.grad_ready(path)` lets the data-parallel reducer launch a bucket as soon as it is complete.
"""
import os

import torch
from torch.autograd import Function

from . import _lib
from ._lib import GATHER_DGRAD, GATHER_FPROP, GATHER_PLAIN, SdtAttnDesc, SdtConvGeom, call

BF16 = torch.bfloat16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def copy_cols(wide, parts, cols, rows, to_wide):
    """sdt_copy_cols_bf16: column segments of `wide` (rows, ld) <-> the contiguous-row matrices `parts` (None entries: zero-filled
    when gathering), up to 32 segments per launch."""
    ld_wide = wide.stride(-2) if wide.dim() > 1 else wide.shape[-1]
    col0 = 0
    for i in range(0, len(parts), 32):
        ps, cs = parts[i: i + 32], cols[i: i + 32]
        n = len(ps)
        ptrs = (_lib.ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ps])
        lds = (_lib.ctypes.c_int64 * n)(*[c if t is None else t.stride(-2) for t, c in zip(ps, cs)])
        cv = (_lib.ctypes.c_int * n)(*cs)
        call("sdt_copy_cols_bf16", wide.data_ptr() + 2 * col0, ld_wide, ptrs, lds, cv, n, rows, int(to_wide), _stream())
        col0 += sum(cs)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ready(store, *paths):
    """The gradients of `paths` have been enqueued: single-use check (ParamStore.note_written), then the data-parallel reducer's
    bucket bookkeeping (store.grad_ready)."""
    note = getattr(store, "note_written", None)
    cb = getattr(store, "grad_ready", None)
    for p in paths:
        if p is not None:
            if note is not None:
                note(p)
            if cb is not None:
                cb(p)


def _check(t, name="tensor"):
    if t.dtype != BF16 or not t.is_contiguous() or not t.is_cuda:
        raise _lib.SdtError(f"{name}: expected a contiguous bf16 device tensor, got {t.dtype} contiguous={t.is_contiguous()} {t.device}")
    return t


def _padded_bias(store, bpath, n):
    if bpath is None:
        return None
    b = store.p(bpath)
    if b.numel() == n:
        return b
    out = torch.zeros(n, dtype=torch.float32, device=b.device)
    out[: b.numel()] = b
    return out


# ----------------------------------------------------------------------------------------- GEMM helpers
class KernelTimer:
    """Optional HIP-event timing of one kernel family on the launch stream (bench.py roofline leg)."""

    def __init__(self):
        self.records = []  # (start_event, end_event, algorithmic_flops)

    @staticmethod
    def event_pair_overhead_ms(n=64):
        """What an empty (start, end) event pair reads on this stream: the marker packets themselves take ~us each, which
        inflates every short launch bracketed by them; summary() reports the raw sum and a sum with it subtracted."""
        pairs = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in pairs)
        return t[len(t) // 2]

    def summary(self):
        torch.cuda.synchronize()
        raw = [r[0].elapsed_time(r[1]) for r in self.records]
        cal = self.event_pair_overhead_ms()
        ms = sum(max(t - cal, 0.0) for t in raw)
        return dict(launches=len(self.records), ms=ms, raw_ms=sum(raw), event_overhead_us=1e3 * cal,
                    flops=float(sum(r[2] for r in self.records)))

    def by_shape(self):
        torch.cuda.synchronize()
        agg = {}
        for r in self.records:
            a = agg.setdefault(r[3], [0, 0.0, 0.0])
            a[0] += 1
            a[1] += r[0].elapsed_time(r[1])
            a[2] += r[2]
        rows = [(k, n, ms, fl / (ms * 1e-3) / 1e12) for k, (n, ms, fl) in agg.items()]
        return sorted(rows, key=lambda r: -r[2])


GEMM_NT_TIMER = None  # set to a KernelTimer to record every sdt_gemm_nt_bf16 launch
GEMM_TN_TIMER = None


def gemm_nt(A, Bt, out, M, N, Kc, taps, lda, ldb, b_tap_stride, *, bias=None, rowbias=None, residual=None,
            rows_per_batch=0, mode=GATHER_PLAIN, geom=None, gn_stats=None, gn_groups=0, b_kmajor=False, b_nseg=0, b_seg_stride=0):
    """b_kmajor: the second operand is B[taps][Kc][ldb] (the Flax kernel layout) instead of Bt[N][ldb] (include/sdt.h).
    rowbias may be a column slice (row pitch = its stride) of a wider matrix."""
    if GEMM_NT_TIMER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _gemm_nt(A, Bt, out, M, N, Kc, taps, lda, ldb, b_tap_stride, bias, rowbias, residual, rows_per_batch, mode, geom,
                 gn_stats, gn_groups, b_kmajor, b_nseg, b_seg_stride)
        e1.record()
        GEMM_NT_TIMER.records.append((e0, e1, 2.0 * M * N * Kc * taps, (M, N, Kc, taps, mode)))
        return
    _gemm_nt(A, Bt, out, M, N, Kc, taps, lda, ldb, b_tap_stride, bias, rowbias, residual, rows_per_batch, mode, geom,
             gn_stats, gn_groups, b_kmajor, b_nseg, b_seg_stride)


_WS_CACHE = {}
_SPLITK_WS = {}


def _splitk_workspace(need, device):
    """Persistent split-K scratch: arrival counters (zeroed once; every sdt_gemm_nt_bf16 launch leaves them zero again,
    include/sdt.h) followed by the partial-sum slabs, so all GEMMs of a stream share it.  (One stream only: two concurrent
    split-K GEMMs must not share a workspace.)"""
    ws = _SPLITK_WS.get(device)
    if ws is None or ws.numel() < need:
        ws = _SPLITK_WS[device] = torch.zeros(max(need, 32 << 20), dtype=torch.uint8, device=device)
    return ws


def _gemm_nt(A, Bt, out, M, N, Kc, taps, lda, ldb, b_tap_stride, bias, rowbias, residual, rows_per_batch, mode, geom,
             gn_stats=None, gn_groups=0, b_kmajor=False, b_nseg=0, b_seg_stride=0):
    key = (M, N, Kc, taps)
    need = _WS_CACHE.get(key)
    if need is None:
        need = _WS_CACHE[key] = _lib.load().sdt_gemm_nt_workspace_bytes(M, N, Kc, taps)
    ws = _splitk_workspace(need, out.device) if need else None
    call("sdt_gemm_nt_bf16", A.data_ptr(), Bt.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(rowbias), _ptr(residual), M, N,
         Kc, taps, lda, ldb, b_tap_stride, N, N if residual is not None else 0, rows_per_batch, mode,
         None if geom is None else _lib.ctypes.addressof(geom), _ptr(ws), ws.numel() if ws is not None else 0, _ptr(gn_stats),
         gn_groups, int(b_kmajor), b_nseg, b_seg_stride, 0 if rowbias is None else rowbias.stride(0), _stream())


# GroupNorm statistics produced by the GEMM / convolution that writes the GroupNorm's input (include/sdt.h gn_stats): the
# epilogues WRITE partial rows (B, nparts, G, 2) - single writer per slot, no atomics, no zero fill - and group_norm(stats=) adds
# them up in row order.  _ARENA: a per-step fp32 scratch cleared by ONE memset (the per-image column sums of the row-bias
# gradients accumulate into slices of it).
_GN_ARENA = {}      # device -> [tensor, next offset, active]
_GN_PARTS = {}


def gn_arena_begin(device, nbytes=1 << 20):
    a = _GN_ARENA.get(device)
    if a is None:
        a = _GN_ARENA[device] = [torch.zeros(nbytes // 4, dtype=torch.float32, device=device), 0, True]
    else:
        a[0].zero_()
    a[1], a[2] = 0, True


def gn_arena_end(device):
    a = _GN_ARENA.get(device)
    if a is not None:
        a[2] = False


def _arena_zeros(n, device):
    """n zero fp32 elements: a slice of the per-step arena (one memset per step) while a step is open, else a fresh tensor."""
    a = _GN_ARENA.get(device)
    if a is not None and a[2] and a[1] + n <= a[0].numel():
        out = a[0][a[1]: a[1] + n]
        a[1] += (n + 3) // 4 * 4
        return out
    return torch.zeros(n, dtype=torch.float32, device=device)


def _gn_parts(M, N, Kc, taps, rows_per_batch, groups, mode, geom):
    """Partial statistics rows per image this contraction's epilogue can write (0: it cannot, the norm runs its own pass)."""
    key = (M, N, Kc, taps, rows_per_batch, groups, mode, None if geom is None else bytes(geom))
    r = _GN_PARTS.get(key)
    if r is None:
        r = _GN_PARTS[key] = int(_lib.load().sdt_gemm_nt_gn_parts(M, N, Kc, taps, rows_per_batch, groups, mode,
                                                                   None if geom is None else _lib.ctypes.addressof(geom)))
    return r


def gemm_tn(A, dY, dW, M, K1, N, K1v, Nv, taps, lda, ldb, *, mode=GATHER_PLAIN, geom=None, dbias=None, n_seg=0, seg_stride=0, sq=None):
    """sq: device address of zeroed sdt_wgrad_sq_slots(K1, N, taps) doubles for the squared-norm partials (include/sdt.h), or None."""
    if GEMM_TN_TIMER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _gemm_tn(A, dY, dW, M, K1, N, K1v, Nv, taps, lda, ldb, mode, geom, dbias, n_seg, seg_stride, sq)
        e1.record()
        GEMM_TN_TIMER.records.append((e0, e1, 2.0 * M * K1 * N * taps, (M, K1, N, taps, mode)))
        return
    _gemm_tn(A, dY, dW, M, K1, N, K1v, Nv, taps, lda, ldb, mode, geom, dbias, n_seg, seg_stride, sq)


# ---- squared gradient norm gathered by the weight-gradient kernels (include/sdt.h sq_slots) -----------------------------------
# Between sq_begin(store) and sq_end(store) (train_step's backward, one process: without an exchange the norm clip_by_global_norm needs
# is that of the gradients as they are written) every weight-gradient launch of a QUANTISED kernel leaf gets a zeroed run of slots
# for its per-wave sums of squares; sq_end returns (slots, used) when every quantised leaf was covered exactly once - the optimizer
# then adds the slots up instead of reading 4 B per parameter back - and None otherwise (the ordinary pass runs).
def sq_begin(store):
    st = getattr(store, "_sq_state", None)  # (kept on the store: its slots live and die with it)
    if st is None:
        want = sum(lf.numel for lf in store.leaves.values() if lf.quantised)
        if not want:
            return
        n = store.quant_total // 512 + (1 << 16)  # 32 x 32 blocks over whole 128-tiles: ~1.3 x numel / 1024, with room for merged launches
        st = store._sq_state = dict(buf=torch.zeros(n, dtype=torch.float64, device=store.device), want=want)
    else:
        st["buf"].zero_()
    st.update(next=0, cov=0, on=True)


def sq_end(store):
    st = getattr(store, "_sq_state", None)
    if st is None or not st.get("on"):
        return None
    st["on"] = False
    return (st["buf"], st["next"]) if st["cov"] == st["want"] else None


def _sq_slots(store, paths, K1, N, taps):
    st = getattr(store, "_sq_state", None)
    if st is None or not st.get("on"):
        return None
    leaves = [store.leaves[p] for p in paths if p is not None and p.endswith("/kernel")]
    if not leaves or not all(lf.quantised for lf in leaves):
        return None
    n = int(_lib.load().sdt_wgrad_sq_slots(K1, N, taps))
    if st["next"] + n > st["buf"].numel():
        st["cov"] = -1 << 60  # out of room: this step takes the ordinary pass (never seen with the sizing of sq_begin)
        return None
    ptr = st["buf"].data_ptr() + 8 * st["next"]
    st["next"] += n
    st["cov"] += sum(lf.numel for lf in leaves)
    return ptr


_TN_WS_CACHE = {}
_TN_WS = {}


def _tn_workspace(need, device):
    """Persistent scratch of the weight-gradient kernels: arrival counters (zero between launches, include/sdt.h) followed by
    the per-split partial tiles; zeroed once, shared by every wgrad launch of a stream."""
    ws = _TN_WS.get(device)
    if ws is None or ws.numel() < need:
        ws = _TN_WS[device] = torch.zeros(max(need, 64 << 20), dtype=torch.uint8, device=device)
    return ws


# ---- Dense weight gradients held back and issued together (include/sdt.h sdt_gemm_tn_wgrad_group) ---------------------------
# Inside `with wgrad_grouping():` (train_step's backward) the weight gradient of a Dense layer / 1x1 convolution is queued instead
# of launched: nothing on the input-gradient chain waits for it, each alone is a 10 - 25 us launch of 25 - 100 tiles, and a dozen
# of them as one launch fill the chip.  The queue is flushed when it holds WGRAD_GROUP_LIMIT jobs and when the context closes
# (before the gradient exchange / optimizer); store.grad_ready fires at the flush.  Outside the context nothing is deferred.
WGRAD_GROUP_LIMIT = int(os.environ.get("SDT_WGRAD_GROUP", "32"))  # 0: never defer (developer A/B)
CONV_GROUP_LIMIT = int(os.environ.get("SDT_CONV_WGRAD_GROUP", "8"))
WGRAD_DUMP = bool(os.environ.get("SDT_WGRAD_DUMP"))
_WGRAD_QUEUE = None
_CONV_QUEUE = []  # deferred convolution weight gradients (sdt_conv_wgrad_group): (problem, tensors kept alive, store, paths, flops)
_NORM_QUEUE = []  # deferred LayerNorm parameter-gradient sums (sdt_norm_param_grads_group): (job, workspace kept alive, store, paths)


class wgrad_grouping:
    def __enter__(self):
        global _WGRAD_QUEUE
        self.prev = _WGRAD_QUEUE
        _WGRAD_QUEUE = [] if WGRAD_GROUP_LIMIT > 0 else None
        return self

    def __exit__(self, *exc):
        global _WGRAD_QUEUE
        try:
            if exc[0] is None:
                flush_wgrads()
        finally:
            _WGRAD_QUEUE = self.prev
            del _NORM_QUEUE[:]
            del _CONV_QUEUE[:]


def flush_norm_grads():
    """One launch for the dgamma / dbeta sums of every LayerNorm whose backward has run since the last flush."""
    if not _NORM_QUEUE:
        return
    step = _lib.load().sdt_norm_param_grads_group_max()
    for i in range(0, len(_NORM_QUEUE), step):
        jobs = _NORM_QUEUE[i: i + step]
        arr = (_lib.SdtNormGradJob * len(jobs))(*[j[0] for j in jobs])
        call("sdt_norm_param_grads_group", arr, len(jobs), _stream())
    for job, keep, store, paths in _NORM_QUEUE:
        _ready(store, *paths)
    del _NORM_QUEUE[:]


def flush_conv_wgrads():
    q = _CONV_QUEUE
    if not q:
        return
    lib = _lib.load()
    step = lib.sdt_gemm_tn_wgrad_group_max()
    for i in range(0, len(q), step):
        jobs = q[i: i + step]
        arr = (_lib.SdtConvWgradProblem * len(jobs))(*[j[0] for j in jobs])
        need = lib.sdt_conv_wgrad_group_workspace_bytes(arr, len(jobs))
        ws = _tn_workspace(need, jobs[0][1][0].device) if need else None
        if WGRAD_DUMP:  # developer: the launch's problems, replayed by tools/tn_group_micro.py
            print("WGRAD_GROUP conv " + ";".join(f"{j[0].geom.batch},{j[0].geom.out_h},{j[0].geom.out_w},{j[0].K1},{j[0].N},{j[0].geom.kh},{j[0].geom.stride}"
                                               for j in jobs), flush=True)
        e0 = e1 = None
        if GEMM_TN_TIMER is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call("sdt_conv_wgrad_group", arr, len(jobs), _ptr(ws), ws.numel() if ws is not None else 0, _stream())
        if e0 is not None:
            e1.record()
            GEMM_TN_TIMER.records.append((e0, e1, sum(j[4] for j in jobs), ("conv group", len(jobs))))
    for prob, keep, store, paths, fl in q:
        _ready(store, *paths)
    del q[:]


def wgrad_conv(x, dy, dW, geom, lf, taps, M_out, *, dbias=None, store=None, paths=()):
    """Weight gradient of a k x k convolution: queued while a wgrad_grouping context is open (3x3 / stride 1 ones then share launches),
    launched otherwise."""
    sq = _sq_slots(store, paths, lf.Rp, lf.Cp, taps)
    if _WGRAD_QUEUE is None:
        gemm_tn(x, dy, dW, M_out, lf.Rp, lf.Cp, lf.R, lf.C, taps, lf.Rp, lf.Cp, mode=GATHER_FPROP, geom=geom, dbias=dbias, sq=sq)
        _ready(store, *paths)
        return
    prob = _lib.SdtConvWgradProblem(x.data_ptr(), dy.data_ptr(), dW.data_ptr(), _ptr(dbias), geom, lf.Rp, lf.Cp, lf.R, lf.C, lf.Rp, lf.Cp, sq,
                                    int(dW.dtype == BF16))
    _CONV_QUEUE.append((prob, (x, dy), store, paths, 2.0 * M_out * lf.Rp * lf.Cp * taps))
    if len(_CONV_QUEUE) >= CONV_GROUP_LIMIT:
        flush_conv_wgrads()


def flush_wgrads():
    flush_norm_grads()
    flush_conv_wgrads()
    q = _WGRAD_QUEUE
    if not q:
        return
    lib = _lib.load()
    step = min(len(q), lib.sdt_gemm_tn_wgrad_group_max())
    for i in range(0, len(q), step):
        jobs = q[i: i + step]
        arr = (_lib.SdtTnProblem * len(jobs))(*[j[0] for j in jobs])
        need = lib.sdt_gemm_tn_wgrad_group_workspace_bytes(arr, len(jobs))
        ws = _tn_workspace(need, jobs[0][1][0].device) if need else None
        if WGRAD_DUMP:
            print("WGRAD_GROUP dense " + ";".join(f"{j[0].M},{j[0].K1},{j[0].N},{j[0].lda},{j[0].ldb},{j[0].n_seg}" for j in jobs), flush=True)
        e0 = e1 = None
        if GEMM_TN_TIMER is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call("sdt_gemm_tn_wgrad_group", arr, len(jobs), _ptr(ws), ws.numel() if ws is not None else 0, _stream())
        if e0 is not None:
            e1.record()
            GEMM_TN_TIMER.records.append((e0, e1, sum(2.0 * j[0].M * j[0].K1 * j[0].N for j in jobs), ("group", len(jobs))))
    for prob, keep, store, paths in q:
        _ready(store, *paths)
    del q[:]


def wgrad_dense(x, dy, dW, M, K1, N, K1v, Nv, lda, ldb, *, dbias=None, n_seg=0, seg_stride=0, store=None, paths=()):
    """dW[K1v][Nv] (+ dbias) of a Dense layer / 1x1 convolution: queued while a wgrad_grouping context is open, launched otherwise."""
    sq = _sq_slots(store, paths, K1, N, 1)
    if _WGRAD_QUEUE is None:
        gemm_tn(x, dy, dW, M, K1, N, K1v, Nv, 1, lda, ldb, dbias=dbias, n_seg=n_seg, seg_stride=seg_stride, sq=sq)
        _ready(store, *paths)
        return
    prob = _lib.SdtTnProblem(x.data_ptr(), dy.data_ptr(), dW.data_ptr(), _ptr(dbias), M, K1, N, K1v, Nv, lda, ldb,
                             n_seg if n_seg else Nv, n_seg, seg_stride, sq, int(dW.dtype == BF16))
    _WGRAD_QUEUE.append((prob, (x, dy), store, paths))  # (x, dy) stay alive until the grouped launch has been enqueued
    if len(_WGRAD_QUEUE) >= WGRAD_GROUP_LIMIT:
        flush_wgrads()


def _gemm_tn(A, dY, dW, M, K1, N, K1v, Nv, taps, lda, ldb, mode, geom, dbias=None, n_seg=0, seg_stride=0, sq=None):
    ldw = n_seg if n_seg else Nv
    gp = None if geom is None else _lib.ctypes.addressof(geom)
    key = (M, K1, N, taps, n_seg, mode, None if geom is None else bytes(geom))
    need = _TN_WS_CACHE.get(key)
    if need is None:
        need = _TN_WS_CACHE[key] = _lib.load().sdt_gemm_tn_workspace_bytes(M, K1, N, taps, n_seg, mode, gp)
    ws = _tn_workspace(need, dY.device) if need else None
    # dW: the store's gradient view of the leaf - float32, or bf16 for the kernel leaves of a store that keeps them so (ParamStore.grad16)
    call("sdt_gemm_tn_wgrad", A.data_ptr(), dY.data_ptr(), dW.data_ptr(), int(dW.dtype == BF16), _ptr(dbias), M, K1, N, K1v, Nv, taps, lda, ldb, ldw,
         K1v * Nv, n_seg, seg_stride, mode, gp, _ptr(ws), ws.numel() if ws is not None else 0, sq, _stream())


def reduce_workspace(need, device):
    """The stream's shared split workspace (64 KiB of self-resetting arrival counters + scratch, include/sdt.h): the ordered
    cross-workgroup reductions (column sums, loss, gradient norm) use it like the split GEMMs do."""
    return _splitk_workspace(max(int(need), 1), device)


def colsum(dy, db, M, N, ld):
    need = _lib.load().sdt_colsum_workspace_bytes(1, M, N)
    ws = reduce_workspace(need, dy.device)
    call("sdt_colsum_accumulate", dy.data_ptr(), db.data_ptr(), M, N, ld, ws.data_ptr(), ws.numel(), _stream())


# ----------------------------------------------------------------------------------------- Linear
class _Linear(Function):
    """flax nn.Dense: y = x @ kernel (+ bias) (+ residual), kernel [in,out] (bf16 compute, fp32 accumulate)."""

    @staticmethod
    def forward(ctx, x, residual, store, wpath, bpath, gn_groups):
        _check(x, "linear input")
        W, lf = store.wmat(wpath)
        K = x.shape[-1]
        if K != lf.Rp:
            raise _lib.SdtError(f"{wpath}: input width {K} != padded in-features {lf.Rp}")
        M = x.numel() // K
        y = torch.empty(*x.shape[:-1], lf.Cp, dtype=BF16, device=x.device)
        if residual is not None:
            _check(residual, "linear residual")
        stats, rpb = None, 0
        if gn_groups and x.dim() == 3:  # (B, HW, C): statistics per image for the GroupNorm that reads y
            rpb = x.shape[1]
            nparts = _gn_parts(M, lf.Cp, lf.Rp, 1, rpb, gn_groups, GATHER_PLAIN, None)
            if nparts:
                stats = torch.empty(x.shape[0], nparts, gn_groups, 2, dtype=torch.float32, device=x.device)
        gemm_nt(x, W, y, M, lf.Cp, lf.Rp, 1, lf.Rp, lf.Cp, 0, bias=_padded_bias(store, bpath, lf.Cp), residual=residual,
                rows_per_batch=rpb if stats is not None else 0, gn_stats=stats, gn_groups=gn_groups if stats is not None else 0,
                b_kmajor=True)  # y = x @ W: W [in,out] is the k-major operand as it stands
        ctx.save_for_backward(x)
        ctx.meta = (store, wpath, bpath, residual is not None)
        if gn_groups:
            ctx.set_materialize_grads(False)  # no zero-filled "gradient" for the statistics output
            if stats is not None:
                ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, dstats=None):
        if dy is None:
            return None, None, None, None, None, None
        (x,) = ctx.saved_tensors
        store, wpath, bpath, has_res = ctx.meta
        dy = dy.contiguous()
        W, lf = store.wmat(wpath)
        M = x.numel() // lf.Rp
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm_nt(dy, W, dx, M, lf.Rp, lf.Cp, 1, lf.Cp, lf.Cp, 0)
        if store.trainable:
            wgrad_dense(x, dy, store.g(wpath), M, lf.Rp, lf.Cp, lf.R, lf.C, lf.Rp, lf.Cp,
                        dbias=store.g(bpath) if bpath is not None else None, store=store, paths=(wpath, bpath))
        return dx, (dy if has_res else None), None, None, None, None


def linear(x, store, name, residual=None, gn_groups=0):
    """gn_groups > 0: returns (y, stats) where stats are the GroupNorm(gn_groups) statistics of y accumulated by the GEMM's
    epilogue (None when this shape cannot fuse them); pass them to group_norm(..., stats=)."""
    bpath = name + "/bias" if store.has(name + "/bias") else None
    return _Linear.apply(x, residual, store, name + "/kernel", bpath, gn_groups)


class _LinearMulti(Function):
    """n flax nn.Dense layers that share their input (attention to_q/to_k/to_v; to_k/to_v of the context) as ONE GEMM each
    for forward, input gradient and weight gradient: y = x @ [W_0 | W_1 | ...] (+ biases), output (..., n*N).
    Needs the leaves adjacent in the store's flat buffers (ParamStore.mergeable)."""

    @staticmethod
    def forward(ctx, x, store, wpaths, bpaths):
        _check(x, "linear_multi input")
        lfs = [store.leaves[w] for w in wpaths]
        lf, n = lfs[0], len(lfs)
        K, N = lf.Rp, lf.Cp
        if x.shape[-1] != K:
            raise _lib.SdtError(f"{wpaths[0]}: input width {x.shape[-1]} != in-features {K}")
        M = x.numel() // K
        Wn = store.w[lf.w_off: lf.w_off + n * K * N]                     # n x [K][N] back to back: column segment t = leaf t
        bias = None
        if bpaths is not None:
            b0 = store.leaves[bpaths[0]]
            bias = store.master[b0.offset: b0.offset + n * N]
        y = torch.empty(*x.shape[:-1], n * N, dtype=BF16, device=x.device)
        gemm_nt(x, Wn, y, M, n * N, K, 1, K, N, 0, bias=bias, b_kmajor=True, b_nseg=N, b_seg_stride=K * N)
        ctx.save_for_backward(x)
        ctx.meta = (store, wpaths, bpaths)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        store, wpaths, bpaths = ctx.meta
        dy = dy.contiguous()
        lfs = [store.leaves[w] for w in wpaths]
        lf, n = lfs[0], len(lfs)
        K, N = lf.Rp, lf.Cp
        M = x.numel() // K
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            W = store.w[lf.w_off: lf.w_off + n * K * N]                  # n x [K][N]: reduction segment t = leaf t
            gemm_nt(dy, W, dx, M, K, N, n, n * N, N, K * N)
        if store.trainable:
            g0 = store.grad_view(lf.offset, lf.offset + n * K * N)
            db = None
            if bpaths is not None:
                b0 = store.leaves[bpaths[0]]
                db = store.grad_view(b0.offset, b0.offset + n * N)
            wgrad_dense(x, dy, g0, M, K, n * N, K, n * N, K, n * N, dbias=db, n_seg=N, seg_stride=lfs[1].offset - lf.offset,
                        store=store, paths=tuple(wpaths) + tuple(bpaths or ()))
        return dx, None, None, None


def linear_multi(x, store, names):
    """Returns the concatenated outputs (..., n*N) of the Dense layers `names` applied to x, or None when their leaves are
    not laid out back to back (the caller then applies them one by one)."""
    wpaths = tuple(n + "/kernel" for n in names)
    has_b = [store.has(n + "/bias") for n in names]
    if any(has_b) and not all(has_b):
        return None
    bpaths = tuple(n + "/bias" for n in names) if all(has_b) else None
    if not store.mergeable(wpaths, bpaths):
        return None
    return _LinearMulti.apply(x, store, wpaths, bpaths)


# ----------------------------------------------------------------------------------------- Conv2d (NHWC)
class _Conv2d(Function):
    """flax nn.Conv on NHWC, HWIO kernel: implicit GEMM (im2col gathered inside the kernel)."""

    @staticmethod
    def forward(ctx, x, rowbias, residual, store, wpath, bpath, stride, pad, gn_groups):
        _check(x, "conv input")
        W, lf = store.wmat(wpath)
        B, H, Wd, C = x.shape
        kh, kw = store.leaves[wpath].shape[:2]
        if C != lf.Rp:
            raise _lib.SdtError(f"{wpath}: input channels {C} != padded in-channels {lf.Rp}")
        (pt, pb), (pl, pr) = pad
        OH = (H + pt + pb - kh) // stride + 1
        OW = (Wd + pl + pr - kw) // stride + 1
        geom = SdtConvGeom(B, H, Wd, OH, OW, kh, kw, stride, pt, pl)
        plain = kh == 1 and kw == 1 and stride == 1 and pt == 0 and pl == 0
        y = torch.empty(B, OH, OW, lf.Cp, dtype=BF16, device=x.device)
        M = B * OH * OW
        mode = GATHER_PLAIN if plain else GATHER_FPROP
        stats = None
        nparts = _gn_parts(M, lf.Cp, lf.Rp, kh * kw, OH * OW, gn_groups, mode, None if plain else geom) if gn_groups else 0
        if nparts:
            stats = torch.empty(B, nparts, gn_groups, 2, dtype=torch.float32, device=x.device)
        gemm_nt(x, W, y, M, lf.Cp, lf.Rp, kh * kw, lf.Rp, lf.Cp, lf.Rp * lf.Cp, bias=_padded_bias(store, bpath, lf.Cp),
                rowbias=rowbias, residual=residual, rows_per_batch=OH * OW, mode=mode, geom=None if plain else geom,
                gn_stats=stats, gn_groups=gn_groups if stats is not None else 0, b_kmajor=True)  # HWIO kernel: [tap][in][out]
        ctx.save_for_backward(x)
        ctx.meta = (store, wpath, bpath, geom, plain, rowbias is not None, residual is not None)
        if gn_groups:
            ctx.set_materialize_grads(False)  # no zero-filled "gradient" for the statistics output
            if stats is not None:
                ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, dstats=None):
        if dy is None:
            return None, None, None, None, None, None, None, None, None
        (x,) = ctx.saved_tensors
        store, wpath, bpath, geom, plain, has_rb, has_res = ctx.meta
        dy = dy.contiguous()
        W, lf = store.wmat(wpath)
        B, H, Wd, C = x.shape
        taps = geom.kh * geom.kw
        M_out = B * geom.out_h * geom.out_w
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm_nt(dy, W, dx, B * H * Wd, lf.Rp, lf.Cp, taps, lf.Cp, lf.Cp, lf.Rp * lf.Cp,
                    mode=GATHER_PLAIN if plain else GATHER_DGRAD, geom=None if plain else geom)
        if store.trainable and plain:  # 1x1: a Dense layer over the pixels
            wgrad_dense(x, dy, store.g(wpath), M_out, lf.Rp, lf.Cp, lf.R, lf.C, lf.Rp, lf.Cp,
                        dbias=store.g(bpath) if bpath is not None else None, store=store, paths=(wpath, bpath))
        elif store.trainable:
            wgrad_conv(x, dy, store.g(wpath), geom, lf, taps, M_out, dbias=store.g(bpath) if bpath is not None else None,
                       store=store, paths=(wpath, bpath))
        drb = None
        if has_rb:  # gradient of the broadcast (B, Cout) row bias = per-image column sums (ordered partial sums, one launch)
            drb = torch.empty(B, lf.Cp, dtype=BF16, device=x.device)
            ws = reduce_workspace(_lib.load().sdt_colsum_workspace_bytes(B, geom.out_h * geom.out_w, lf.Cp), x.device)
            call("sdt_colsum_batched_bf16", dy.data_ptr(), drb.data_ptr(), B, geom.out_h * geom.out_w, lf.Cp, lf.Cp, ws.data_ptr(),
                 ws.numel(), _stream())
        return dx, drb, (dy if has_res else None), None, None, None, None, None, None


def conv2d(x, store, name, stride=1, pad=1, rowbias=None, residual=None, gn_groups=0):
    """gn_groups > 0: returns (y, stats): see linear()."""
    if isinstance(pad, int):
        pad = ((pad, pad), (pad, pad))
    bpath = name + "/bias" if store.has(name + "/bias") else None
    return _Conv2d.apply(x, rowbias, residual, store, name + "/kernel", bpath, stride, pad, gn_groups)


# ----------------------------------------------------------------------------------------- norms
class _GroupNorm(Function):
    @staticmethod
    def forward(ctx, x, store, name, groups, eps, silu, skip, parts):
        _check(x, "groupnorm input")
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        y = torch.empty_like(x)
        stats = torch.empty(B, groups, 2, dtype=torch.float32, device=x.device)
        nparts = 0
        # scratch: the partial rows of the norm's own statistics pass, or the first-level sums of many producer rows
        need = _lib.load().sdt_groupnorm_fwd_workspace_bytes(B, HW, C, groups)
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        if parts is not None:
            if parts.dim() != 4 or parts.shape[0] != B or tuple(parts.shape[2:]) != (groups, 2) or not parts.is_contiguous():
                raise _lib.SdtError(f"{name}: producer statistics have shape {tuple(parts.shape)}, expected {(B, 'nparts', groups, 2)}")
            nparts = parts.shape[1]
        call("sdt_groupnorm_fwd", x.data_ptr(), store.p(name + "/scale").data_ptr(), store.p(name + "/bias").data_ptr(),
             y.data_ptr(), stats.data_ptr(), B, HW, C, groups, eps, int(silu), _ptr(parts), nparts, _ptr(ws), need, _stream())
        ctx.save_for_backward(x, stats)
        ctx.meta = (store, name, groups, eps, silu)
        ctx.set_materialize_grads(False)
        return (y, x) if skip else y  # x handed back as the skip branch: its gradient is folded into this op's backward

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, stats = ctx.saved_tensors
        store, name, groups, eps, silu = ctx.meta
        dy = dy.contiguous()
        if dskip is not None:
            dskip = dskip.contiguous()
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        dx = torch.empty_like(x)
        dg = store.g(name + "/scale").data_ptr() if store.trainable else None
        db = store.g(name + "/bias").data_ptr() if store.trainable else None
        need = _lib.load().sdt_groupnorm_bwd_workspace_bytes(B, HW, C)
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        call("sdt_groupnorm_bwd", x.data_ptr(), dy.data_ptr(), stats.data_ptr(), store.p(name + "/scale").data_ptr(),
             store.p(name + "/bias").data_ptr(), dx.data_ptr(), dg, db, _ptr(dskip), B, HW, C, groups, eps,
             int(silu), ws.data_ptr(), need, _stream())
        if store.trainable:
            _ready(store, name + "/scale", name + "/bias")
        return dx, None, None, None, None, None, None, None


def group_norm(x, store, name, groups=32, eps=1e-5, silu=False, skip=False, stats=None):
    """skip=True returns (norm(x), x'): use x' for the branch that bypasses the norm (residual / shortcut); the gradient
    arriving on x' is then added inside the norm's backward kernel instead of by a separate add launch."""
    return _GroupNorm.apply(x, store, name, groups, eps, silu, skip, stats)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, store, name, eps, skip):
        _check(x, "layernorm input")
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mr = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        call("sdt_layernorm_fwd", x.data_ptr(), store.p(name + "/scale").data_ptr(), store.p(name + "/bias").data_ptr(),
             y.data_ptr(), mr.data_ptr(), M, C, eps, _stream())
        ctx.save_for_backward(x, mr)
        ctx.meta = (store, name)
        ctx.set_materialize_grads(False)
        return (y, x) if skip else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, mr = ctx.saved_tensors
        store, name = ctx.meta
        dy = dy.contiguous()
        if dskip is not None:
            dskip = dskip.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = store.g(name + "/scale").data_ptr() if store.trainable else None
        db = store.g(name + "/bias").data_ptr() if store.trainable else None
        need = _lib.load().sdt_layernorm_bwd_workspace_bytes(M, C) if store.trainable else 0  # partial rows of dgamma / dbeta
        ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
        defer = store.trainable and _WGRAD_QUEUE is not None  # inside train_step's backward: the sums of all norms in one launch
        call("sdt_layernorm_bwd", x.data_ptr(), dy.data_ptr(), store.p(name + "/scale").data_ptr(), mr.data_ptr(),
             dx.data_ptr(), dg, db, _ptr(dskip), M, C, int(defer), _ptr(ws), need, _stream())
        if defer:
            job = _lib.SdtNormGradJob(ws.data_ptr(), dg, db, int(_lib.load().sdt_layernorm_bwd_partial_rows(M, C)), C)
            _NORM_QUEUE.append((job, ws, store, (name + "/scale", name + "/bias")))
        elif store.trainable:
            _ready(store, name + "/scale", name + "/bias")
        return dx, None, None, None, None


def layer_norm(x, store, name, eps=1e-5, skip=False):
    """skip=True: see group_norm."""
    return _LayerNorm.apply(x, store, name, eps, skip)


class _Fanout(Function):
    """n aliases of x for n consumers; the n gradients are summed by ONE launch (fp32 accumulation) instead of the
    autograd engine's chain of n-1 binary adds."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g.contiguous() for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        out = torch.empty_like(gs[0])
        for i in range(0, len(gs), 31):  # 32 pointers per launch; later launches chain the running sum in
            chunk = ([out] if i else []) + gs[i:i + 31]
            arr = (_lib.ctypes.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
            call("sdt_sum_n_bf16", arr, len(chunk), out.data_ptr(), out.numel(), _stream())
        return out, None


class _ColSlices(Function):
    """n equal column slices of a (..., n*C) tensor as separate tensors (views: no copies); the n gradients are gathered back
    into one (..., n*C) tensor by n strided copies.  Lets ONE GEMM produce the outputs of n Dense layers whose consumers are
    different ops (the time-embedding projections of all ResBlocks of one width, the [k|v] projections of all cross-attention
    blocks of one width)."""

    @staticmethod
    def forward(ctx, y, n):
        ctx.set_materialize_grads(False)
        ctx.n = n
        C = y.shape[-1] // n
        return tuple(y[..., i * C: (i + 1) * C] for i in range(n))

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        g0 = next((g for g in grads if g is not None), None)
        if g0 is None:
            return None, None
        C = g0.shape[-1]
        rows = g0.numel() // C
        dy = torch.empty(*g0.shape[:-1], n * C, dtype=BF16, device=g0.device)
        copy_cols(dy.view(rows, n * C), [None if g is None else g.contiguous().view(rows, C) for g in grads], [C] * n, rows, True)
        return dy, None


def col_slices(y, n):
    return _ColSlices.apply(y, n)


def fanout(x, n):
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * max(n, 1)
    _check(x, "fanout input")
    return _Fanout.apply(x, n)


# ----------------------------------------------------------------------------------------- activations
class _Act(Function):
    @staticmethod
    def forward(ctx, x, act):
        _check(x, "activation input")
        y = torch.empty_like(x)
        call("sdt_act_fwd", x.data_ptr(), y.data_ptr(), x.numel(), act, _stream())
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        call("sdt_act_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), ctx.act, _stream())
        return dx, None


def silu(x):
    return _Act.apply(x, _lib.ACT_SILU)


def quick_gelu(x):
    return _Act.apply(x, _lib.ACT_QUICK_GELU)


def gelu_erf(x):
    return _Act.apply(x, _lib.ACT_GELU_ERF)


class _Geglu(Function):
    @staticmethod
    def forward(ctx, h):
        _check(h, "geglu input")
        F2 = h.shape[-1]
        M = h.numel() // F2
        out = torch.empty(*h.shape[:-1], F2 // 2, dtype=BF16, device=h.device)
        call("sdt_geglu_fwd", h.data_ptr(), out.data_ptr(), M, F2 // 2, _stream())
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        dout = dout.contiguous()
        F2 = h.shape[-1]
        dh = torch.empty_like(h)
        call("sdt_geglu_bwd", h.data_ptr(), dout.data_ptr(), dh.data_ptr(), h.numel() // F2, F2 // 2, _stream())
        return dh


def geglu(h):
    return _Geglu.apply(h)


class _FeedForwardGeglu(Function):
    """diffusers FlaxFeedForward: y = (a * gelu_tanh(g)) @ W2 + b2 (+ residual), [a | g] = x @ W1 + b1, with the GEGLU inside FF1's
    epilogue (include/sdt.h sdt_ff_geglu_fwd): one launch stores h = [a | g] (kept for the backward) and the gated product - the
    separate GEGLU launch and its re-read of h are gone (-0.15 ms per step).  The backward keeps the three separate kernels: the same
    fusion in the epilogue of FF2's input gradient (two tanh per element beside four resident waves) measured no gain."""

    @staticmethod
    def forward(ctx, x, residual, store, n1, n2):
        _check(x, "feed-forward input")
        W1, l1 = store.wmat(n1 + "/kernel")
        W2, l2 = store.wmat(n2 + "/kernel")
        K, F = l1.Rp, l1.Cp // 2
        M = x.numel() // K
        h = torch.empty(*x.shape[:-1], 2 * F, dtype=BF16, device=x.device)
        f = torch.empty(*x.shape[:-1], F, dtype=BF16, device=x.device)
        b1 = _ptr(_padded_bias(store, n1 + "/bias" if store.has(n1 + "/bias") else None, l1.Cp))
        if GEMM_NT_TIMER is not None:  # (bench.py's roofline leg: this launch belongs to the sdt_gemm_nt_bf16 family)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call("sdt_ff_geglu_fwd", x.data_ptr(), W1.data_ptr(), b1, h.data_ptr(), f.data_ptr(), M, F, K, _stream())
            e1.record()
            GEMM_NT_TIMER.records.append((e0, e1, 2.0 * M * 2 * F * K, (M, 2 * F, K, 1, "ff1+geglu")))
        else:
            call("sdt_ff_geglu_fwd", x.data_ptr(), W1.data_ptr(), b1, h.data_ptr(), f.data_ptr(), M, F, K, _stream())
        y = torch.empty(*x.shape[:-1], l2.Cp, dtype=BF16, device=x.device)
        if residual is not None:
            _check(residual, "feed-forward residual")
        gemm_nt(f, W2, y, M, l2.Cp, l2.Rp, 1, l2.Rp, l2.Cp, 0, bias=_padded_bias(store, n2 + "/bias" if store.has(n2 + "/bias") else None, l2.Cp),
                residual=residual, b_kmajor=True)
        ctx.save_for_backward(x, h, f)
        ctx.meta = (store, n1, n2, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, f = ctx.saved_tensors
        store, n1, n2, has_res = ctx.meta
        dy = dy.contiguous()
        W1, l1 = store.wmat(n1 + "/kernel")
        W2, l2 = store.wmat(n2 + "/kernel")
        K, F = l1.Rp, l1.Cp // 2
        M = x.numel() // K
        df = torch.empty_like(f)
        gemm_nt(dy, W2, df, M, l2.Rp, l2.Cp, 1, l2.Cp, l2.Cp, 0)
        dh = torch.empty_like(h)
        call("sdt_geglu_bwd", h.data_ptr(), df.data_ptr(), dh.data_ptr(), M, F, _stream())
        if store.trainable:
            b2 = n2 + "/bias" if store.has(n2 + "/bias") else None
            wgrad_dense(f, dy, store.g(n2 + "/kernel"), M, l2.Rp, l2.Cp, l2.R, l2.C, l2.Rp, l2.Cp,
                        dbias=store.g(b2) if b2 is not None else None, store=store, paths=(n2 + "/kernel", b2))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm_nt(dh, W1, dx, M, l1.Rp, l1.Cp, 1, l1.Cp, l1.Cp, 0)
        if store.trainable:
            b1 = n1 + "/bias" if store.has(n1 + "/bias") else None
            wgrad_dense(x, dh, store.g(n1 + "/kernel"), M, l1.Rp, l1.Cp, l1.R, l1.C, l1.Rp, l1.Cp,
                        dbias=store.g(b1) if b1 is not None else None, store=store, paths=(n1 + "/kernel", b1))
        return dx, (dy if has_res else None), None, None, None


def feed_forward_geglu(x, store, n1, n2, residual=None):
    """The transformer block's feed-forward (Dense 8C with GEGLU, Dense C) of x (..., C); GEGLU inside FF1's epilogue where the shape is
    served (sdt_ff_geglu_supported), the three separate ops otherwise."""
    l1, l2 = store.leaves[n1 + "/kernel"], store.leaves[n2 + "/kernel"]
    K, F = l1.Rp, l1.Cp // 2
    M = x.numel() // K
    if (l1.batch == 1 and l2.batch == 1 and l1.Cp == l1.C and l2.Rp == F and l2.Cp == K and x.shape[-1] == K
            and _lib.load().sdt_ff_geglu_supported(M, F, K)):
        return _FeedForwardGeglu.apply(x, residual, store, n1, n2)
    return linear(geglu(linear(x, store, n1)), store, n2, residual=residual)


# ----------------------------------------------------------------------------------------- data movement
class _Upsample2x(Function):
    @staticmethod
    def forward(ctx, x):
        _check(x, "upsample input")
        B, H, W, C = x.shape
        y = torch.empty(B, 2 * H, 2 * W, C, dtype=BF16, device=x.device)
        call("sdt_upsample2x_fwd", x.data_ptr(), y.data_ptr(), B, H, W, C, _stream())
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, C, dtype=BF16, device=dy.device)
        call("sdt_upsample2x_bwd", dy.data_ptr(), dx.data_ptr(), B, H, W, C, _stream())
        return dx


def upsample2x(x):
    return _Upsample2x.apply(x)


class _ConcatChannels(Function):
    @staticmethod
    def forward(ctx, a, b):
        _check(a, "concat a")
        _check(b, "concat b")
        Ca, Cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // Ca
        y = torch.empty(*a.shape[:-1], Ca + Cb, dtype=BF16, device=a.device)
        copy_cols(y.view(rows, Ca + Cb), [a.view(rows, Ca), b.view(rows, Cb)], [Ca, Cb], rows, True)
        ctx.meta = (a.shape, b.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        sa, sb = ctx.meta
        dy = dy.contiguous()
        Ca, Cb = sa[-1], sb[-1]
        rows = dy.numel() // (Ca + Cb)
        da = torch.empty(sa, dtype=BF16, device=dy.device)
        db = torch.empty(sb, dtype=BF16, device=dy.device)
        copy_cols(dy.view(rows, Ca + Cb), [da.view(rows, Ca), db.view(rows, Cb)], [Ca, Cb], rows, False)
        return da, db


def concat_channels(a, b):
    return _ConcatChannels.apply(a, b)


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        _check(a, "add a")
        _check(b, "add b")
        y = torch.empty_like(a)
        call("sdt_add_bf16", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


# ----------------------------------------------------------------------------------------- attention
class _Attention(Function):
    """softmax(q k^T * scale) v per (batch, head); q (B,Nq,H*D), k/v (B,Nk,H*D)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, causal, key_weight=None):
        for t, n in ((q, "q"), (k, "k"), (v, "v")):
            _check(t, n)
        B, Nq, C = q.shape
        Nk = k.shape[1]
        D = C // heads
        desc = SdtAttnDesc(B, heads, Nq, Nk, D, C, k.shape[2], v.shape[2], C, scale, int(causal), 0, 0, 0, 0, _kw_ptr(key_weight, Nk))
        ctx.key_weight = key_weight  # keeps the device array the descriptor points at alive
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, Nq, dtype=torch.float32, device=q.device)
        call("sdt_attention_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
             _lib.ctypes.addressof(desc), _stream())
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.desc = desc
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        need = _lib.load().sdt_attention_bwd_workspace_bytes(_lib.ctypes.addressof(ctx.desc))
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)
        call("sdt_attention_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
             dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), need, _lib.ctypes.addressof(ctx.desc), _stream())
        return dq, dk, dv, None, None, None, None


def _kw_ptr(key_weight, Nk):
    if key_weight is None:
        return None
    if key_weight.dtype != torch.float32 or not key_weight.is_contiguous() or key_weight.numel() != Nk:
        raise ValueError(f"key_weight must be a contiguous float32 array of {Nk} elements")
    return key_weight.data_ptr()


def attention(q, k, v, heads, scale, causal=False, key_weight=None):
    """key_weight: optional (Nk,) f32 device array w > 0, P = softmax(scale q.k + ln w) (SdtAttnDesc.key_weight)."""
    return _Attention.apply(q, k, v, heads, scale, causal, key_weight)


class _AttentionPacked(Function):
    """attention() on the packed projections of linear_multi: a = [q|k|v] (B,N,3C) for self-attention, or a = q (B,Nq,C)
    with b = [k|v] (B,Nk,2C); heads are column slices as before, the row strides are the packed widths, and the backward
    writes dq/dk/dv straight into the packed gradient tensors."""

    @staticmethod
    def forward(ctx, a, b, heads, scale, causal, key_weight=None):
        _check(a, "attention q")
        B, Nq = a.shape[0], a.shape[1]
        if b is None:
            C = a.shape[2] // 3
            q, k, v, Nk, ldq, ldkv = a.data_ptr(), a.data_ptr() + 2 * C, a.data_ptr() + 4 * C, Nq, 3 * C, 3 * C
        else:  # b may be a column slice of a wider projection (ops.col_slices): row pitch b.stride(1), rows of all images evenly spaced
            C = a.shape[2]
            Nk = b.shape[1]
            if (b.dtype != BF16 or not b.is_cuda or b.shape[2] != 2 * C or b.stride(2) != 1 or b.stride(1) % 8 or b.stride(1) < 2 * C
                    or (B > 1 and b.stride(0) != Nk * b.stride(1))):
                raise _lib.SdtError(f"attention kv: expected bf16 (B, Nk, 2C) rows at a constant pitch, got {b.dtype} {tuple(b.shape)} strides {b.stride()}")
            q, k, v, ldq, ldkv = a.data_ptr(), b.data_ptr(), b.data_ptr() + 2 * C, C, b.stride(1)
        D = C // heads
        ldg = ldkv if b is None else 2 * C  # the gradient of [k|v] is a packed tensor of its own
        desc = SdtAttnDesc(B, heads, Nq, Nk, D, ldq, ldkv, ldkv, C, scale, int(causal), ldq, ldg, ldg, 0, _kw_ptr(key_weight, Nk))
        ctx.key_weight = key_weight
        out = torch.empty(B, Nq, C, dtype=BF16, device=a.device)
        lse = torch.empty(B, heads, Nq, dtype=torch.float32, device=a.device)
        call("sdt_attention_fwd", q, k, v, out.data_ptr(), lse.data_ptr(), _lib.ctypes.addressof(desc), _stream())
        ctx.save_for_backward(a, b, out, lse)
        ctx.desc, ctx.C = desc, C
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, out, lse = ctx.saved_tensors
        C = ctx.C
        dout = dout.contiguous()
        da = torch.empty_like(a)
        db = None if b is None else torch.empty(b.shape, dtype=BF16, device=b.device)
        if b is None:
            q, k, v = a.data_ptr(), a.data_ptr() + 2 * C, a.data_ptr() + 4 * C
            dq, dk, dv = da.data_ptr(), da.data_ptr() + 2 * C, da.data_ptr() + 4 * C
        else:
            q, k, v = a.data_ptr(), b.data_ptr(), b.data_ptr() + 2 * C
            dq, dk, dv = da.data_ptr(), db.data_ptr(), db.data_ptr() + 2 * C
        need = _lib.load().sdt_attention_bwd_workspace_bytes(_lib.ctypes.addressof(ctx.desc))
        ws = torch.empty(need, dtype=torch.uint8, device=a.device)
        call("sdt_attention_bwd", q, k, v, out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dq, dk, dv, ws.data_ptr(), need,
             _lib.ctypes.addressof(ctx.desc), _stream())
        return da, db, None, None, None, None


def attention_packed(a, b, heads, scale, causal=False, key_weight=None):
    return _AttentionPacked.apply(a, b, heads, scale, causal, key_weight)


# ----------------------------------------------------------------------------------------- CLIP embeddings
class _Embedding(Function):
    @staticmethod
    def forward(ctx, anchor, ids, store, tok_path, pos_path, S):
        tok, pos = store.p(tok_path), store.p(pos_path)
        D = tok.shape[1]
        rows = ids.numel()
        out = torch.empty(*ids.shape, D, dtype=BF16, device=ids.device)
        call("sdt_embedding_fwd", ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), out.data_ptr(), rows, S, D, _stream())
        ctx.save_for_backward(ids)
        ctx.meta = (store, tok_path, pos_path, S, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        store, tok_path, pos_path, S, D = ctx.meta
        if store.trainable:
            dout = dout.contiguous()
            call("sdt_embedding_bwd", ids.data_ptr(), dout.data_ptr(), store.g(tok_path).data_ptr(),
                 store.g(pos_path).data_ptr(), ids.numel(), S, D, _stream())
            _ready(store, tok_path, pos_path)
        return None, None, None, None, None, None


def embedding(ids, store, tok_path, pos_path, S, anchor):
    """anchor: any tensor that requires grad, so autograd records the node (ids are integers)."""
    return _Embedding.apply(anchor, ids, store, tok_path, pos_path, S)
