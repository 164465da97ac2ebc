"""Flat, HBM-resident parameter / gradient / optimizer-state storage for one trained model.

Mirrors what the reference keeps in a flax TrainState + optax state (training_utils.py:383-387, 420-425;
lion_quant.py:12-17) but laid out for the MI355X: ONE contiguous fp32 master buffer, ONE fp32 gradient buffer
(the RCCL all-reduce payload, bucketed by contiguous ranges), int8 codes + fp32 inverse scales for the
quantised leaves, fp32 momentum for the rest, optional fp32 EMA, and ONE bf16 compute copy W in the Flax layout itself that
every GEMM reads (forward as a k-major operand, input gradient as a row-major one): a mirror of the master, element for
element, written by the optimizer sweep.  Leaves are grouped into four contiguous segments by
(quantised?, weight-decayed?) so the fused optimizer sweep is at most four launches per model.

Leaf naming / layouts are the diffusers-Flax ones: conv kernel HWIO, Dense kernel [in,out];
`create_mask` keeps the reference's exact-path-component semantics (training_utils.py:116-131).
"""
import math
import os
from dataclasses import dataclass

import torch

from . import _lib, lion_codec

_THRESHOLDS = {}  # device -> float32[128] decision thresholds of the 8-bit Lion codec (lion_codec.quantization_thresholds)


def lion_thresholds(device):
    device = torch.device(device)
    t = _THRESHOLDS.get(device)
    if t is None:
        t = _THRESHOLDS[device] = torch.from_numpy(lion_codec.quantization_thresholds().copy()).to(device)
    return t


def create_mask(paths, excluded):
    """training_utils.py:116-131: True iff no path component equals an excluded pattern."""
    out = {}
    for k in paths:
        comps = tuple(k.split("/"))
        out[k] = not any(e in comps for e in excluded)
    return out


def _ceil(a, b):
    return (a + b - 1) // b * b


# Segment ends (and the bucket boundaries of the sharded optimizer) are multiples of this many elements, so that any bucket cut
# into 1, 2, 4 or 8 equal slices gives slices that are whole quantisation blocks (block sizes up to 256) and 64-element aligned
SEG_ALIGN = 8 * 256


@dataclass
class Leaf:
    path: str
    shape: tuple
    numel: int
    offset: int  # element offset into master / grad
    quantised: bool
    decayed: bool
    # bf16 compute copies (matrix leaves only)
    batch: int = 0
    R: int = 0
    C: int = 0
    Rp: int = 0
    Cp: int = 0
    w_off: int = -1


class EmaView:
    """The EMA copy of a store's parameters: what the reference threads through train_step as `unet_ema_params` /
    `text_encoder_ema_params` (training_utils.py:735-746) and hands to save_model for the -EMA checkpoint (training.py:281-296).
    The buffer itself lives in the store (the optimizer kernel updates it in the same sweep)."""

    def __init__(self, store):
        self.store = store

    @property
    def ema(self):
        return self.store.ema

    def export(self):
        return self.store.export("ema")


class ParamStore:
    def __init__(self, spec, *, device, quantise=True, quant_excluded=(), wd_excluded=(), block_size=16,
                 with_ema=False, trainable=True, quant_mask=None, decay_mask=None, grad_bf16=True):
        """spec: ordered list of (path, shape) in forward-execution order.  quant_mask / decay_mask: explicit {path: bool}
        trees (True = quantise / decay) in place of the exclusion patterns (lion_quant.lion_8bit takes masks).
        grad_bf16=False keeps every gradient float32 (lion_quant's GradientTransformation facade: its caller hands in float32 updates
        and must get the arithmetic of lion_quant.py on exactly those)."""
        self.device = torch.device(device)
        self.block_size = block_size
        self.trainable = trainable
        paths = [p for p, _ in spec]
        qmask = create_mask(paths, quant_excluded) if (quantise and trainable) else {p: False for p in paths}
        dmask = create_mask(paths, wd_excluded) if wd_excluded else {p: True for p in paths}
        if quant_mask is not None:
            qmask = {p: bool(quant_mask[p]) and trainable for p in paths}
        if decay_mask is not None:
            dmask = {p: bool(decay_mask[p]) for p in paths}
        segs = {(True, True): [], (True, False): [], (False, True): [], (False, False): []}
        for p, shp in spec:
            segs[(qmask[p], dmask[p])].append((p, tuple(shp)))
        self.leaves = {}
        self.segments = []  # (quantised, decayed, start, end)
        off = 0
        order = []
        padded = []
        for key in ((True, True), (True, False), (False, True), (False, False)):
            start = off
            for p, shp in segs[key]:
                n = math.prod(shp)
                if key[0] and n % block_size != 0:
                    raise ValueError(f"{p}: numel {n} is not a multiple of quant_block_size {block_size} "
                                     "(lion_quant.py:70 reshape(-1, block_size) would fail too)")
                lf = Leaf(p, shp, n, off, key[0], key[1])
                if p.endswith("/kernel") and len(shp) in (2, 4):
                    if len(shp) == 2:
                        lf.batch, lf.R, lf.C = 1, shp[0], shp[1]
                    else:
                        lf.batch, lf.R, lf.C = shp[0] * shp[1], shp[2], shp[3]
                    lf.Rp, lf.Cp = _ceil(lf.R, 8), _ceil(lf.C, 8)
                    if (lf.Rp, lf.Cp) != (lf.R, lf.C):  # zero-padded copy (4-channel latents, 3-channel pixels): its own slot
                        lf.w_off = -2
                        padded.append(lf)
                    else:  # W is the bf16 mirror of the master, element for element: the optimizer sweep writes it
                        lf.w_off = off
                self.leaves[p] = lf
                order.append(p)
                off = _ceil(off + n, 8)  # 32-byte fp32 / 16-byte bf16 alignment of every leaf (W views are GEMM operands)
            off = _ceil(off, SEG_ALIGN)
            self.segments.append((key[0], key[1], start, off))
        self.order = order
        self.total = off
        self.quant_total = self.segments[1][3]  # end of the two quantised segments
        dev = self.device
        self.master = torch.zeros(self.total, dtype=torch.float32, device=dev)
        wp = self.total
        for lf in padded:
            lf.w_off = wp
            wp += lf.batch * lf.Rp * lf.Cp
        # w: [0, total) mirrors the master in bf16 (Flax layouts: W of every unpadded matrix leaf lives at its master offset),
        # followed by the zero-padded copies of the few leaves whose channel counts are not multiples of 8
        self.w = torch.zeros(max(wp, 8), dtype=torch.bfloat16, device=dev)
        self._padded = padded
        self.thresholds = lion_thresholds(dev) if trainable else None
        # Gradient storage [r4]: the QUANTISED segments - the Dense / conv kernels, > 96 % of the parameters - keep their gradients in
        # bf16 (grad16, elements [0, quant_total)): the reference's kernel cotangents have exactly that precision (flax Dense / Conv
        # with dtype=bfloat16 cast the fp32 kernel to bf16, so the transposed contraction hands back a bf16 value that optax
        # widens), the weight-gradient kernels round their fp32 sums once when they store, and the Lion-8bit sweep widens them
        # again: 2 B instead of 4 B per parameter through the wgrad stores, the norm pass, the optimizer sweep and the gradient
        # exchange.  Everything else (biases, norm parameters, embeddings, unquantised kernels: [quant_total, total)) stays
        # fp32 in `grad`.  A store with a quantised leaf that is NOT a matrix kernel (its gradient is accumulated in fp32 by the
        # norm / embedding kernels) keeps the whole buffer fp32 (SDT_GRAD_BF16=0 does the same: developer A/B).
        self.grad16 = None
        self.g32_base = 0
        if trainable:
            bf16_ok = (grad_bf16 and self.quant_total > 0 and os.environ.get("SDT_GRAD_BF16", "1") != "0"
                       and all(lf.w_off != -1 for lf in self.leaves.values() if lf.quantised))
            if bf16_ok:
                self.grad16 = torch.zeros(self.quant_total, dtype=torch.bfloat16, device=dev)
                self.g32_base = self.quant_total
            self.grad = torch.zeros(max(self.total - self.g32_base, 4), dtype=torch.float32, device=dev)
            self.codes = torch.full((max(self.quant_total, 4),), 3, dtype=torch.int8, device=dev)  # quant(0) == 3
            self.inv_scale = torch.ones(max(self.quant_total // block_size, 1), dtype=torch.float32, device=dev)
            self.mom = torch.zeros(max(self.total - self.quant_total, 4), dtype=torch.float32, device=dev)
            self.sqnorm = torch.zeros(1, dtype=torch.float64, device=dev)
            # arrival counter + per-workgroup double partials of the gradient-norm pass (ordered sum, no atomics; zeroed once)
            self.sq_ws = torch.zeros(_lib.load().sdt_sqnorm_workspace_bytes() if dev.type == "cuda" else 8, dtype=torch.uint8, device=dev)
            self.ema = torch.zeros(self.total, dtype=torch.float32, device=dev) if with_ema else None
        else:
            self.grad = self.codes = self.inv_scale = self.mom = self.sqnorm = self.ema = self.sq_ws = None
        self.count = 0
        self._prep = None
        self._zero = None
        # Sharded optimizer (dp.GradReducer(shard=True)): fp32 master / EMA / momentum of a scattered slice are current only on
        # its owner.  sharded marks the store; state_whole is False from a sharded sweep until GradReducer.gather_state() - a
        # COLLECTIVE every rank calls - has made the buffers whole again.  Exports refuse to read a store that is not whole.
        self.sharded = False
        self.state_whole = True
        self._written = None  # armed by zero_grad(): paths whose gradient has been written this step (note_written)

    # ------------------------------------------------------------------ views
    def p(self, path):
        lf = self.leaves[path]
        return self.master[lf.offset: lf.offset + lf.numel].view(lf.shape)

    def g(self, path):
        lf = self.leaves[path]
        return self.grad_view(lf.offset, lf.offset + lf.numel).view(lf.shape)

    def grad_view(self, a, b):
        """Elements [a, b) of the gradient: a bf16 view when the range lies in the quantised segments of a store that keeps those in
        bf16 (grad16), a float32 view otherwise.  A range never straddles the two (segment ends are bucket and leaf boundaries)."""
        if self.grad16 is not None and a < self.quant_total:
            if b > self.quant_total:
                raise ValueError(f"gradient range [{a}, {b}) straddles the bf16 / fp32 boundary at {self.quant_total}")
            return self.grad16[a:b]
        return self.grad[a - self.g32_base: b - self.g32_base]

    def grad_bytes(self):
        """Bytes of one rank's gradient (the exchange payload)."""
        return 4 * (self.total - self.g32_base) + (2 * self.quant_total if self.grad16 is not None else 0)

    def has(self, path):
        return path in self.leaves

    def mergeable(self, wpaths, bpaths=None):
        """True when the Dense kernels `wpaths` (same [in,out], unpadded) sit back to back in the master / grad buffers and the
        bf16 mirror (and their biases back to back too), so one GEMM can serve them all (ops.linear_multi)."""
        lfs = [self.leaves[p] for p in wpaths]
        a = lfs[0]
        if a.batch != 1 or a.R != a.Rp or a.C != a.Cp or a.C % 64:
            return False
        n = a.R * a.C
        for i, lf in enumerate(lfs):
            if (lf.batch, lf.R, lf.C, lf.Rp, lf.Cp) != (1, a.R, a.C, a.R, a.C):
                return False
            if lf.offset != a.offset + i * n or lf.w_off != a.w_off + i * n:
                return False
        if bpaths is not None:
            bs = [self.leaves[p] for p in bpaths]
            if any(b.offset != bs[0].offset + i * a.C or b.numel != a.C for i, b in enumerate(bs)):
                return False
        return True

    def load(self, tensors, init_ema=True):
        """Copy a {path: tensor} tree (Flax layouts) into the master buffer (host or device tensors)."""
        self.begin_external_write()
        for p, lf in self.leaves.items():
            t = tensors[p]
            if tuple(t.shape) != lf.shape:
                raise ValueError(f"{p}: expected shape {lf.shape}, got {tuple(t.shape)}")
            self.p(p).copy_(t.to(device=self.device, dtype=torch.float32))
        if self.ema is not None and init_ema:
            self.ema.copy_(self.master)
        if self.device.type == "cuda":
            self.prepare(full=True)  # whoever writes the master refreshes the bf16 copies
        self.state_whole = True      # every rank has just written the whole buffers

    def begin_external_write(self):
        """Call before master / EMA / momentum / W are written from outside a step (load, load_training_state).  Sharded optimizer:
        the all-gather of the bf16 mirrors of the last step may still be running on the communication stream and would overwrite
        what the load is about to put into W; a device synchronize drains it (dp.GradReducer.wait_gathered's contract)."""
        if self.sharded and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _gather(self):
        """Exports and checkpoints read master / EMA / momentum of EVERY leaf: with the sharded optimizer that needs the
        collective GradReducer.gather_state() first, called on every rank (a save under `if rank == 0:` must not start one)."""
        if self.sharded and not self.state_whole:
            raise RuntimeError("sharded optimizer: fp32 master / EMA / momentum slices are current only on their owning ranks; call "
                               "GradReducer.gather_state() on EVERY rank before exporting or saving this store")

    def set_grad_flat(self, flat):
        """Write a whole float32 gradient (master order, length `total`): the kernel leaves' part is rounded to bf16 where the store keeps
        it so (tests, host-side gradient injection)."""
        flat = flat.to(self.device)
        if self.grad16 is not None:
            self.grad16.copy_(flat[: self.quant_total])
        self.grad[: self.total - self.g32_base].copy_(flat[self.g32_base: self.total])

    def fill_grad(self, value):
        self.grad.fill_(value)
        if self.grad16 is not None:
            self.grad16.fill_(value)

    def grad_flat(self):
        """The whole gradient as float32 in master order (a copy when part of it is kept in bf16)."""
        return self._whole_grad()

    def _whole_grad(self):
        """The gradient as one float32 buffer in master order (exports / tests): the bf16 part widened exactly."""
        if self.grad16 is None:
            return self.grad[: self.total]
        return torch.cat([self.grad16.float(), self.grad[: self.total - self.g32_base]])

    def export(self, which="master"):
        self._gather()
        buf = self._whole_grad() if which == "grad" else {"master": self.master, "ema": self.ema}[which]
        return {p: buf[lf.offset: lf.offset + lf.numel].view(lf.shape).detach().clone() for p, lf in self.leaves.items()}

    def export_host(self, which="master"):
        """{path: numpy view} over ONE device->host copy of the flat buffer (checkpoint writers; ~700 leaves per UNet)."""
        self._gather()
        buf = self._whole_grad() if which == "grad" else {"master": self.master, "ema": self.ema}[which]
        host = buf.detach().cpu().numpy()
        return {p: host[lf.offset: lf.offset + lf.numel].reshape(lf.shape) for p, lf in self.leaves.items()}

    def export_momentum(self):
        """{path: (codes int8 [n/bs,bs], inv_scale f32 [n/bs,1])} for quantised leaves, f32 array otherwise."""
        self._gather()
        out = {}
        bs = self.block_size
        for p, lf in self.leaves.items():
            if lf.quantised:
                c = self.codes[lf.offset: lf.offset + lf.numel].view(-1, bs).clone()
                s = self.inv_scale[lf.offset // bs: (lf.offset + lf.numel) // bs].view(-1, 1).clone()
                out[p] = (c, s)
            else:
                o = lf.offset - self.quant_total
                out[p] = self.mom[o: o + lf.numel].view(lf.shape).clone()
        return out

    # ------------------------------------------------------------------ bf16 compute copies
    def _build_prep(self):
        """Two descriptor tables for sdt_param_prepare (fp32 master -> bf16 W): `full` lists every matrix leaf; `step` only the
        zero-padded ones - the optimizer sweep itself mirrors the master into W for all the others."""
        tables = {}
        for which in ("full", "step"):
            descs, tile0 = [], 0
            for p in self.order:
                lf = self.leaves[p]
                if lf.w_off < 0 or (which == "step" and lf.w_off == lf.offset):
                    continue
                if which == "step" and self.sharded and lf.quantised:
                    continue  # its master is stale on the ranks that do not own it: built from the gathered mirror (prepare)
                descs.append(_lib.SdtPrepDesc(lf.offset, lf.w_off, 0, lf.batch, lf.R, lf.C, lf.Rp, lf.Cp, tile0, 0))
                tile0 += lf.batch * ((lf.Rp + 63) // 64) * ((lf.Cp + 63) // 64)
            if not descs:
                tables[which] = (None, 0, 0)
                continue
            arr = (_lib.SdtPrepDesc * len(descs))(*descs)
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            tables[which] = (dev, len(descs), tile0)
        self._prep = tables

    def prepare(self, stream=None, full=False):
        """Make the bf16 compute copy current (one launch).  full: W of every matrix leaf from the fp32 master (after the master
        was written from outside: load(), a checkpoint).  Otherwise (start of a training step) W already mirrors the master -
        the optimizer sweep wrote it - and only the few zero-padded leaves are converted."""
        if self._prep is None:
            self._build_prep()
        whole = full or not self.trainable
        dev, nd, tiles = self._prep["full" if whole else "step"]
        if stream is not None and stream != torch.cuda.current_stream().cuda_stream:
            # the copies below run on torch's current stream: one stream for the whole conversion, or the writes to W are unordered
            raise _lib.SdtError("ParamStore.prepare: make the stream current (torch.cuda.stream) instead of passing another one")
        s = torch.cuda.current_stream().cuda_stream
        if nd:
            _lib.call("sdt_param_prepare", self.master.data_ptr(), self.w.data_ptr(), None, dev.data_ptr(), nd, tiles, s)
        for lf in self._padded:
            mirror = self.w[lf.offset: lf.offset + lf.numel].view(lf.batch, lf.R, lf.C)
            if whole:  # the mirror slot of a padded leaf is otherwise only written by the optimizer sweep
                mirror.copy_(self.master[lf.offset: lf.offset + lf.numel].view(lf.batch, lf.R, lf.C))
            elif self.sharded and lf.quantised:
                # sharded optimizer: the all-gathered bf16 mirror is current on every rank, the fp32 master only on the owner of
                # the slice; the padded copy is the same bf16 values re-pitched (pad lanes stay zero)
                self.w[lf.w_off: lf.w_off + lf.batch * lf.Rp * lf.Cp].view(lf.batch, lf.Rp, lf.Cp)[:, :lf.R, :lf.C].copy_(mirror)

    def wmat(self, path):
        """(W view [batch,Rp,Cp], leaf) of a kernel leaf: the bf16 compute copy, Flax layout."""
        lf = self.leaves[path]
        n = lf.batch * lf.Rp * lf.Cp
        return self.w[lf.w_off: lf.w_off + n].view(lf.batch, lf.Rp, lf.Cp), lf

    # ------------------------------------------------------------------ optimizer
    def _build_zero_ranges(self):
        """float4 ranges of the gradient buffer that kernels ACCUMULATE into and that therefore start each step at zero: norm
        scales / biases (atomic partial sums) and embeddings (scatter-add).  Dense / conv kernels and their biases are
        written whole by sdt_gemm_tn_wgrad (single writer per element) and are left alone - clearing them was a 3.9 GB
        fill per step."""
        written = set()
        for p, lf in self.leaves.items():
            if lf.w_off >= 0:
                written.add(p)
                b = p[: -len("kernel")] + "bias"
                if b in self.leaves:
                    written.add(b)
        # everything that is not a written leaf: accumulated-into leaves AND the alignment gaps between leaves (the global-norm
        # and optimizer sweeps run over whole segments, gaps included, so gaps must hold zeros whatever the buffer held before)
        spans, pos = [], 0  # element ranges
        for p in self.order:  # offsets increase along self.order and are multiples of 8
            if p not in written:
                continue
            lf = self.leaves[p]
            if lf.offset > pos:  # (rounding the start down may clear the tail of the written leaf before it: it is rewritten later)
                spans.append([pos, lf.offset])
            pos = lf.offset + lf.numel
        if pos < self.total:
            spans.append([pos, self.total])
        # one table per buffer, in 16-byte units relative to its base: 8 bf16 elements (grad16) / 4 float32 elements (grad)
        chunk = _lib.load().sdt_zero_ranges_chunk()
        tables = []
        for buf, lo, hi, per in ((self.grad16, 0, self.quant_total, 8), (self.grad, self.g32_base, self.total, 4)):
            flat = []
            if buf is not None:
                for a, b in spans:
                    a, b = max(a, lo), min(b, hi)
                    if a >= b:
                        continue
                    ua, ub = (a - lo) // per, (b - lo + per - 1) // per  # (b is a leaf offset, the buffer end or quant_total: all multiples of 8)
                    for c in range(ua, ub, chunk):
                        flat += [c, min(chunk, ub - c)]
            tables.append((buf, torch.tensor(flat, dtype=torch.int64).to(self.device) if flat else None, len(flat) // 2))
        self._zero = tables

    def note_written(self, path):
        """ops reports every gradient leaf its backward kernels have produced.  Kernel / bias gradients are WRITTEN, not
        accumulated (one writer per step, no zero fill): a leaf consumed twice between two zero_grad() calls - tied weights, two
        text-encoder calls, micro-batch accumulation - would silently keep only its last contribution, so that is refused."""
        w = self._written
        if w is None:
            return
        if path in w:
            raise RuntimeError(f"{path}: gradient produced twice in one step; Dense / conv gradients are written, not accumulated "
                               "(single use per step: INTEGRATION.md)")
        w.add(path)

    def zero_grad(self, everything=False):
        """Start of a step: clear the accumulated-into leaves (one launch).  everything=True clears the whole buffer.
        Arms the single-use check (note_written) until the optimizer step."""
        self._written = set()
        if everything:
            self.grad.zero_()
            if self.grad16 is not None:
                self.grad16.zero_()
            return
        if self._zero is None:
            self._build_zero_ranges()
        for buf, dev, n in self._zero:
            if n:
                _lib.call("sdt_zero_ranges", buf.data_ptr(), dev.data_ptr(), n, torch.cuda.current_stream().cuda_stream)

    def optimizer_step(self, *, lr, wd, b1=0.9, b2=0.99, max_norm=1.0, ema_rate=0.0, stream=None, shard=None, sq_partials=None):
        """clip_by_global_norm(max_norm) -> Lion (8-bit / fp32 momentum) -> decay -> -lr -> apply (-> EMA).
        training_utils.py:379-387 + :732 + :735-746, fused; no host synchronisation (the norm stays on device).
        max_norm None: no clipping (the bare lion_8bit transformation, lion_quant.py:159-211).
        shard: None, or (pieces, sq_done) from dp.GradReducer (sharded optimizer): pieces = [(a, b, quantised, decayed)] element
        ranges this rank updates (its slices of the quantised buckets + the replicated non-quantised segments); sq_done: the
        squared norm of the sharded part is already in self.sqnorm, all-reduced over the ranks - only the replicated part is
        added here."""
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        sq_ptr = None
        if shard is None:
            pieces = [(a, b, q, d) for (q, d, a, b) in self.segments]
            norm_ranges = [(0, self.total)]
            if sq_partials is not None:
                # (slots, used) from ops.sq_end: the weight-gradient kernels left the sums of squares of every quantised leaf's gradient
                # in their slots; only the non-quantised segments (biases, norm parameters, embeddings) are read back here
                norm_ranges = [(a, b) for (q, d, a, b) in self.segments if not q]
        else:
            pieces, _ = shard
            norm_ranges = [(a, b) for (a, b, q, d) in pieces if not q]
        if max_norm is not None:
            if shard is None:
                self.sqnorm.zero_()
                if sq_partials is not None and sq_partials[1]:
                    _lib.call("sdt_sum_f64_accumulate", sq_partials[0].data_ptr(), sq_partials[1], self.sqnorm.data_ptr(),
                              self.sq_ws.data_ptr(), self.sq_ws.numel(), s)
            for a, b in norm_ranges:
                self.sqnorm_accumulate(a, b, s)
            sq_ptr = self.sqnorm.data_ptr()
        else:
            max_norm = 1.0
        ema_on = self.ema is not None and ema_rate
        for (a, b, quant, decay) in pieces:
            n = b - a
            if n == 0:
                continue
            ema_ptr = self.ema.data_ptr() + 4 * a if ema_on else None
            wd_eff = wd if decay else 0.0
            if quant:
                g16 = self.grad16 is not None
                _lib.call("sdt_lion8_step", self.master.data_ptr() + 4 * a, self.grad16.data_ptr() + 2 * a if g16 else self.grad.data_ptr() + 4 * a,
                          int(g16), self.codes.data_ptr() + a, self.inv_scale.data_ptr() + 4 * (a // self.block_size), ema_ptr,
                          self.w.data_ptr() + 2 * a, n, self.block_size, sq_ptr, self.thresholds.data_ptr(), max_norm, lr,
                          wd_eff, b1, b2, ema_rate if ema_on else 0.0, s)
            else:
                _lib.call("sdt_lion32_step", self.master.data_ptr() + 4 * a, self.grad.data_ptr() + 4 * (a - self.g32_base),
                          self.mom.data_ptr() + 4 * (a - self.quant_total), ema_ptr, self.w.data_ptr() + 2 * a, n, sq_ptr,
                          max_norm, lr, wd_eff, b1, b2, ema_rate if ema_on else 0.0, s)
        self.count += 1
        self._written = None
        if shard is not None and self.sharded:
            self.state_whole = False

    def sqnorm_accumulate(self, a, b, stream=None):
        """self.sqnorm += sum of squares of gradient elements [a, b) (double; ordered partial sums), whichever buffer(s) hold them."""
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        q = self.quant_total if self.grad16 is not None else 0
        if a < min(b, q):
            _lib.call("sdt_sqnorm_accumulate_bf16", self.grad16.data_ptr() + 2 * a, min(b, q) - a, self.sqnorm.data_ptr(), self.sq_ws.data_ptr(),
                      self.sq_ws.numel(), s)
        a = max(a, q)
        if b > a:
            _lib.call("sdt_sqnorm_accumulate", self.grad.data_ptr() + 4 * (a - self.g32_base), b - a, self.sqnorm.data_ptr(), self.sq_ws.data_ptr(),
                      self.sq_ws.numel(), s)

    def grad_norm(self):
        """Host read of the last step's global gradient norm (forces a sync; logging only)."""
        return float(self.sqnorm.sqrt().item())

    def shard_buckets(self, world, bucket_bytes=64 << 20):
        """Buckets for the sharded optimizer (dp.GradReducer shard=True): [(a, b, quantised, decayed)], each inside ONE segment and
        (b - a) a multiple of world * 256, covering the whole buffer in offset order.  Rank r owns elements
        [a + r*(b-a)/world, a + (r+1)*(b-a)/world) of every QUANTISED bucket (gradient reduce-scatter, clip + Lion-8bit + EMA on
        the slice, all-gather of the bf16 mirror); non-quantised buckets (biases, norms, embeddings: read from the fp32 master by the
        forward kernels) stay replicated."""
        if SEG_ALIGN % (world * 256):
            raise ValueError(f"sharded optimizer: world size {world} does not divide {SEG_ALIGN // 256}")
        out = []
        for quant, decay, a, b in self.segments:
            width = 2 if (quant and self.grad16 is not None) else 4  # bytes per gradient element of this segment
            per = max(bucket_bytes // width // SEG_ALIGN, 1) * SEG_ALIGN
            while a < b:
                e = min(a + per, b)
                out.append((a, e, quant, decay))
                a = e
        return out

    def bucket_ranges(self, bucket_bytes=64 << 20):
        """Contiguous [start,end) element ranges of the gradient, each inside ONE of its buffers (bf16 kernel gradients / float32 rest)
        and about bucket_bytes long, + the leaves each one needs."""
        ranges = []
        regions = [(0, self.total, 4)] if self.grad16 is None else [(0, self.quant_total, 2), (self.quant_total, self.total, 4)]
        for lo, hi, width in regions:
            per = max(bucket_bytes // width, 1)
            a = lo
            while a < hi:
                b = min(a + per, hi)
                ranges.append((a, b))
                a = b
        owners = [[] for _ in ranges]
        starts = [a for a, _ in ranges]
        import bisect
        for p, lf in self.leaves.items():
            lo, hi = lf.offset, lf.offset + max(lf.numel, 1) - 1
            i = max(bisect.bisect_right(starts, lo) - 1, 0)
            while i < len(ranges) and ranges[i][0] <= hi:
                if ranges[i][1] > lo:
                    owners[i].append(p)
                i += 1
        return ranges, owners
