"""The reference's optimizer module surface (lion_quant.py:12-17, 20-156, 159-211) over the HIP kernels: `lion_8bit(...)` returns
an optax-shaped (init, update) pair whose state is `ScaleBy8bitLionState(count, mu_quant, mu_quant_flag)`.

train_step does not go through this facade - `create_lion_optimizer_states` builds the flat `ParamStore` directly and
`ParamStore.optimizer_step` fuses clip + Lion + decay + lr + apply + EMA into one sweep.  The facade exists
for callers that hold the reference's optimizer contract: trees are flat `{"path/with/slashes": device tensor}`
dicts (the reference's pytrees, flattened), `update` returns the optax `updates` tree (`params + updates` = the stepped
parameters), and the arithmetic is `sdt_lion8_step` / `sdt_lion32_step` - no CPU fallback."""
from typing import Any, Callable, NamedTuple

import torch

from .params import ParamStore


class ScaleBy8bitLionState(NamedTuple):
    """lion_quant.py:12-17.  mu_quant: {path: (int8 codes [n/bs, bs], f32 inverse scales [n/bs, 1])} for quantised leaves, the
    fp32 momentum otherwise; mu_quant_flag: the quantisation mask it was built with."""
    count: Any
    mu_quant: Any
    mu_quant_flag: Any


class GradientTransformation(NamedTuple):
    init: Callable
    update: Callable


def lion_8bit(learning_rate, b1=0.9, b2=0.99, mu_scale_dtype=None, block_size=64, weight_decay=1e-3, mask=None,
              excluded_layer_mask=None):
    """lion_quant.py:159-211: chain(scale_by_lion_8bit, add_decayed_weights(weight_decay, mask), scale_by_learning_rate).
    mask: {path: bool}, True = decay; excluded_layer_mask: {path: bool}, True = quantise that leaf's momentum (the reference's
    argument name notwithstanding, lion_quant.py:203-205).  mu_scale_dtype is accepted and ignored (fp32 scales)."""
    if callable(learning_rate):
        raise NotImplementedError("learning-rate schedules: pass the current value (the reference trains at a constant rate)")
    holder = {}

    def _store(params):
        st = holder.get("store")
        if st is None:
            dev = next(iter(params.values())).device
            spec = [(p, tuple(v.shape)) for p, v in params.items()]
            qm = excluded_layer_mask if excluded_layer_mask is not None else {p: False for p in params}
            st = holder["store"] = ParamStore(spec, device=dev, block_size=block_size, quant_mask=qm,
                                              decay_mask=mask if mask is not None else {p: True for p in params},
                                              grad_bf16=False)  # the caller's float32 updates, unrounded (lion_quant.py:133-154)
        return st

    def _state(st):
        return ScaleBy8bitLionState(count=st.count, mu_quant=st.export_momentum(),
                                    mu_quant_flag={p: lf.quantised for p, lf in st.leaves.items()})

    def init_fn(params):
        st = _store(params)  # codes of quantise(0) and unit scales / zero momenta (lion_quant.py:115-131)
        return _state(st)

    def update_fn(updates, state, params=None):
        if params is None:
            raise ValueError("lion_8bit.update needs params (add_decayed_weights reads them)")
        st = _store(params)
        if state.count != st.count:
            raise ValueError("lion_8bit: state does not belong to this transformation's last step")
        st.load(params, init_ema=False)
        for p in st.leaves:
            st.g(p).copy_(updates[p])
        st.optimizer_step(lr=float(learning_rate), wd=float(weight_decay), b1=b1, b2=b2, max_norm=None)
        new = st.export("master")
        return {p: new[p] - params[p].to(torch.float32) for p in params}, _state(st)

    return GradientTransformation(init_fn, update_fn)
