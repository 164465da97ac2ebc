"""Oracle (test infrastructure): global-norm clip, 8-bit blockwise Lion, fp32 Lion, EMA.

NumPy float32 restatement of
  * lion_quant.py:49-64    offset / _quantize / _dequantize (5th-root int8 codec)
  * lion_quant.py:66-92    _block_quantize / _block_dequantize (stored scale = 1/absmax)
  * lion_quant.py:97-113   _update_moment_quant
  * lion_quant.py:115-131  init_fn (quantised zeros -> code 3, inverse scale 1)
  * lion_quant.py:133-154  update_fn
  * lion_quant.py:201-211  chain: scale_by_lion_8bit -> add_decayed_weights -> -lr
  * training_utils.py:116-131  create_mask (exact path-component match)
  * training_utils.py:355-387  hyper-parameters: lr=1e-6/7, wd=1e-2*7, b1=.9, b2=.99, clip 1
  * training_utils.py:537-544  compute_model_ema
and of the optax pieces the reference chains (third-party, not in /root/reference):
  optax.clip_by_global_norm, optax.lion (scale_by_lion), add_decayed_weights,
  _scale_by_learning_rate, apply_updates.
"""
import numpy as np

F32 = np.float32
OFFSET = F32(3.7398995e-09)  # lion_quant.py:49
MIN_NORM = F32(0.0)  # lion_quant.py:50


def quantize(x):
    # lion_quant.py:52-59
    x = np.asarray(x, dtype=F32)
    xo = (x + OFFSET).astype(F32)
    s = np.sign(xo).astype(F32)
    q = np.power(np.abs(xo), F32(1 / 5)).astype(F32)
    q = ((q * s) * F32(127)).astype(F32)
    return np.rint(q).astype(np.int8)  # rint == round-half-even == jnp.round


def dequantize(q):
    # lion_quant.py:61-64 ; (q/127)**5 is an integer power -> repeated multiply in f32
    t = (np.asarray(q).astype(F32) / F32(127)).astype(F32)
    t2 = (t * t).astype(F32)
    t4 = (t2 * t2).astype(F32)
    return ((t4 * t).astype(F32) - OFFSET).astype(F32)


def block_quantize(leaf, block_size):
    # lion_quant.py:66-80
    flat = np.asarray(leaf, dtype=F32).reshape(-1, block_size)
    absmax = np.max(np.abs(flat), axis=-1, keepdims=True).astype(F32)
    inv = (F32(1) / np.where(absmax <= MIN_NORM, F32(1), absmax)).astype(F32)
    codes = quantize((flat * inv).astype(F32))
    return codes, inv


def block_dequantize(shape, codes, inv):
    # lion_quant.py:82-92
    return (dequantize(codes) / inv).astype(F32).reshape(shape)


def init_state(params, quant_mask, block_size):
    """lion_quant.py:115-131. ``params``/``quant_mask``: flat dicts keyed by '/'-joined path."""
    mu = {}
    for k, p in params.items():
        if quant_mask is not None and quant_mask[k]:
            mu[k] = block_quantize(np.zeros(p.shape, F32), block_size)
        else:
            mu[k] = np.zeros(p.shape, F32)
    return {"count": 0, "mu": mu}


def create_mask(paths, excluded):
    """training_utils.py:116-131: leaf included iff NO path component equals an excluded name."""
    out = {}
    for k in paths:
        comps = tuple(k.split("/"))
        out[k] = not any(e in comps for e in excluded)
    return out


def global_norm(grads):
    # optax.global_norm: sqrt(sum_leaves sum(g^2)) ; accumulate in f64 then round (order-free)
    tot = 0.0
    for g in grads.values():
        g64 = np.asarray(g, dtype=np.float64)
        tot += float(np.sum(g64 * g64))
    return F32(np.sqrt(tot))


def clip_by_global_norm(grads, max_norm=1.0):
    # optax.clip_by_global_norm (training_utils.py:380, :417)
    n = global_norm(grads)
    if n < F32(max_norm):
        return {k: np.asarray(g, F32) for k, g in grads.items()}, n
    return {k: ((np.asarray(g, F32) / n) * F32(max_norm)).astype(F32) for k, g in grads.items()}, n


def lion_step(params, grads, state, *, lr, wd, b1=0.9, b2=0.99, block_size=16,
              decay_mask=None, clip=1.0):
    """One optimizer application = TrainState.apply_gradients (training_utils.py:732-733)
    with tx = chain(clip_by_global_norm(1), lion_8bit | optax.lion) (training_utils.py:355-387).

    Returns (new_params, new_state, grad_norm).  Quantised leaves are those whose
    state entry is a (codes, inv_scale) tuple (lion_quant.py:94-95)."""
    g_clip, gnorm = clip_by_global_norm(grads, clip) if clip else (grads, global_norm(grads))
    c1, c1m = F32(b1), F32(1.0 - b1)
    c2, c2m = F32(b2), F32(1 - b2)
    new_p, new_mu = {}, {}
    for k, p in params.items():
        p = np.asarray(p, F32)
        g = np.asarray(g_clip[k], F32)
        m = state["mu"][k]
        if isinstance(m, tuple):
            mf = block_dequantize(p.shape, m[0], m[1])
        else:
            mf = m
        u = np.sign((c1m * g + c1 * mf).astype(F32)).astype(F32)  # lion_quant.py:141-145
        mnew = (c2m * g + c2 * mf).astype(F32)  # lion_quant.py:105-109
        new_mu[k] = block_quantize(mnew, block_size) if isinstance(m, tuple) else mnew
        if decay_mask is None or decay_mask[k]:
            u = (u + F32(wd) * p).astype(F32)  # add_decayed_weights
        u = (F32(-lr) * u).astype(F32)  # _scale_by_learning_rate
        new_p[k] = (p + u).astype(F32)  # apply_updates
    return new_p, {"count": state["count"] + 1, "mu": new_mu}, gnorm


def ema_update(ema, params, rate):
    # training_utils.py:537-544
    r, rm = F32(rate), F32(1 - rate)
    return {k: (r * np.asarray(ema[k], F32) + rm * np.asarray(params[k], F32)).astype(F32) for k in ema}
