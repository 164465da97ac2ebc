"""Host half of the 8-bit Lion codec (reference lion_quant.py:49-64): the decision thresholds of `_quantize`.

`_quantize(x) = round_half_even(sign(x + offset) * |x + offset| ** (1/5) * 127)` is a monotone step function of
a = |x + offset|, so it is fully described by 127 float32 thresholds T[c] (c = 1..127): the smallest a whose code is >= c.
The optimizer kernel (csrc/optimizer.hip) estimates the code with v_log/v_exp and settles it against T, which makes its
integer exactly the one float32 `power` produces on the host - no ulp disagreement between a device powf and the host's at
the rounding boundaries.  The table is built here, once, with the same float32 NumPy operations the definition spells out
(power, multiply, rint), by bisection over the float32 bit patterns of [0, 1]."""
import numpy as np

OFFSET = np.float32(3.7398995e-09)  # lion_quant.py:49
_F32 = np.float32
_TABLE = None


def _code_of_abs(a):
    """lion_quant.py:55-58 for a = |x + offset| >= 0 (the sign is applied afterwards and rint is odd-symmetric)."""
    q = np.power(a.astype(_F32), _F32(1 / 5)).astype(_F32)
    return np.rint((q * _F32(127)).astype(_F32)).astype(np.int32)


def quantization_thresholds():
    """float32[128]: T[0] = 0 and, for c = 1..127, the smallest float32 a in (0, 1] with code(a) >= c."""
    global _TABLE
    if _TABLE is None:
        c = np.arange(1, 128, dtype=np.int32)
        lo = np.zeros(127, dtype=np.uint32)                                  # code(0) = 0 < c
        hi = np.full(127, np.array(1.0, _F32).view(np.uint32), np.uint32)    # code(1) = 127 >= c
        assert int(_code_of_abs(np.zeros(1, _F32))[0]) == 0 and int(_code_of_abs(np.ones(1, _F32))[0]) == 127
        while np.any(hi - lo > 1):  # positive float32 values order like their bit patterns
            mid = lo + (hi - lo) // 2
            ge = _code_of_abs(mid.view(_F32)) >= c
            hi = np.where(ge, mid, hi)
            lo = np.where(ge, lo, mid)
        t = np.zeros(128, _F32)
        t[1:] = hi.view(_F32)
        if not np.all(np.diff(t[1:]) > 0):
            raise RuntimeError("lion codec thresholds are not strictly increasing (float32 power is not monotone here)")
        _TABLE = t
    return _TABLE


def quantize_reference(x):
    """Codes by table lookup (what the device kernel computes); used by the host-side self check in tests."""
    xo = (np.asarray(x, _F32) + OFFSET).astype(_F32)
    t = quantization_thresholds()
    mag = np.searchsorted(t[1:], np.abs(xo), side="right").astype(np.int32)
    return (np.sign(xo).astype(np.int32) * mag).astype(np.int8)
