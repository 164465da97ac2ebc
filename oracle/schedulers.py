"""Oracle (test infrastructure): DDPM beta tables, add_noise, get_velocity, min-SNR weights.

Restates, in NumPy float32 (jnp default dtype in the reference):
  * schedulers/scheduling_utils_flax.py:193-219  betas_for_alpha_bar
  * schedulers/scheduling_utils_flax.py:222-263  rescale_betas (zero terminal SNR)
  * schedulers/scheduling_utils_flax.py:266-313  CommonSchedulerState.create
  * schedulers/scheduling_utils_flax.py:316-343  get_sqrt_alpha_prod / add_noise_common / get_velocity_common
  * schedulers/scheduling_ddpm_flax.py:111-124, 281-297  create_state / add_noise / get_velocity
  * training_utils.py:531-568  compute_snrs / min_snr_gamma_loss_rescale
"""
import math

import numpy as np

F32 = np.float32


def betas_for_alpha_bar(n, max_beta=0.999):
    # scheduling_utils_flax.py:193-219
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    out = []
    for i in range(n):
        t1, t2 = i / n, (i + 1) / n
        out.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.asarray(out, dtype=F32)


def rescale_betas(betas):
    # scheduling_utils_flax.py:222-263 ; all arithmetic stays float32 like jnp
    betas = betas.astype(F32)
    alphas = (F32(1) - betas).astype(F32)
    alphas_bar = np.cumprod(alphas, dtype=F32)
    abs_ = np.sqrt(alphas_bar).astype(F32)
    a0 = abs_[0]
    aT = abs_[-1]
    abs_ = (abs_ - aT).astype(F32)
    abs_ = (abs_ * a0 / (a0 - aT)).astype(F32)
    alphas_bar = (abs_ ** 2).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        alphas = (alphas_bar[1:] / alphas_bar[:-1]).astype(F32)
    alphas = np.concatenate([alphas_bar[0:1], alphas]).astype(F32)
    return (F32(1) - alphas).astype(F32)


def make_betas(beta_schedule, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000):
    # scheduling_utils_flax.py:270-300 ; defaults from training_utils.py:223-230
    T = num_train_timesteps
    if beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, T, dtype=F32)
    elif beta_schedule in ("scaled_linear", "zero_snr_scaled_linear"):
        betas = (np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=F32) ** 2).astype(F32)
        if beta_schedule == "zero_snr_scaled_linear":
            betas = rescale_betas(betas)
    elif beta_schedule == "squaredcos_cap_v2":
        betas = betas_for_alpha_bar(T)
    else:
        raise NotImplementedError(f"beta_schedule {beta_schedule} is not implemented")
    return betas


def create_state(beta_schedule, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000):
    """scheduling_utils_flax.py:302-313 -> dict(alphas, betas, alphas_cumprod) float32[T]."""
    betas = make_betas(beta_schedule, beta_start, beta_end, num_train_timesteps)
    alphas = (F32(1.0) - betas).astype(F32)
    alphas_cumprod = np.cumprod(alphas, dtype=F32)
    return {"alphas": alphas, "betas": betas, "alphas_cumprod": alphas_cumprod}


def _coeffs(state, timesteps, ndim):
    # scheduling_utils_flax.py:316-329 (broadcast from the left over (B,C,H,W))
    ac = state["alphas_cumprod"][np.asarray(timesteps)]
    sa = (ac ** F32(0.5)).astype(F32)
    so = ((F32(1) - ac) ** F32(0.5)).astype(F32)
    shape = (-1,) + (1,) * (ndim - 1)
    return sa.reshape(shape), so.reshape(shape)


def add_noise(state, original_samples, noise, timesteps):
    # scheduling_utils_flax.py:332-337
    sa, so = _coeffs(state, timesteps, original_samples.ndim)
    return (sa * original_samples.astype(F32) + so * noise.astype(F32)).astype(F32)


def get_velocity(state, sample, noise, timesteps):
    # scheduling_utils_flax.py:340-343
    sa, so = _coeffs(state, timesteps, sample.ndim)
    return (sa * noise.astype(F32) - so * sample.astype(F32)).astype(F32)


def min_snr_weight(state, timesteps, gamma, prediction_type):
    # training_utils.py:531-568 ; returns float32 (B,) weights (broadcast (B,1,1,1) by caller)
    ac = state["alphas_cumprod"]
    with np.errstate(divide="ignore", invalid="ignore"):
        snrs = (ac / (F32(1) - ac)).astype(F32)
        snr = snrs[np.asarray(timesteps)]
        m = np.minimum(snr, F32(gamma))
        if prediction_type == "v_prediction":
            w = m / (snr + F32(1))
        else:
            w = m / snr
    return w.astype(F32)


# ----------------------------------------------------------------------------- DDIM (sampling path)
# The reference's own schedulers/scheduling_ddim_flax.py: create_state :127-147 (final_alpha_cumprod = 1 when set_alpha_to_one),
# set_timesteps :165-186 (evenly spaced (arange(n) * (T // n))[::-1] + steps_offset), step :199-284 (DDIM eq. 12 with eta = 0,
# the three prediction types, no sample clipping).  The reference builds it at training_utils.py:998-1004 and steps it from
# models/pipeline_flax_stable_diffusion.py:218-232.


def ddim_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=0):
    ratio = num_train_timesteps // num_inference_steps
    return ((np.arange(0, num_inference_steps) * ratio).round()[::-1] + steps_offset).astype(np.int32)


def ddim_step(state, model_output, timestep, sample, num_inference_steps, prediction_type="epsilon",
              num_train_timesteps=1000, set_alpha_to_one=True):
    """One deterministic (eta = 0) DDIM update x_t -> x_{t - T//n}; float32 arrays of the sample's shape."""
    ac = state["alphas_cumprod"]
    prev = int(timestep) - num_train_timesteps // num_inference_steps
    a_t = F32(ac[int(timestep)])
    a_prev = F32(ac[prev]) if prev >= 0 else (F32(1.0) if set_alpha_to_one else F32(ac[0]))
    b_t = F32(1) - a_t
    x, m = sample.astype(F32), model_output.astype(F32)
    if prediction_type == "epsilon":
        x0 = (x - b_t ** F32(0.5) * m) / a_t ** F32(0.5)
        eps = m
    elif prediction_type == "sample":
        x0 = m
        eps = (x - a_t ** F32(0.5) * x0) / b_t ** F32(0.5)
    elif prediction_type == "v_prediction":
        x0 = a_t ** F32(0.5) * x - b_t ** F32(0.5) * m
        eps = a_t ** F32(0.5) * m + b_t ** F32(0.5) * x
    else:
        raise ValueError(f"prediction_type {prediction_type}")
    return (a_prev ** F32(0.5) * x0 + (F32(1) - a_prev) ** F32(0.5) * eps).astype(F32)
