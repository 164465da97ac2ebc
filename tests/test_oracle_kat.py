"""Pins of the CPU oracle: the hand-derived known-answer values of SURVEY.md §8(c) (the reference itself ships
no tests or golden vectors - "parity unpinned"), exact published parameter counts, and internal consistency."""
import math

import numpy as np
import torch

from oracle import lion8, nets, schedulers


def test_scaled_linear_table():
    a = schedulers.create_state("scaled_linear")["alphas_cumprod"]
    assert a.dtype == np.float32 and a.shape == (1000,)
    np.testing.assert_allclose(a[[0, 500, 999]], [0.99915, 0.27633256, 0.0046600895], rtol=2e-6)
    np.testing.assert_allclose(np.sqrt(a[500]), 0.5256735, rtol=1e-6)
    np.testing.assert_allclose(np.sqrt(1 - a[500]), 0.8506864, rtol=1e-6)


def test_zero_snr_table():
    s = schedulers.create_state("zero_snr_scaled_linear")
    a = s["alphas_cumprod"]
    np.testing.assert_allclose(a[[0, 250, 500, 750]], [0.99915, 0.6524570, 0.2410188, 0.0327991], rtol=2e-6)
    np.testing.assert_allclose(a[998], 1.96789e-07, rtol=1e-4)
    assert a[999] == 0.0 and s["betas"][999] == 1.0


def test_other_schedules():
    lin = schedulers.create_state("linear", 0.0001, 0.02)["betas"]
    assert lin[0] == np.float32(0.0001) and lin[-1] == np.float32(0.02)
    cos = schedulers.create_state("squaredcos_cap_v2")["betas"]
    assert cos.max() <= np.float32(0.999) and cos[0] < 1e-3


def test_add_noise_and_velocity_identities():
    st = schedulers.create_state("scaled_linear")
    rng = np.random.default_rng(0)
    x0, e = rng.standard_normal((3, 4, 8, 8), dtype=np.float32), rng.standard_normal((3, 4, 8, 8), dtype=np.float32)
    t = np.array([0, 500, 999])
    n = schedulers.add_noise(st, x0, e, t)
    v = schedulers.get_velocity(st, x0, e, t)
    sa, so = np.sqrt(st["alphas_cumprod"][t])[:, None, None, None], np.sqrt(1 - st["alphas_cumprod"][t])[:, None, None, None]
    np.testing.assert_allclose(sa * n - so * v, x0, atol=2e-6)  # x0 = sqrt(a) x_t - sqrt(1-a) v
    np.testing.assert_allclose(so * n + sa * v, e, atol=2e-6)


def test_lion8_codec_kat():
    x = np.array([0, 1, -1, .5, -.5, .1, -.1, .01, 1e-3, 1e-5, -1e-5, 1e-9, -1e-8], np.float32)
    assert lion8.quantize(x).tolist() == [3, 127, -127, 111, -111, 80, -80, 51, 32, 13, -13, 3, -3]
    np.testing.assert_allclose(lion8.dequantize(np.array([3, 111, 80, -3], np.int8)),
                               [3.615185e-09, 0.5100307, 0.09918164, -1.1094985e-08], rtol=1e-6)


def test_lion8_block_kat():
    blk = np.array([0.02, -0.5, 0.25, 0, 1e-4, -1e-4, 0.125, -0.0625, 0.3, 0.4, -0.45, 0.05, 0.001, -0.002, 0.49, -0.01], np.float32)
    codes, inv = lion8.block_quantize(blk, 16)
    assert inv.item() == 2.0
    assert codes.ravel().tolist() == [67, -127, 111, 3, 23, -23, 96, -84, 115, 121, -124, 80, 37, -42, 126, -58]
    assert abs(np.abs(lion8.block_dequantize(blk.shape, codes, inv) - blk).max() - 0.00937748) < 1e-7


def test_lion8_init_state_is_code_3():
    st = lion8.init_state({"a/kernel": np.zeros((4, 16), np.float32), "a/bias": np.zeros(4, np.float32)},
                          {"a/kernel": True, "a/bias": False}, 16)
    codes, inv = st["mu"]["a/kernel"]
    assert (codes == 3).all() and (inv == 1).all() and st["mu"]["a/bias"].dtype == np.float32


def test_lion_step_matches_fp32_formula_on_unquantised_leaf():
    rng = np.random.default_rng(1)
    p = {"w": rng.standard_normal(64).astype(np.float32)}
    g = {"w": (rng.standard_normal(64) * 0.01).astype(np.float32)}
    st = lion8.init_state(p, None, 16)
    newp, st2, gn = lion8.lion_step(p, g, st, lr=1e-3, wd=0.1)
    u = np.sign(np.float32(0.1) * g["w"])
    np.testing.assert_allclose(newp["w"], p["w"] - np.float32(1e-3) * (u + np.float32(0.1) * p["w"]), rtol=1e-6)
    np.testing.assert_allclose(st2["mu"]["w"], np.float32(1 - 0.99) * g["w"], rtol=1e-6)
    assert gn < 1.0


def test_clip_by_global_norm():
    g = {"a": np.full(4, 3.0, np.float32), "b": np.full(9, 4.0, np.float32)}
    c, n = lion8.clip_by_global_norm(g, 1.0)
    np.testing.assert_allclose(n, math.sqrt(4 * 9 + 9 * 16), rtol=1e-6)
    np.testing.assert_allclose(math.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in c.values())), 1.0, rtol=1e-6)
    small = {"a": np.full(4, 0.1, np.float32)}
    c2, _ = lion8.clip_by_global_norm(small, 1.0)
    assert (c2["a"] == small["a"]).all()


def test_create_mask_exact_component_match():
    m = lion8.create_mask(["conv_in/kernel", "conv_in/bias", "a/time_embedding/linear_1/kernel", "x/time_emb_proj/kernel",
                           "d/conv_input/kernel", "norm/scale"],
                          ["bias", "scale", "conv_in", "time_embedding", "time_emb_proj"])
    assert m == {"conv_in/kernel": False, "conv_in/bias": False, "a/time_embedding/linear_1/kernel": False,
                 "x/time_emb_proj/kernel": False, "d/conv_input/kernel": True, "norm/scale": False}


def test_parameter_counts_match_published_sizes():
    for name, exp in (("sd15", 859_520_964), ("sd21", 865_910_724), ("sdxl", 2_567_463_684)):
        s = nets.unet_param_shapes(nets.unet_config(name))
        assert sum(math.prod(v) for v in s.values()) == exp
    assert sum(math.prod(v) for v in nets.vae_encoder_param_shapes(nets.vae_config("sd")).values()) == 34_163_664
    assert sum(math.prod(v) for v in nets.clip_param_shapes(nets.clip_config()).values()) == 123_060_480


def test_context_lengths():
    hs = torch.zeros(2 * 3, 77, 8)
    assert nets.assemble_context(hs, 2, True).shape[1] == 227
    assert nets.assemble_context(hs, 2, False).shape[1] == 231
    hs1 = torch.arange(2 * 77 * 8, dtype=torch.float32).reshape(2, 77, 8)
    c = nets.assemble_context(hs1, 2, True)
    assert c.shape[1] == 152  # the single chunk appears twice: [:-1] and [1:]
    assert torch.equal(c[:, :76], hs1[:, :-1]) and torch.equal(c[:, 76:], hs1[:, 1:])
    assert nets.assemble_context(hs1, 2, False).shape[1] == 77


def test_unet_every_leaf_used_and_output_shape():
    cfg = nets.unet_config("tiny")
    p = nets.init_params(nets.unet_param_shapes(cfg), 0)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    out = nets.unet_forward(p, cfg, torch.randn(2, 4, 8, 8), torch.tensor([3, 700]), torch.randn(2, 77, 48))
    assert out.shape == (2, 4, 8, 8)
    grads = torch.autograd.grad(out.square().mean(), list(p.values()), allow_unused=True)
    assert all(g is not None for g in grads), "a parameter leaf is not reached by unet_forward"


def test_vae_and_clip_shapes():
    vc, cc = nets.vae_config("tiny"), nets.clip_config("tiny")
    vp = nets.init_params(nets.vae_encoder_param_shapes(vc), 1)
    m = nets.vae_encode_moments(vp, vc, torch.rand(1, 3, 32, 32))
    assert m.shape == (1, 4, 4, 8)
    lat = nets.vae_sample_latents(m, torch.zeros(1, 4, 4, 4))
    assert lat.shape == (1, 4, 4, 4) and torch.allclose(lat, m[..., :4].permute(0, 3, 1, 2) * 0.18215)
    cp = nets.init_params(nets.clip_param_shapes(cc), 2)
    hs = nets.clip_text_forward(cp, cc, torch.randint(0, 1000, (2, 77)))
    assert hs.shape == (2, 77, 48)
    # causal: changing a later token must not change earlier positions
    ids = torch.randint(0, 1000, (1, 77))
    ids2 = ids.clone()
    ids2[0, 50] = (ids2[0, 50] + 1) % 1000
    a, b = nets.clip_text_forward(cp, cc, ids), nets.clip_text_forward(cp, cc, ids2)
    assert torch.allclose(a[0, :50], b[0, :50], atol=1e-6) and not torch.allclose(a[0, 50:], b[0, 50:])


def test_timestep_embedding_kat():
    e = nets.timestep_embedding(torch.tensor([0, 10]), 320)
    assert torch.allclose(e[0, :160], torch.ones(160)) and torch.allclose(e[0, 160:], torch.zeros(160))
    inv1 = math.exp(-math.log(10000.0) / 160)
    assert abs(e[1, 1].item() - math.cos(10 * inv1)) < 1e-6 and abs(e[1, 161].item() - math.sin(10 * inv1)) < 1e-6


def test_ddim_timesteps_and_step_identities():
    """Sampling path.  diffusers' FlaxDDIMScheduler is not vendored: these are properties the published
    algorithm must satisfy (DDIM eq. 12, eta = 0), not outputs captured from diffusers."""
    from oracle import schedulers as s
    ts = s.ddim_timesteps(50)
    assert ts.dtype == np.int32 and ts[0] == 980 and ts[-1] == 0 and len(ts) == 50 and np.all(np.diff(ts) == -20)
    assert s.ddim_timesteps(20, steps_offset=1)[0] == 951 and s.ddim_timesteps(20, steps_offset=1)[-1] == 1
    st = s.create_state("scaled_linear")
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((2, 4, 8, 8)).astype(np.float32)
    eps = rng.standard_normal((2, 4, 8, 8)).astype(np.float32)
    t, n = 600, 50
    xt = s.add_noise(st, x0, eps, np.array([t, t]))
    prev = t - 1000 // n
    want = s.add_noise(st, x0, eps, np.array([prev, prev]))  # the exact model lands on the same (x0, eps) pair one step earlier
    v = s.get_velocity(st, x0, eps, np.array([t, t]))
    for ptype, out in (("epsilon", eps), ("sample", x0), ("v_prediction", v)):
        got = s.ddim_step(st, out, t, xt, n, ptype)
        assert got.dtype == np.float32 and np.allclose(got, want, atol=2e-5), ptype
    # past the last step alpha_prod_prev = 1 (set_alpha_to_one): the update returns x0 itself
    x_last = s.add_noise(st, x0, eps, np.array([0, 0]))
    assert np.allclose(s.ddim_step(st, eps, 0, x_last, n, "epsilon"), x0, atol=2e-5)
    a0 = st["alphas_cumprod"][0]
    keep = s.ddim_step(st, eps, 0, x_last, n, "epsilon", set_alpha_to_one=False)
    assert np.allclose(keep, np.sqrt(a0) * x0 + np.sqrt(1 - a0) * eps, atol=2e-5)


def test_vae_decoder_shapes_and_published_size():
    from oracle import nets as on
    shapes = on.vae_decoder_param_shapes(on.vae_config("sd"))
    n = sum(int(np.prod(v)) for v in shapes.values())
    enc = sum(int(np.prod(v)) for v in on.vae_encoder_param_shapes(on.vae_config("sd")).values())
    assert n + enc == 83_653_863  # AutoencoderKL of SD1.x: 83.65 M parameters (encoder 34.16 M + decoder 49.49 M + quant convs)
    cfg = on.vae_config("tiny")
    p = on.init_params(on.vae_decoder_param_shapes(cfg), 3)
    y = on.vae_decode(p, cfg, torch.randn(2, 4, 6, 4))
    assert tuple(y.shape) == (2, 32, 48, 3)


def test_key_chunk_weights_equal_the_chunked_algorithm():
    """a9c: a literal restatement of the patched key chunking (chunks at arange(0, num_kv, c) with c = min(n_query, num_kv), each taken
    by a CLAMPED slice like jax.lax.dynamic_slice, merged with the running-max weights of _query_chunk_attention) equals one softmax
    with ln(multiplicity) added to the logits - the closed form the oracle and the HIP kernels use."""
    from oracle import nets as on
    assert on.key_chunk_weights(64, 77).tolist() == [1.0] * 13 + [2.0] * 51 + [1.0] * 13     # SD1.5 mid block at 512x512
    w227 = on.key_chunk_weights(64, 227)
    assert [i for i in range(227) if w227[i] == 2] == list(range(163, 192))                 # k = 3 caption windows
    for nq, nk in ((4096, 77), (256, 77), (81, 77), (77, 77), (64, 128), (1024, 1024)):
        assert bool((on.key_chunk_weights(nq, nk) == 1).all()), (nq, nk)
    g = torch.Generator().manual_seed(0)
    for nq, nk in ((64, 77), (16, 77), (64, 227), (144, 231)):
        d = 8
        q, k, v = torch.randn(nq, d, generator=g), torch.randn(nk, d, generator=g), torch.randn(nk, d, generator=g)
        c = min(nq, nk)
        vals, wts, mxs = [], [], []
        for start in range(0, nk, c):
            s0 = min(start, nk - c)                      # dynamic_slice clamps the start index
            s = (q / d ** 0.5) @ k[s0: s0 + c].T
            m = s.max(-1, keepdim=True).values
            e = torch.exp(s - m)
            vals.append(e @ v[s0: s0 + c]); wts.append(e.sum(-1, keepdim=True)); mxs.append(m)
        gm = torch.stack(mxs).max(0).values
        num = sum(val * torch.exp(m - gm) for val, m in zip(vals, mxs))
        den = sum(wt * torch.exp(m - gm) for wt, m in zip(wts, mxs))
        chunked = num / den
        closed = on.attention_core(q[None], k[None], v[None], 1, d ** -0.5, key_logit_bias=torch.log(on.key_chunk_weights(nq, nk)))[0]
        assert torch.allclose(chunked, closed, atol=1e-5), (nq, nk)


def test_bf16_points_mode_rounds_where_the_reference_holds_bf16():
    """oracle.nets.bf16_points: module outputs are bf16-representable, the mode changes results at the bf16 noise level (not at
    the fp32 one, not grossly), forward and backward, and leaving the context restores the fp32 oracle bit for bit."""
    import torch
    shapes = {}
    nets._resnet_shapes(shapes, "r", 64, 32, 48)
    w = nets.init_params(shapes, 5)
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(2, 8, 8, 64, generator=g), torch.randn(2, 48, generator=g)
    y0 = nets.resnet_block(x, t, w, "r")
    with nets.bf16_points():
        xr = x.clone().requires_grad_(True)
        y1 = nets.resnet_block(xr, t, w, "r")
        (gx,) = torch.autograd.grad(y1, xr, torch.ones_like(y1))
    assert torch.equal(y1, y1.to(torch.bfloat16).to(torch.float32)), "block output is not a bf16 value"
    x0 = x.clone().requires_grad_(True)
    (g0,) = torch.autograd.grad(nets.resnet_block(x0, t, w, "r"), x0, torch.ones_like(y0))
    eg = float((gx - g0).norm() / g0.norm())
    assert 5e-4 < eg < 3e-2, eg  # cotangents pass through the same rounding points
    e = float((y1 - y0).norm() / y0.norm())
    assert 5e-4 < e < 2e-2, e
    assert torch.equal(nets.resnet_block(x, t, w, "r"), y0)
