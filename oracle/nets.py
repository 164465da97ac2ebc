"""Oracle (test infrastructure): fp32 PyTorch-CPU restatement of the networks train_step calls.

The arithmetic lives in third-party packages that are NOT under /root/reference and
not installed here; it is restated from their published algorithm and anchored on the
reference's call sites:
  * diffusers==0.21.4 (requirements.txt:1) models/unet_2d_condition_flax.py,
    unet_2d_blocks_flax.py, attention_flax.py (as patched by key_chunk_patch.patch:1-9:
    key_chunk_size == query length -> exact softmax attention), resnet_flax.py,
    embeddings_flax.py                      -> call site training_utils.py:678-684
  * diffusers==0.21.4 models/vae_flax.py   -> call site training_utils.py:574-586
  * transformers FlaxCLIPTextModel          -> call site training_utils.py:635-640
Parameters are flat dicts keyed by the Flax module path ('/'-joined) in Flax layouts:
conv kernel HWIO, Dense kernel [in,out], norms scale/bias.
Activations are NHWC like Flax.  Two precisions:
  * default: everything float32 (the fp32 "truth"; the bf16 HIP path is held to it with the
    loose tolerances written in each test);
  * ``with bf16_points():`` values are rounded to bfloat16 (round-to-nearest-even, forward
    AND cotangent) at the points where the reference's modules hold bf16 tensors - all
    three models are built with dtype=jnp.bfloat16 over float32 parameters
    (training_utils.py:209-222): flax nn.Conv / nn.Dense cast inputs, kernel and bias to
    bf16, produce a bf16 result and add the bias in bf16; GroupNorm / LayerNorm take
    statistics in float32 and return bf16; activations, residual adds and the attention
    core (scaled q, logits, exponentials, weighted values) are bf16 tensors.  Inside a
    contraction the accumulation stays float32 (as on the MXU and in the MFMA kernels).
    This is the mode the HIP path is gated against tightly; it is still a restatement
    (which intermediate XLA keeps in f32 inside a fusion is not observable here).
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- configs


def unet_config(name="sd15", **over):
    """diffusers UNet2DConditionModel config dicts for the BASELINE.json model families."""
    base = dict(
        in_channels=4, out_channels=4, layers_per_block=2, flip_sin_to_cos=True, freq_shift=0,
        norm_num_groups=32, use_linear_projection=False, transformer_layers_per_block=1,
        addition_embed_type=None, addition_time_embed_dim=None,
        projection_class_embeddings_input_dim=None,
    )
    if name == "sd15":
        base.update(
            down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
            up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
            block_out_channels=(320, 640, 1280, 1280), attention_head_dim=8, cross_attention_dim=768)
    elif name == "sd21":
        base.update(
            down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
            up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
            block_out_channels=(320, 640, 1280, 1280), attention_head_dim=(5, 10, 20, 20),
            cross_attention_dim=1024, use_linear_projection=True)
    elif name == "sdxl":
        base.update(
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
            block_out_channels=(320, 640, 1280), attention_head_dim=(5, 10, 20),
            cross_attention_dim=2048, use_linear_projection=True,
            transformer_layers_per_block=(1, 2, 10), addition_embed_type="text_time",
            addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)
    elif name == "tiny":
        # small config with every structural feature of sd15 (for fast parity tests)
        base.update(
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
            block_out_channels=(32, 64), attention_head_dim=2, cross_attention_dim=48,
            layers_per_block=1)
    else:
        raise ValueError(name)
    base.update(over)
    return base


def vae_config(name="sd"):
    if name == "sd":
        return dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                    layers_per_block=2, norm_num_groups=32)
    if name == "tiny":
        return dict(in_channels=3, latent_channels=4, block_out_channels=(32, 32, 64, 64),
                    layers_per_block=1, norm_num_groups=32)
    raise ValueError(name)


def clip_config(name="clip_l"):
    if name == "clip_l":
        return dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                    num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                    layer_norm_eps=1e-5)
    if name == "openclip_h":  # SD2.1 text tower (OpenCLIP ViT-H/14, penultimate-layer export: 23 layers)
        return dict(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                    num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5)
    if name == "openclip_bigg":  # SDXL second text tower
        return dict(vocab_size=49408, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32,
                    num_attention_heads=20, max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5)
    if name == "tiny":
        return dict(vocab_size=1000, hidden_size=48, intermediate_size=96, num_hidden_layers=2,
                    num_attention_heads=3, max_position_embeddings=77, hidden_act="quick_gelu",
                    layer_norm_eps=1e-5)
    raise ValueError(name)


def dual_clip_config(first="clip_l", second="openclip_bigg"):
    """SDXL: two text towers whose hidden states are concatenated on the feature axis (build-side extension: the reference's
    train_step holds one text encoder and cannot drive SDXL, SURVEY.md §8(d) note)."""
    return dict(towers=[clip_config(first), clip_config(second)], prefixes=["text_encoder/", "text_encoder_2/"])


def _per_block(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


# ----------------------------------------------------------------------------- parameter shapes


def _conv(shapes, p, cin, cout, k=3):
    shapes[p + "/kernel"] = (k, k, cin, cout)
    shapes[p + "/bias"] = (cout,)


def _dense(shapes, p, cin, cout, bias=True):
    shapes[p + "/kernel"] = (cin, cout)
    if bias:
        shapes[p + "/bias"] = (cout,)


def _norm(shapes, p, c):
    shapes[p + "/scale"] = (c,)
    shapes[p + "/bias"] = (c,)


def _resnet_shapes(shapes, p, cin, cout, temb_ch):
    _norm(shapes, p + "/norm1", cin)
    _conv(shapes, p + "/conv1", cin, cout)
    if temb_ch:
        _dense(shapes, p + "/time_emb_proj", temb_ch, cout)
    _norm(shapes, p + "/norm2", cout)
    _conv(shapes, p + "/conv2", cout, cout)
    if cin != cout:
        _conv(shapes, p + "/conv_shortcut", cin, cout, k=1)


def _transformer_shapes(shapes, p, c, ctx_dim, depth, linear_proj):
    _norm(shapes, p + "/norm", c)
    if linear_proj:
        _dense(shapes, p + "/proj_in", c, c)
        _dense(shapes, p + "/proj_out", c, c)
    else:
        _conv(shapes, p + "/proj_in", c, c, k=1)
        _conv(shapes, p + "/proj_out", c, c, k=1)
    for k in range(depth):
        b = f"{p}/transformer_blocks_{k}"
        for a, kd in (("attn1", c), ("attn2", ctx_dim)):
            _dense(shapes, f"{b}/{a}/to_q", c, c, bias=False)
            _dense(shapes, f"{b}/{a}/to_k", kd, c, bias=False)
            _dense(shapes, f"{b}/{a}/to_v", kd, c, bias=False)
            _dense(shapes, f"{b}/{a}/to_out_0", c, c)
        _dense(shapes, f"{b}/ff/net_0/proj", c, c * 8)
        _dense(shapes, f"{b}/ff/net_2", c * 4, c)
        for n in ("norm1", "norm2", "norm3"):
            _norm(shapes, f"{b}/{n}", c)


def unet_param_shapes(cfg):
    """Flax param tree of diffusers 0.21.4 FlaxUNet2DConditionModel (names: SURVEY.md §8(b)4)."""
    s = {}
    boc = cfg["block_out_channels"]
    nb = len(boc)
    temb = boc[0] * 4
    depth = _per_block(cfg["transformer_layers_per_block"], nb)
    lpb = cfg["layers_per_block"]
    ctx = cfg["cross_attention_dim"]
    lin = cfg["use_linear_projection"]
    _conv(s, "conv_in", cfg["in_channels"], boc[0])
    _dense(s, "time_embedding/linear_1", boc[0], temb)
    _dense(s, "time_embedding/linear_2", temb, temb)
    if cfg["addition_embed_type"] == "text_time":
        _dense(s, "add_embedding/linear_1", cfg["projection_class_embeddings_input_dim"], temb)
        _dense(s, "add_embedding/linear_2", temb, temb)
    out_ch = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        in_ch, out_ch = out_ch, boc[i]
        for j in range(lpb):
            _resnet_shapes(s, f"down_blocks_{i}/resnets_{j}", in_ch if j == 0 else out_ch, out_ch, temb)
            if t == "CrossAttnDownBlock2D":
                _transformer_shapes(s, f"down_blocks_{i}/attentions_{j}", out_ch, ctx, depth[i], lin)
        if i != nb - 1:
            _conv(s, f"down_blocks_{i}/downsamplers_0/conv", out_ch, out_ch)
    mid = boc[-1]
    _resnet_shapes(s, "mid_block/resnets_0", mid, mid, temb)
    _transformer_shapes(s, "mid_block/attentions_0", mid, ctx, depth[-1], lin)
    _resnet_shapes(s, "mid_block/resnets_1", mid, mid, temb)
    rev = list(reversed(boc))
    rdepth = list(reversed(depth))
    out_ch = rev[0]
    for i, t in enumerate(cfg["up_block_types"]):
        prev, out_ch = out_ch, rev[i]
        in_ch = rev[min(i + 1, nb - 1)]
        for j in range(lpb + 1):
            skip = in_ch if j == lpb else out_ch
            rin = prev if j == 0 else out_ch
            _resnet_shapes(s, f"up_blocks_{i}/resnets_{j}", rin + skip, out_ch, temb)
            if t == "CrossAttnUpBlock2D":
                _transformer_shapes(s, f"up_blocks_{i}/attentions_{j}", out_ch, ctx, rdepth[i], lin)
        if i != nb - 1:
            _conv(s, f"up_blocks_{i}/upsamplers_0/conv", out_ch, out_ch)
    _norm(s, "conv_norm_out", boc[0])
    _conv(s, "conv_out", boc[0], cfg["out_channels"])
    return s


def vae_encoder_param_shapes(cfg):
    """Encoder half (+quant_conv) of diffusers 0.21.4 FlaxAutoencoderKL."""
    s = {}
    boc = cfg["block_out_channels"]
    _conv(s, "encoder/conv_in", cfg["in_channels"], boc[0])
    out_ch = boc[0]
    for i in range(len(boc)):
        in_ch, out_ch = out_ch, boc[i]
        for j in range(cfg["layers_per_block"]):
            _resnet_shapes(s, f"encoder/down_blocks_{i}/resnets_{j}", in_ch if j == 0 else out_ch, out_ch, 0)
        if i != len(boc) - 1:
            _conv(s, f"encoder/down_blocks_{i}/downsamplers_0/conv", out_ch, out_ch)
    c = boc[-1]
    _resnet_shapes(s, "encoder/mid_block/resnets_0", c, c, 0)
    a = "encoder/mid_block/attentions_0"
    _norm(s, a + "/group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        _dense(s, f"{a}/{n}", c, c)
    _resnet_shapes(s, "encoder/mid_block/resnets_1", c, c, 0)
    _norm(s, "encoder/conv_norm_out", c)
    _conv(s, "encoder/conv_out", c, 2 * cfg["latent_channels"])
    _conv(s, "quant_conv", 2 * cfg["latent_channels"], 2 * cfg["latent_channels"], k=1)
    return s


def vae_decoder_param_shapes(cfg):
    """Decoder half (+post_quant_conv) of diffusers 0.21.4 FlaxAutoencoderKL (vae_flax.py FlaxDecoder: conv_in, mid block,
    up blocks over reversed block_out_channels with layers_per_block + 1 resnets each and an upsampler on all but the last,
    conv_norm_out, conv_out); used by the sampling path only (pipeline_flax_stable_diffusion.py:245-249)."""
    s = {}
    boc = list(cfg["block_out_channels"])[::-1]
    lc = cfg["latent_channels"]
    _conv(s, "post_quant_conv", lc, lc, k=1)
    _conv(s, "decoder/conv_in", lc, boc[0])
    c = boc[0]
    _resnet_shapes(s, "decoder/mid_block/resnets_0", c, c, 0)
    a = "decoder/mid_block/attentions_0"
    _norm(s, a + "/group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        _dense(s, f"{a}/{n}", c, c)
    _resnet_shapes(s, "decoder/mid_block/resnets_1", c, c, 0)
    out_ch = boc[0]
    for i in range(len(boc)):
        in_ch, out_ch = out_ch, boc[i]
        for j in range(cfg["layers_per_block"] + 1):
            _resnet_shapes(s, f"decoder/up_blocks_{i}/resnets_{j}", in_ch if j == 0 else out_ch, out_ch, 0)
        if i != len(boc) - 1:
            _conv(s, f"decoder/up_blocks_{i}/upsamplers_0/conv", out_ch, out_ch)
    _norm(s, "decoder/conv_norm_out", boc[-1])
    _conv(s, "decoder/conv_out", boc[-1], cfg["in_channels"])
    return s


def clip_param_shapes(cfg, prefix=""):
    """transformers FlaxCLIPTextModel param tree (text_model/...)."""
    if "towers" in cfg:
        out = {}
        for i, c in enumerate(cfg["towers"]):
            out.update(clip_param_shapes(c, cfg["prefixes"][i]))
        return out
    s = {}
    d, f = cfg["hidden_size"], cfg["intermediate_size"]
    s[prefix + "text_model/embeddings/token_embedding/embedding"] = (cfg["vocab_size"], d)
    s[prefix + "text_model/embeddings/position_embedding/embedding"] = (cfg["max_position_embeddings"], d)
    for i in range(cfg["num_hidden_layers"]):
        b = f"{prefix}text_model/encoder/layers/{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _dense(s, f"{b}/self_attn/{n}", d, d)
        _norm(s, f"{b}/layer_norm1", d)
        _norm(s, f"{b}/layer_norm2", d)
        _dense(s, f"{b}/mlp/fc1", d, f)
        _dense(s, f"{b}/mlp/fc2", f, d)
    _norm(s, prefix + "text_model/final_layer_norm", d)
    return s


def init_params(shapes, seed=0, dtype=torch.float32):
    """Deterministic synthetic weights: fan-in scaled normal for kernels and
    embeddings, scale=1 / bias=0 for norms, small normal bias elsewhere."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shp = shapes[k]
        leaf = k.rsplit("/", 1)[1]
        if leaf == "kernel":
            fan_in = 1
            for d in shp[:-1]:
                fan_in *= d
            out[k] = torch.randn(shp, generator=g, dtype=dtype) / math.sqrt(fan_in)
        elif leaf == "embedding":
            out[k] = torch.randn(shp, generator=g, dtype=dtype) * 0.02
        elif leaf == "scale":
            out[k] = 1.0 + 0.1 * torch.randn(shp, generator=g, dtype=dtype)
        else:
            out[k] = 0.02 * torch.randn(shp, generator=g, dtype=dtype)
    return out


# ----------------------------------------------------------------------------- bf16 rounding points

_BF16_POINTS = False


class bf16_points:
    """Context manager switching the oracle to the reference's bf16 module semantics (see the module docstring)."""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global _BF16_POINTS
        self.prev, _BF16_POINTS = _BF16_POINTS, self.on
        return self

    def __exit__(self, *exc):
        global _BF16_POINTS
        _BF16_POINTS = self.prev


class _RoundBF16(torch.autograd.Function):
    """x.astype(bfloat16) held in float32 storage; the cotangent of a dtype conversion is the converted cotangent."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _r(x):
    return _RoundBF16.apply(x) if _BF16_POINTS else x


# ----------------------------------------------------------------------------- primitives (NHWC)


def conv2d(x, p, name, stride=1, pad=1):
    """flax nn.Conv, NHWC input, HWIO kernel. pad: int or ((top,bottom),(left,right))."""
    w = _r(p[name + "/kernel"]).permute(3, 2, 0, 1)
    xc = _r(x).permute(0, 3, 1, 2)
    if not isinstance(pad, int):
        (pt, pb), (pl, pr) = pad
        xc = F.pad(xc, (pl, pr, pt, pb))
        pad = 0
    if not _BF16_POINTS:
        return F.conv2d(xc, w, p[name + "/bias"], stride=stride, padding=pad).permute(0, 2, 3, 1)
    y = _r(F.conv2d(xc, w, None, stride=stride, padding=pad)).permute(0, 2, 3, 1)  # bf16 result of the contraction ...
    return _r(y + _r(p[name + "/bias"]))                                              # ... then the bias add, in bf16


def dense(x, p, name):
    y = _r(_r(x) @ _r(p[name + "/kernel"]))
    b = p.get(name + "/bias")
    return y if b is None else _r(y + _r(b))


def group_norm(x, p, name, groups, eps):
    """flax nn.GroupNorm over NHWC (stats per (n, group) over HW x C/groups), fp32."""
    n, h, w, c = x.shape
    xg = x.reshape(n, h * w, groups, c // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 3), keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + eps)).reshape(n, h, w, c)
    return _r(y * p[name + "/scale"] + p[name + "/bias"])  # statistics and affine in f32, result in the module dtype


def layer_norm(x, p, name, eps=1e-5):
    return _r(F.layer_norm(x, (x.shape[-1],), p[name + "/scale"], p[name + "/bias"], eps))


def silu(x):
    return _r(x * _r(torch.sigmoid(x)))


def gelu_tanh(x):
    # flax nn.gelu default approximate=True
    return _r(0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3))))


def attention_core(q, k, v, heads, scale, causal=False, key_logit_bias=None):
    """softmax(q k^T * scale + key_logit_bias) v per (batch, head).  With key_chunk_patch.patch the key chunk of
    jax_memory_efficient_attention spans every key for self-attention (exact softmax); for the text keys of cross-attention
    the chunking can count keys twice - key_logit_bias = ln(key_chunk_weights) carries that.
    q: (B,Nq,C) k,v: (B,Nk,C); key_logit_bias (Nk,) or None."""
    b, nq, c = q.shape
    nk = k.shape[1]
    d = c // heads
    qh = q.reshape(b, nq, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(b, nk, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(b, nk, heads, d).permute(0, 2, 1, 3)
    s = _r(_r(qh * scale) @ kh.transpose(-1, -2))
    if key_logit_bias is not None:
        s = s + key_logit_bias
    if causal:
        m = torch.full((nq, nk), float("-inf")).triu(1)
        s = s + m
    if not _BF16_POINTS:
        o = torch.softmax(s, dim=-1) @ vh
    else:  # the memory-efficient attention keeps un-normalised bf16 exponentials and divides the bf16 sums at the end
        e = _r(torch.exp(s - s.max(dim=-1, keepdim=True).values.detach()))
        o = _r(_r(e @ vh) / _r(e.sum(dim=-1, keepdim=True)))
    return o.permute(0, 2, 1, 3).reshape(b, nq, c)


# ----------------------------------------------------------------------------- UNet


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0):
    """diffusers embeddings_flax.get_sinusoidal_embeddings."""
    half = dim // 2
    inc = math.log(10000.0) / (half - freq_shift)
    inv = torch.exp(torch.arange(half, dtype=torch.float32) * -inc)
    e = t.to(torch.float32)[:, None] * inv[None]
    return _r(torch.cat([torch.cos(e), torch.sin(e)], -1) if flip_sin_to_cos else torch.cat([torch.sin(e), torch.cos(e)], -1))


def resnet_block(x, temb, p, name, groups=32, eps=1e-5):
    """diffusers FlaxResnetBlock2D. temb None for the VAE variant."""
    h = conv2d(silu(group_norm(x, p, name + "/norm1", groups, eps)), p, name + "/conv1")
    if temb is not None:
        h = _r(h + dense(silu(temb), p, name + "/time_emb_proj")[:, None, None, :])
    h = conv2d(silu(group_norm(h, p, name + "/norm2", groups, eps)), p, name + "/conv2")
    if name + "/conv_shortcut/kernel" in p:
        x = conv2d(x, p, name + "/conv_shortcut", pad=0)
    return _r(h + x)


def key_chunk_weights(n_query, num_kv):
    """SURVEY.md §8 a9c.  diffusers 0.21.4 attention_flax.py, as patched by key_chunk_patch.patch:5-6 (key_chunk_size =
    flatten_latent_dim = the query count): _query_chunk_attention maps over jnp.arange(0, num_kv, key_chunk_size) with
    key_chunk_size = min(key_chunk_size, num_kv) and takes every chunk with jax.lax.dynamic_slice, which clamps a start index so
    that the slice fits.  When the chunk does not divide num_kv the last chunk therefore starts at num_kv - chunk and overlaps the
    previous one; chunk results are merged by summing exp-weights, so the overlapped keys count twice.  Returns the (num_kv,)
    multiplicities (all ones for self-attention and whenever n_query >= num_kv or the chunk divides num_kv).  Restated from the
    published third-party source and JAX's documented clamping; not captured from a run (jax is not installed)."""
    c = min(n_query, num_kv)
    w = torch.zeros(num_kv)
    for start in range(0, num_kv, c):
        s0 = min(start, num_kv - c)
        w[s0: s0 + c] += 1
    return w


def _attn(x, ctx, p, name, heads, chunked_keys=True):
    c = x.shape[-1]
    q = dense(x, p, name + "/to_q")
    k = dense(ctx, p, name + "/to_k")
    v = dense(ctx, p, name + "/to_v")
    bias = None
    if chunked_keys:
        w = key_chunk_weights(x.shape[1], ctx.shape[1])
        if not bool((w == 1).all()):
            bias = torch.log(w)
    o = attention_core(q, k, v, heads, (c // heads) ** -0.5, key_logit_bias=bias)
    return dense(o, p, name + "/to_out_0")


def transformer_2d(x, ctx, p, name, heads, depth, linear_proj, groups=32, chunked_keys=True):
    """diffusers FlaxTransformer2DModel + FlaxBasicTransformerBlock;
    GroupNorm eps 1e-5 (flax default), LayerNorm eps 1e-5, GEGLU with tanh GELU."""
    n, hh, ww, c = x.shape
    res = x
    h = group_norm(x, p, name + "/norm", groups, 1e-5)
    if linear_proj:
        h = dense(h.reshape(n, hh * ww, c), p, name + "/proj_in")
    else:
        h = conv2d(h, p, name + "/proj_in", pad=0).reshape(n, hh * ww, c)
    for k in range(depth):
        b = f"{name}/transformer_blocks_{k}"
        hn = layer_norm(h, p, b + "/norm1")
        h = _r(h + _attn(hn, hn, p, b + "/attn1", heads))
        h = _r(h + _attn(layer_norm(h, p, b + "/norm2"), ctx, p, b + "/attn2", heads, chunked_keys))
        f = dense(layer_norm(h, p, b + "/norm3"), p, b + "/ff/net_0/proj")
        lin, gate = f.chunk(2, dim=-1)
        h = _r(h + dense(_r(lin * gelu_tanh(gate)), p, b + "/ff/net_2"))
    if linear_proj:
        h = dense(h, p, name + "/proj_out").reshape(n, hh, ww, c)
    else:
        h = conv2d(h.reshape(n, hh, ww, c), p, name + "/proj_out", pad=0)
    return _r(h + res)


def upsample_nearest2x(x):
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def unet_forward(p, cfg, sample_nchw, timesteps, ctx, added_cond=None):
    """diffusers 0.21.4 FlaxUNet2DConditionModel.__call__; returns NCHW sample."""
    boc = cfg["block_out_channels"]
    nb = len(boc)
    heads = _per_block(cfg["attention_head_dim"], nb)  # Flax: attention_head_dim == head COUNT
    depth = _per_block(cfg["transformer_layers_per_block"], nb)
    lpb = cfg["layers_per_block"]
    lin = cfg["use_linear_projection"]
    g = cfg["norm_num_groups"]
    ck = cfg.get("emulate_key_chunks", True)  # a9c: the patched key chunking counts overlapped keys twice (key_chunk_weights)
    t_emb = timestep_embedding(timesteps, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    t_emb = dense(silu(dense(t_emb, p, "time_embedding/linear_1")), p, "time_embedding/linear_2")
    if cfg["addition_embed_type"] == "text_time":
        tid = added_cond["time_ids"]
        te = timestep_embedding(tid.flatten(), cfg["addition_time_embed_dim"], True, 0).reshape(tid.shape[0], -1)
        a = torch.cat([added_cond["text_embeds"], te], -1)
        t_emb = _r(t_emb + dense(silu(dense(a, p, "add_embedding/linear_1")), p, "add_embedding/linear_2"))
    x = conv2d(sample_nchw.permute(0, 2, 3, 1), p, "conv_in")
    skips = [x]
    for i, t in enumerate(cfg["down_block_types"]):
        for j in range(lpb):
            x = resnet_block(x, t_emb, p, f"down_blocks_{i}/resnets_{j}", g)
            if t == "CrossAttnDownBlock2D":
                x = transformer_2d(x, ctx, p, f"down_blocks_{i}/attentions_{j}", heads[i], depth[i], lin, g, ck)
            skips.append(x)
        if i != nb - 1:
            x = conv2d(x, p, f"down_blocks_{i}/downsamplers_0/conv", stride=2, pad=1)
            skips.append(x)
    x = resnet_block(x, t_emb, p, "mid_block/resnets_0", g)
    x = transformer_2d(x, ctx, p, "mid_block/attentions_0", heads[-1], depth[-1], lin, g, ck)
    x = resnet_block(x, t_emb, p, "mid_block/resnets_1", g)
    rheads, rdepth = list(reversed(heads)), list(reversed(depth))
    for i, t in enumerate(cfg["up_block_types"]):
        for j in range(lpb + 1):
            x = torch.cat([x, skips.pop()], dim=-1)
            x = resnet_block(x, t_emb, p, f"up_blocks_{i}/resnets_{j}", g)
            if t == "CrossAttnUpBlock2D":
                x = transformer_2d(x, ctx, p, f"up_blocks_{i}/attentions_{j}", rheads[i], rdepth[i], lin, g, ck)
        if i != nb - 1:
            x = conv2d(upsample_nearest2x(x), p, f"up_blocks_{i}/upsamplers_0/conv")
    assert not skips
    x = conv2d(silu(group_norm(x, p, "conv_norm_out", g, 1e-5)), p, "conv_out")
    return x.permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------- VAE encoder


def vae_encode_moments(p, cfg, pixel_nchw):
    """diffusers 0.21.4 FlaxAutoencoderKL.encode up to the moments (B,h,w,2*latent), NHWC
    (SURVEY.md §8(c) 'VAE encoder'; GroupNorm eps 1e-6 in the VAE)."""
    g = cfg["norm_num_groups"]
    boc = cfg["block_out_channels"]
    x = conv2d(pixel_nchw.permute(0, 2, 3, 1), p, "encoder/conv_in")
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            x = resnet_block(x, None, p, f"encoder/down_blocks_{i}/resnets_{j}", g, 1e-6)
        if i != len(boc) - 1:
            x = conv2d(x, p, f"encoder/down_blocks_{i}/downsamplers_0/conv", stride=2, pad=((0, 1), (0, 1)))
    x = resnet_block(x, None, p, "encoder/mid_block/resnets_0", g, 1e-6)
    a = "encoder/mid_block/attentions_0"
    n, hh, ww, c = x.shape
    h = group_norm(x, p, a + "/group_norm", g, 1e-6).reshape(n, hh * ww, c)
    q, k, v = dense(h, p, a + "/query"), dense(h, p, a + "/key"), dense(h, p, a + "/value")
    # single head; q and k each scaled by C^(-1/4)  ==  logits scaled by C^(-1/2)
    o = attention_core(q, k, v, 1, c ** -0.5)
    x = _r(x + dense(o, p, a + "/proj_attn").reshape(n, hh, ww, c))
    x = resnet_block(x, None, p, "encoder/mid_block/resnets_1", g, 1e-6)
    x = conv2d(silu(group_norm(x, p, "encoder/conv_norm_out", g, 1e-6)), p, "encoder/conv_out")
    return conv2d(x, p, "quant_conv", pad=0)


def vae_decode(p, cfg, latents_nhwc):
    """diffusers 0.21.4 FlaxAutoencoderKL.decode: latents (B,h,w,latent) NHWC -> image (B,8h,8w,3) NHWC."""
    g = cfg["norm_num_groups"]
    boc = list(cfg["block_out_channels"])[::-1]
    x = conv2d(conv2d(latents_nhwc, p, "post_quant_conv", pad=0), p, "decoder/conv_in")
    x = resnet_block(x, None, p, "decoder/mid_block/resnets_0", g, 1e-6)
    a = "decoder/mid_block/attentions_0"
    n, hh, ww, c = x.shape
    h = group_norm(x, p, a + "/group_norm", g, 1e-6).reshape(n, hh * ww, c)
    o = attention_core(dense(h, p, a + "/query"), dense(h, p, a + "/key"), dense(h, p, a + "/value"), 1, c ** -0.5)
    x = _r(x + dense(o, p, a + "/proj_attn").reshape(n, hh, ww, c))
    x = resnet_block(x, None, p, "decoder/mid_block/resnets_1", g, 1e-6)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet_block(x, None, p, f"decoder/up_blocks_{i}/resnets_{j}", g, 1e-6)
        if i != len(boc) - 1:
            x = conv2d(upsample_nearest2x(x), p, f"decoder/up_blocks_{i}/upsamplers_0/conv")
    return conv2d(silu(group_norm(x, p, "decoder/conv_norm_out", g, 1e-6)), p, "decoder/conv_out")


def vae_sample_latents(moments_nhwc, eps_nhwc, scale=0.18215):
    """FlaxDiagonalGaussianDistribution.sample + training_utils.py:582-586 -> NCHW latents."""
    mean, logvar = moments_nhwc.chunk(2, dim=-1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    return ((mean + std * eps_nhwc) * scale).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------- CLIP text


def clip_text_forward(p, cfg, input_ids, prefix=""):
    """transformers FlaxCLIPTextModel(...)[0]: last_hidden_state after final_layer_norm.
    Causal mask; the batch's attention_mask is not passed (training_utils.py:635-640).
    Two-tower configs (dual_clip_config): input_ids (B, 2, 77) -> features concatenated."""
    if "towers" in cfg:
        return torch.cat([clip_text_forward(p, c, input_ids[:, i], cfg["prefixes"][i]) for i, c in enumerate(cfg["towers"])], -1)
    b, s = input_ids.shape
    d = cfg["hidden_size"]
    heads = cfg["num_attention_heads"]
    eps = cfg["layer_norm_eps"]
    x = _r(p[prefix + "text_model/embeddings/token_embedding/embedding"])[input_ids.long()]
    x = _r(x + _r(p[prefix + "text_model/embeddings/position_embedding/embedding"])[:s][None])
    for i in range(cfg["num_hidden_layers"]):
        L = f"{prefix}text_model/encoder/layers/{i}"
        h = layer_norm(x, p, L + "/layer_norm1", eps)
        q, k, v = (dense(h, p, f"{L}/self_attn/{n}") for n in ("q_proj", "k_proj", "v_proj"))
        o = attention_core(q, k, v, heads, (d // heads) ** -0.5, causal=True)
        x = _r(x + dense(o, p, L + "/self_attn/out_proj"))
        h = dense(layer_norm(x, p, L + "/layer_norm2", eps), p, L + "/mlp/fc1")
        if cfg["hidden_act"] == "quick_gelu":
            h = _r(h * _r(torch.sigmoid(1.702 * h)))
        else:
            h = _r(F.gelu(h))
        x = _r(x + dense(h, p, L + "/mlp/fc2"))
    return layer_norm(x, p, prefix + "text_model/final_layer_norm", eps)


def assemble_context(hs, batch, strip_bos_eos):
    """training_utils.py:643-673: (B*k,77,D) -> (B,k,77,D) -> concat (227 tokens for k=3,
    152 for k=1 with strip: the single chunk appears twice)."""
    d = hs.shape[-1]
    e = hs.reshape(batch, -1, 77, d)
    if strip_bos_eos:
        return torch.cat([e[:, 0, :-1, :], e[:, 1:-1, 1:-1, :].reshape(batch, -1, d), e[:, -1, 1:, :]], dim=1)
    return e.reshape(batch, -1, d)
