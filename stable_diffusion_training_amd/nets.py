"""The three networks train_step runs, expressed over the HIP operators of ops.py.

  * UNet2DCondition  - diffusers 0.21.4 FlaxUNet2DConditionModel.__call__ (training_utils.py:678-684)
  * VAE encoder      - diffusers 0.21.4 FlaxAutoencoderKL.encode (training_utils.py:574-579), frozen / no grad
  * VAE decoder      - FlaxAutoencoderKL.decode, sampling path only (models/pipeline_flax_stable_diffusion.py:245-249)
  * CLIP text model  - transformers FlaxCLIPTextModel (training_utils.py:635-640), trained

Parameter trees use the diffusers-Flax names and layouts so `create_mask` patterns and
checkpoints stay drop-in.  `*_spec(cfg)` return the leaves as an ordered (path, shape) list in forward execution
order: the flat gradient buffer is laid out in that order, so backward completes all-reduce buckets back to front.
Activations are NHWC bf16 with channels padded to a multiple of 8 (4-channel latents / 3-channel pixels -> 8).
"""
import math
import os

import torch

from . import _lib, ops

# ----------------------------------------------------------------------------- configs (diffusers config.json keys)
_UNET_DEFAULTS = dict(in_channels=4, out_channels=4, layers_per_block=2, flip_sin_to_cos=True, freq_shift=0,
                      norm_num_groups=32, use_linear_projection=False, transformer_layers_per_block=1,
                      addition_embed_type=None, addition_time_embed_dim=None, projection_class_embeddings_input_dim=None)

UNET_CONFIGS = {
    "sd15": dict(down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                 up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                 block_out_channels=(320, 640, 1280, 1280), attention_head_dim=8, cross_attention_dim=768),
    "sd21": dict(down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                 up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                 block_out_channels=(320, 640, 1280, 1280), attention_head_dim=(5, 10, 20, 20),
                 cross_attention_dim=1024, use_linear_projection=True),
    "sdxl": dict(down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                 up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                 block_out_channels=(320, 640, 1280), attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
                 use_linear_projection=True, transformer_layers_per_block=(1, 2, 10), addition_embed_type="text_time",
                 addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816),
    "tiny": dict(down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
                 block_out_channels=(32, 64), attention_head_dim=2, cross_attention_dim=48, layers_per_block=1),
}
VAE_CONFIGS = {
    "sd": dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32),
    "tiny": dict(in_channels=3, latent_channels=4, block_out_channels=(32, 32, 64, 64), layers_per_block=1, norm_num_groups=32),
}
CLIP_CONFIGS = {
    "clip_l": dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                   max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5),
    # SD2.1's text tower (OpenCLIP ViT-H/14 text model, penultimate-layer export: 23 layers) and SDXL's second tower (bigG, 32 layers)
    "openclip_h": dict(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                       max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5),
    "openclip_bigg": dict(vocab_size=49408, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                          max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5),
    "tiny": dict(vocab_size=1000, hidden_size=48, intermediate_size=96, num_hidden_layers=2, num_attention_heads=3,
                 max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5),
}


def unet_config(name="sd15", **over):
    cfg = dict(_UNET_DEFAULTS)
    cfg.update(UNET_CONFIGS[name])
    cfg.update(over)
    return cfg


def vae_config(name="sd"):
    return dict(VAE_CONFIGS[name])


def clip_config(name="clip_l"):
    return dict(CLIP_CONFIGS[name])


def _per_block(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _pad8(c):
    return (c + 7) // 8 * 8


_GROUP_SHARED = os.environ.get("SDT_GROUP_SHARED", "1") != "0"  # developer A/B: 0 = every block projects temb / the context itself


# ----------------------------------------------------------------------------- parameter specs (forward order)
class _Spec(list):
    def __init__(self, *a):
        super().__init__(*a)
        self.groups = {}  # key -> [index in the list where the group goes, [(kernel leaves), ...], [(bias leaves), ...]]

    def _defer(self, key, kernels, biases=()):
        g = self.groups.setdefault(key, [len(self), [], []])
        g[1] += kernels
        g[2] += biases

    def finish(self):
        """The leaf list.  Dense layers that consume the SAME tensor in many blocks - the time-embedding projection of every
        ResBlock (input silu(temb)) and the cross-attention to_k / to_v of every transformer block (input: the text context) -
        are placed back to back per output width, kernels then biases, at the position of the group's first member, so that
        ops.linear_multi runs each width as ONE GEMM per pass (SD1.5: 22 + 32 projections, 162 launches per step -> 18).  A
        group's gradients complete when the backward reaches its first member, which is where the group sits in the buffers,
        so the exchange buckets still complete in buffer order.  Names and shapes are the diffusers ones; only positions move."""
        out = list(self)
        for at, kernels, biases in sorted(self.groups.values(), key=lambda g: -g[0]):
            out[at:at] = kernels + biases
        return out

    def conv(self, p, cin, cout, k=3):
        self.append((p + "/kernel", (k, k, cin, cout)))
        self.append((p + "/bias", (cout,)))

    def dense(self, p, cin, cout, bias=True):
        self.append((p + "/kernel", (cin, cout)))
        if bias:
            self.append((p + "/bias", (cout,)))

    def norm(self, p, c):
        self.append((p + "/scale", (c,)))
        self.append((p + "/bias", (c,)))

    def resnet(self, p, cin, cout, temb):
        self.norm(p + "/norm1", cin)
        self.conv(p + "/conv1", cin, cout)
        if temb:  # placed next to the other projections of the same width (finish())
            q = p + "/time_emb_proj"
            self._defer(("temb", cout), [(q + "/kernel", (temb, cout))], [(q + "/bias", (cout,))])
        self.norm(p + "/norm2", cout)
        self.conv(p + "/conv2", cout, cout)
        if cin != cout:
            self.conv(p + "/conv_shortcut", cin, cout, k=1)

    def transformer(self, p, c, ctx, depth, lin):
        self.norm(p + "/norm", c)
        if lin:
            self.dense(p + "/proj_in", c, c)
        else:
            self.conv(p + "/proj_in", c, c, k=1)
        for k in range(depth):
            b = f"{p}/transformer_blocks_{k}"
            self.norm(b + "/norm1", c)
            for n, kd in (("to_q", c), ("to_k", c), ("to_v", c)):
                self.dense(f"{b}/attn1/{n}", kd, c, bias=False)
            self.dense(f"{b}/attn1/to_out_0", c, c)
            self.norm(b + "/norm2", c)
            self.dense(f"{b}/attn2/to_q", c, c, bias=False)
            self._defer(("kv", c), [(f"{b}/attn2/to_k/kernel", (ctx, c)), (f"{b}/attn2/to_v/kernel", (ctx, c))])
            self.dense(f"{b}/attn2/to_out_0", c, c)
            self.norm(b + "/norm3", c)
            self.dense(f"{b}/ff/net_0/proj", c, 8 * c)
            self.dense(f"{b}/ff/net_2", 4 * c, c)
        if lin:
            self.dense(p + "/proj_out", c, c)
        else:
            self.conv(p + "/proj_out", c, c, k=1)


def unet_spec(cfg):
    s = _Spec()
    boc = cfg["block_out_channels"]
    nb, lpb, ctx, lin = len(boc), cfg["layers_per_block"], cfg["cross_attention_dim"], cfg["use_linear_projection"]
    temb = boc[0] * 4
    depth = _per_block(cfg["transformer_layers_per_block"], nb)
    s.dense("time_embedding/linear_1", boc[0], temb)
    s.dense("time_embedding/linear_2", temb, temb)
    if cfg["addition_embed_type"] == "text_time":
        s.dense("add_embedding/linear_1", cfg["projection_class_embeddings_input_dim"], temb)
        s.dense("add_embedding/linear_2", temb, temb)
    s.conv("conv_in", cfg["in_channels"], boc[0])
    out_ch = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        in_ch, out_ch = out_ch, boc[i]
        for j in range(lpb):
            s.resnet(f"down_blocks_{i}/resnets_{j}", in_ch if j == 0 else out_ch, out_ch, temb)
            if t == "CrossAttnDownBlock2D":
                s.transformer(f"down_blocks_{i}/attentions_{j}", out_ch, ctx, depth[i], lin)
        if i != nb - 1:
            s.conv(f"down_blocks_{i}/downsamplers_0/conv", out_ch, out_ch)
    mid = boc[-1]
    s.resnet("mid_block/resnets_0", mid, mid, temb)
    s.transformer("mid_block/attentions_0", mid, ctx, depth[-1], lin)
    s.resnet("mid_block/resnets_1", mid, mid, temb)
    rev, rdepth = list(reversed(boc)), list(reversed(depth))
    out_ch = rev[0]
    for i, t in enumerate(cfg["up_block_types"]):
        prev, out_ch = out_ch, rev[i]
        in_ch = rev[min(i + 1, nb - 1)]
        for j in range(lpb + 1):
            skip = in_ch if j == lpb else out_ch
            s.resnet(f"up_blocks_{i}/resnets_{j}", (prev if j == 0 else out_ch) + skip, out_ch, temb)
            if t == "CrossAttnUpBlock2D":
                s.transformer(f"up_blocks_{i}/attentions_{j}", out_ch, ctx, rdepth[i], lin)
        if i != nb - 1:
            s.conv(f"up_blocks_{i}/upsamplers_0/conv", out_ch, out_ch)
    s.norm("conv_norm_out", boc[0])
    s.conv("conv_out", boc[0], cfg["out_channels"])
    return s.finish()


def time_emb_groups(leaves):
    """{width: [resnet path, ...]} in buffer order, from the leaves of a UNet parameter store."""
    groups = {}
    for p in leaves:
        if p.endswith("/time_emb_proj/kernel"):
            groups.setdefault(leaves[p].shape[1], []).append(p[: -len("/time_emb_proj/kernel")])
    return groups


def vae_encoder_spec(cfg):
    s = _Spec()
    boc = cfg["block_out_channels"]
    s.conv("encoder/conv_in", cfg["in_channels"], boc[0])
    out_ch = boc[0]
    for i in range(len(boc)):
        in_ch, out_ch = out_ch, boc[i]
        for j in range(cfg["layers_per_block"]):
            s.resnet(f"encoder/down_blocks_{i}/resnets_{j}", in_ch if j == 0 else out_ch, out_ch, 0)
        if i != len(boc) - 1:
            s.conv(f"encoder/down_blocks_{i}/downsamplers_0/conv", out_ch, out_ch)
    c = boc[-1]
    s.resnet("encoder/mid_block/resnets_0", c, c, 0)
    a = "encoder/mid_block/attentions_0"
    s.norm(a + "/group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        s.dense(f"{a}/{n}", c, c)
    s.resnet("encoder/mid_block/resnets_1", c, c, 0)
    s.norm("encoder/conv_norm_out", c)
    s.conv("encoder/conv_out", c, 2 * cfg["latent_channels"])
    s.conv("quant_conv", 2 * cfg["latent_channels"], 2 * cfg["latent_channels"], k=1)
    return list(s)


def vae_decoder_spec(cfg):
    """Decoder half (+post_quant_conv) of FlaxAutoencoderKL, forward order (sampling path)."""
    s = _Spec()
    boc = tuple(cfg["block_out_channels"])[::-1]
    lc = cfg["latent_channels"]
    s.conv("post_quant_conv", lc, lc, k=1)
    s.conv("decoder/conv_in", lc, boc[0])
    c = boc[0]
    s.resnet("decoder/mid_block/resnets_0", c, c, 0)
    a = "decoder/mid_block/attentions_0"
    s.norm(a + "/group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        s.dense(f"{a}/{n}", c, c)
    s.resnet("decoder/mid_block/resnets_1", c, c, 0)
    out_ch = boc[0]
    for i in range(len(boc)):
        in_ch, out_ch = out_ch, boc[i]
        for j in range(cfg["layers_per_block"] + 1):
            s.resnet(f"decoder/up_blocks_{i}/resnets_{j}", in_ch if j == 0 else out_ch, out_ch, 0)
        if i != len(boc) - 1:
            s.conv(f"decoder/up_blocks_{i}/upsamplers_0/conv", out_ch, out_ch)
    s.norm("decoder/conv_norm_out", boc[-1])
    s.conv("decoder/conv_out", boc[-1], cfg["in_channels"])
    return list(s)


def clip_text_spec(cfg, prefix=""):
    if "towers" in cfg:  # SDXL: two text towers in one parameter store (dual_clip_config)
        return [leaf for i, c in enumerate(cfg["towers"]) for leaf in clip_text_spec(c, cfg["prefixes"][i])]
    s = _Spec()
    d, f = cfg["hidden_size"], cfg["intermediate_size"]
    s.append((prefix + "text_model/embeddings/token_embedding/embedding", (cfg["vocab_size"], d)))
    s.append((prefix + "text_model/embeddings/position_embedding/embedding", (cfg["max_position_embeddings"], d)))
    for i in range(cfg["num_hidden_layers"]):
        b = f"{prefix}text_model/encoder/layers/{i}"
        s.norm(b + "/layer_norm1", d)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s.dense(f"{b}/self_attn/{n}", d, d)
        s.norm(b + "/layer_norm2", d)
        s.dense(b + "/mlp/fc1", d, f)
        s.dense(b + "/mlp/fc2", f, d)
    s.norm(prefix + "text_model/final_layer_norm", d)
    return list(s)


def dual_clip_config(first="clip_l", second="openclip_bigg"):
    """SDXL conditions its UNet on the hidden states of two text towers concatenated along the feature axis (768 + 1280 = 2048 =
    cross_attention_dim).  The reference's train_step holds ONE text-encoder state (training_utils.py:635-640) and cannot drive
    SDXL (SURVEY.md §8(d) note); here both towers live in one parameter store under the diffusers sub-folder names, so the
    step's signature, optimizer sweep and gradient exchange are unchanged."""
    return dict(towers=[clip_config(first), clip_config(second)], prefixes=["text_encoder/", "text_encoder_2/"])


def init_params(spec, seed=0):
    """Synthetic weights (no checkpoints offline): fan-in scaled normal kernels, ~1 norm scales, small biases."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in sorted(spec):
        leaf = k.rsplit("/", 1)[1]
        if leaf == "kernel":
            out[k] = torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[:-1]))
        elif leaf == "embedding":
            out[k] = torch.randn(shp, generator=g) * 0.02
        elif leaf == "scale":
            out[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            out[k] = 0.02 * torch.randn(shp, generator=g)
    return out


# ----------------------------------------------------------------------------- UNet
def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0):
    out = torch.empty(t.shape[0], dim, dtype=torch.bfloat16, device=t.device)
    _lib.call("sdt_timestep_embedding", t.data_ptr(), out.data_ptr(), t.shape[0], dim, int(flip_sin_to_cos), float(freq_shift),
              torch.cuda.current_stream().cuda_stream)
    return out


def _time_emb_projections(st, temb_act):
    """Linear(silu(temb)) of every ResBlock (diffusers FlaxResnetBlock2D.time_emb_proj): one GEMM per channel width where the
    projections of that width are laid out back to back (unet_spec), column slices handed to the blocks; one GEMM per block
    otherwise.  The gradients of the slices are gathered and flow back through ONE input-gradient and ONE weight-gradient GEMM
    per width."""
    groups = time_emb_groups(st.leaves)
    acts = iter(ops.fanout(temb_act, len(groups)))
    out = {}
    for width, names in groups.items():
        a = next(acts)
        y = ops.linear_multi(a, st, tuple(n + "/time_emb_proj" for n in names)) if len(names) > 1 and _GROUP_SHARED else None
        if y is not None:
            for n, rb in zip(names, ops.col_slices(y, len(names))):
                out[n] = rb
        else:
            for n, alias in zip(names, ops.fanout(a, len(names))):
                out[n] = ops.linear(alias, st, n + "/time_emb_proj")
    return out


def _context_projections(st, ctx):
    """to_k / to_v of every cross-attention block applied to the text context (diffusers FlaxAttention: key = to_k(context),
    value = to_v(context)): one GEMM per channel width where those kernels are laid out back to back (unet_spec), the blocks
    reading their [k|v] as a column slice of its output; otherwise each block gets an alias of the context and projects it
    itself.  Returns {"<block>/attn2": _PackedKV or context alias}."""
    groups = {}
    for p in st.leaves:
        if p.endswith("/attn2/to_k/kernel"):
            groups.setdefault(st.leaves[p].shape[1], []).append(p[: -len("/to_k/kernel")])
    acts = iter(ops.fanout(ctx, len(groups)))
    out = {}
    for width, names in groups.items():
        a = next(acts)
        y = ops.linear_multi(a, st, tuple(n + t for n in names for t in ("/to_k", "/to_v"))) if len(names) > 1 and _GROUP_SHARED else None
        if y is not None:
            for n, kv in zip(names, ops.col_slices(y, len(names))):
                out[n] = _PackedKV(kv)
        else:
            for n, alias in zip(names, ops.fanout(a, len(names))):
                out[n] = alias
    return out


def _resnet(x, rb, st, name, groups, eps, xs=None):
    """rb: the block's time-embedding row bias (B, C_out) or None (VAE).  xs: GroupNorm statistics of x when its producer
    accumulated them (ops.conv2d / ops.linear gn_groups=).  Returns (output, statistics of the output for the next GroupNorm, or
    None)."""
    h, x = ops.group_norm(x, st, name + "/norm1", groups, eps, silu=True, skip=True, stats=xs)
    h, hs = ops.conv2d(h, st, name + "/conv1", rowbias=rb, gn_groups=groups)
    h = ops.group_norm(h, st, name + "/norm2", groups, eps, silu=True, stats=hs)
    sc = ops.conv2d(x, st, name + "/conv_shortcut", pad=0) if st.has(name + "/conv_shortcut/kernel") else x
    return ops.conv2d(h, st, name + "/conv2", residual=sc, gn_groups=groups)


_KEY_WEIGHTS = {}


def key_chunk_weights(n_query, num_kv, device):
    """How often diffusers' memory-efficient attention, as the reference patches it (key_chunk_patch.patch: key chunk = query
    count), counts each key: chunks start at 0, c, 2c, ... with c = min(n_query, num_kv), and jax.lax.dynamic_slice clamps the last
    start so that the slice fits - when c does not divide num_kv that chunk overlaps the one before it and the overlapped keys enter
    the softmax sum twice (512x512 mid block, 64 queries x 77 keys -> keys 13..63).  None when every key counts once."""
    key = (n_query, num_kv, str(device))
    if key not in _KEY_WEIGHTS:
        c = min(n_query, num_kv)
        w = torch.zeros(num_kv, dtype=torch.float32)
        for start in range(0, num_kv, c):
            s0 = min(start, num_kv - c)
            w[s0: s0 + c] += 1
        _KEY_WEIGHTS[key] = None if bool((w == 1).all()) else w.to(device)
    return _KEY_WEIGHTS[key]


class _PackedKV:
    """The [k|v] projections (B, Nk, 2C) of one cross-attention block, computed ahead of the blocks (_context_projections)."""

    def __init__(self, kv):
        self.kv = kv
        self.shape = kv.shape


def _attn(x, ctx, st, name, heads, residual, chunked_keys=True):
    """ctx None: self-attention; otherwise one alias of the text context (ops.fanout) or the block's precomputed _PackedKV.
    The projections that share an input run as one GEMM (ops.linear_multi) and attention reads / differentiates the packed
    tensor in place."""
    c = x.shape[-1]
    scale = (c // heads) ** -0.5
    kw = key_chunk_weights(x.shape[1], ctx.shape[1], x.device) if (ctx is not None and chunked_keys) else None
    if isinstance(ctx, _PackedKV):
        o = ops.attention_packed(ops.linear(x, st, name + "/to_q"), ctx.kv, heads, scale, key_weight=kw)
        return ops.linear(o, st, name + "/to_out_0", residual=residual)
    if ctx is None:
        qkv = ops.linear_multi(x, st, (name + "/to_q", name + "/to_k", name + "/to_v"))
        if qkv is not None:
            o = ops.attention_packed(qkv, None, heads, scale)
        else:
            xq, xk, xv = ops.fanout(x, 3)
            o = ops.attention(ops.linear(xq, st, name + "/to_q"), ops.linear(xk, st, name + "/to_k"),
                              ops.linear(xv, st, name + "/to_v"), heads, scale)
    else:
        q = ops.linear(x, st, name + "/to_q")
        kv = ops.linear_multi(ctx, st, (name + "/to_k", name + "/to_v"))
        if kv is not None:
            o = ops.attention_packed(q, kv, heads, scale, key_weight=kw)
        else:
            ck, cv = ops.fanout(ctx, 2)
            o = ops.attention(q, ops.linear(ck, st, name + "/to_k"), ops.linear(cv, st, name + "/to_v"), heads, scale, key_weight=kw)
    return ops.linear(o, st, name + "/to_out_0", residual=residual)


def _transformer(x, ctx, st, name, heads, depth, lin, groups, xs=None, chunked_keys=True):
    """ctx: {"<block>/attn2": alias of the text context or _PackedKV} (_context_projections).  xs / second result: see _resnet."""
    B, H, W, C = x.shape
    h, x = ops.group_norm(x, st, name + "/norm", groups, 1e-5, skip=True, stats=xs)
    if lin:
        h = ops.linear(h.view(B, H * W, C), st, name + "/proj_in")
    else:
        h = ops.conv2d(h, st, name + "/proj_in", pad=0).view(B, H * W, C)
    for k in range(depth):
        b = f"{name}/transformer_blocks_{k}"
        hn, h = ops.layer_norm(h, st, b + "/norm1", skip=True)
        h = _attn(hn, None, st, b + "/attn1", heads, h)
        hn, h = ops.layer_norm(h, st, b + "/norm2", skip=True)
        h = _attn(hn, ctx[b + "/attn2"], st, b + "/attn2", heads, h, chunked_keys)
        hn, h = ops.layer_norm(h, st, b + "/norm3", skip=True)
        h = ops.feed_forward_geglu(hn, st, b + "/ff/net_0/proj", b + "/ff/net_2", residual=h)
    if lin:
        y, ys = ops.linear(h, st, name + "/proj_out", residual=x.view(B, H * W, C), gn_groups=groups)
        return y.view(B, H, W, C), ys
    return ops.conv2d(h.view(B, H, W, C), st, name + "/proj_out", pad=0, residual=x, gn_groups=groups)


def unet_forward(st, cfg, x, timesteps, ctx, added_cond=None):
    """x: (B,h,w,pad8(in_channels)) bf16 NHWC; timesteps int32 (B,); ctx (B,L,cross_dim) bf16.
    Returns (B,h,w,pad8(out_channels)) bf16 NHWC (the reference returns NCHW; see train_step)."""
    boc = cfg["block_out_channels"]
    nb, lpb, lin, g = len(boc), cfg["layers_per_block"], cfg["use_linear_projection"], cfg["norm_num_groups"]
    heads = _per_block(cfg["attention_head_dim"], nb)  # Flax: attention_head_dim is the head COUNT
    depth = _per_block(cfg["transformer_layers_per_block"], nb)
    ck = cfg.get("emulate_key_chunks", True)  # False: exact softmax over the text keys (see key_chunk_weights)
    te = timestep_embedding(timesteps, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"]).requires_grad_(True)
    temb = ops.linear(ops.silu(ops.linear(te, st, "time_embedding/linear_1")), st, "time_embedding/linear_2")
    if cfg["addition_embed_type"] == "text_time":
        tid = added_cond["time_ids"]
        tide = timestep_embedding(tid.reshape(-1).to(torch.int32), cfg["addition_time_embed_dim"], True, 0.0).view(tid.shape[0], -1)
        a = ops.concat_channels(added_cond["text_embeds"].to(torch.bfloat16).contiguous(), tide).requires_grad_(True)
        temb = ops.add(temb, ops.linear(ops.silu(ops.linear(a, st, "add_embedding/linear_1")), st, "add_embedding/linear_2"))
    # every resnet consumes silu(temb) and every cross-attention consumes ctx twice: hand out aliases whose gradients are
    # summed by one launch each instead of a chain of binary adds
    rowbias = _time_emb_projections(st, ops.silu(temb))  # {resnet path: (B, C_out) row bias of its first convolution}
    ctx = _context_projections(st, ctx)  # {attn2 path: the block's [k|v] projections}
    if not x.requires_grad:
        x = x.detach().requires_grad_(True)  # anchors the autograd tape (weights are not autograd leaves)
    # (x, xs): every block output travels with the GroupNorm statistics its producing GEMM accumulated (xs None after a concat)
    x, xs = ops.conv2d(x, st, "conv_in", gn_groups=g)
    skips = [x]
    for i, t in enumerate(cfg["down_block_types"]):
        for j in range(lpb):
            x, xs = _resnet(x, rowbias[f"down_blocks_{i}/resnets_{j}"], st, f"down_blocks_{i}/resnets_{j}", g, 1e-5, xs)
            if t == "CrossAttnDownBlock2D":
                x, xs = _transformer(x, ctx, st, f"down_blocks_{i}/attentions_{j}", heads[i], depth[i], lin, g, xs, ck)
            skips.append(x)
        if i != nb - 1:
            x, xs = ops.conv2d(x, st, f"down_blocks_{i}/downsamplers_0/conv", stride=2, pad=1, gn_groups=g)
            skips.append(x)
    x, xs = _resnet(x, rowbias["mid_block/resnets_0"], st, "mid_block/resnets_0", g, 1e-5, xs)
    x, xs = _transformer(x, ctx, st, "mid_block/attentions_0", heads[-1], depth[-1], lin, g, xs, ck)
    x, xs = _resnet(x, rowbias["mid_block/resnets_1"], st, "mid_block/resnets_1", g, 1e-5, xs)
    rheads, rdepth = list(reversed(heads)), list(reversed(depth))
    for i, t in enumerate(cfg["up_block_types"]):
        for j in range(lpb + 1):
            x = ops.concat_channels(x, skips.pop())  # statistics of a concatenation: the standalone pass
            x, xs = _resnet(x, rowbias[f"up_blocks_{i}/resnets_{j}"], st, f"up_blocks_{i}/resnets_{j}", g, 1e-5, None)
            if t == "CrossAttnUpBlock2D":
                x, xs = _transformer(x, ctx, st, f"up_blocks_{i}/attentions_{j}", rheads[i], rdepth[i], lin, g, xs, ck)
        if i != nb - 1:
            x = ops.conv2d(ops.upsample2x(x), st, f"up_blocks_{i}/upsamplers_0/conv")
            xs = None
    assert not skips
    x = ops.group_norm(x, st, "conv_norm_out", g, 1e-5, silu=True, stats=xs)
    return ops.conv2d(x, st, "conv_out")


# ----------------------------------------------------------------------------- VAE encoder (frozen)
def _vae_attention(x, st, a, groups, xs=None):
    """Single-head attention with head dim C (= 512): scores materialised per image (3 % of the encoder's work)."""
    B, H, W, C = x.shape
    N = H * W
    s = torch.cuda.current_stream().cuda_stream
    h = ops.group_norm(x, st, a + "/group_norm", groups, 1e-6, stats=xs).view(B, N, C)
    q, k, v = (ops.linear(h, st, f"{a}/{n}") for n in ("query", "key", "value"))
    scores = torch.empty(N, N, dtype=torch.bfloat16, device=x.device)
    vt = torch.empty(C, N, dtype=torch.bfloat16, device=x.device)
    o = torch.empty(B, N, C, dtype=torch.bfloat16, device=x.device)
    for b in range(B):
        ops.gemm_nt(q[b], k[b], scores, N, N, C, 1, C, C, 0)
        _lib.call("sdt_softmax_rows_inplace", scores.data_ptr(), N, N, float(C) ** -0.5, s)  # q,k each * C^-1/4
        _lib.call("sdt_transpose_bf16", v[b].data_ptr(), vt.data_ptr(), 1, N, C, s)
        ops.gemm_nt(scores, vt, o[b], N, C, N, 1, N, N, 0)
    y, ys = ops.linear(o, st, a + "/proj_attn", residual=x.view(B, N, C), gn_groups=groups)
    return y.view(B, H, W, C), ys


@torch.no_grad()
def vae_encode_moments(st, cfg, pixels_nhwc):
    """pixels (B,H,W,8) bf16 (3 real channels) -> moments (B,H/8,W/8,2*latent) bf16 NHWC."""
    g, boc = cfg["norm_num_groups"], cfg["block_out_channels"]
    x, xs = ops.conv2d(pixels_nhwc, st, "encoder/conv_in", gn_groups=g)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            x, xs = _resnet(x, None, st, f"encoder/down_blocks_{i}/resnets_{j}", g, 1e-6, xs)
        if i != len(boc) - 1:
            x, xs = ops.conv2d(x, st, f"encoder/down_blocks_{i}/downsamplers_0/conv", stride=2, pad=((0, 1), (0, 1)), gn_groups=g)
    x, xs = _resnet(x, None, st, "encoder/mid_block/resnets_0", g, 1e-6, xs)
    x, xs = _vae_attention(x, st, "encoder/mid_block/attentions_0", g, xs)
    x, xs = _resnet(x, None, st, "encoder/mid_block/resnets_1", g, 1e-6, xs)
    x = ops.group_norm(x, st, "encoder/conv_norm_out", g, 1e-6, silu=True, stats=xs)
    x = ops.conv2d(x, st, "encoder/conv_out")
    return ops.conv2d(x, st, "quant_conv", pad=0)


@torch.no_grad()
def vae_decode(st, cfg, latents_nhwc):
    """latents (B,h,w,pad8(latent)) bf16 (already divided by the scaling factor) -> image (B,8h,8w,pad8(3)) bf16 NHWC:
    diffusers 0.21.4 FlaxAutoencoderKL.decode as models/pipeline_flax_stable_diffusion.py:246-249 calls it."""
    g = cfg["norm_num_groups"]
    boc = tuple(cfg["block_out_channels"])[::-1]
    x = ops.conv2d(latents_nhwc, st, "post_quant_conv", pad=0)
    x, xs = ops.conv2d(x, st, "decoder/conv_in", gn_groups=g)
    x, xs = _resnet(x, None, st, "decoder/mid_block/resnets_0", g, 1e-6, xs)
    x, xs = _vae_attention(x, st, "decoder/mid_block/attentions_0", g, xs)
    x, xs = _resnet(x, None, st, "decoder/mid_block/resnets_1", g, 1e-6, xs)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            x, xs = _resnet(x, None, st, f"decoder/up_blocks_{i}/resnets_{j}", g, 1e-6, xs)
        if i != len(boc) - 1:
            x, xs = ops.conv2d(ops.upsample2x(x), st, f"decoder/up_blocks_{i}/upsamplers_0/conv", gn_groups=g)
    x = ops.group_norm(x, st, "decoder/conv_norm_out", g, 1e-6, silu=True, stats=xs)
    return ops.conv2d(x, st, "decoder/conv_out")


# ----------------------------------------------------------------------------- CLIP text encoder (trained)
def clip_text_forward(st, cfg, input_ids, anchor=None, prefix=""):
    """input_ids int32 (B*k, 77) -> last_hidden_state (B*k, 77, D) bf16 after final_layer_norm (causal mask).
    Two-tower configs (dual_clip_config): input_ids (B*k, 2, 77), one row of ids per tower -> (B*k, 77, D1 + D2)."""
    if "towers" in cfg:
        hs = [clip_text_forward(st, c, input_ids[:, i].contiguous(), anchor, cfg["prefixes"][i]) for i, c in enumerate(cfg["towers"])]
        return ops.concat_channels(hs[0], hs[1])
    Bk, S = input_ids.shape
    d, heads, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    if anchor is None:
        anchor = torch.zeros(1, device=input_ids.device, requires_grad=st.trainable)
    x = ops.embedding(input_ids.contiguous(), st, prefix + "text_model/embeddings/token_embedding/embedding",
                      prefix + "text_model/embeddings/position_embedding/embedding", S, anchor)
    act = ops.quick_gelu if cfg["hidden_act"] == "quick_gelu" else ops.gelu_erf
    for i in range(cfg["num_hidden_layers"]):
        L = f"{prefix}text_model/encoder/layers/{i}"
        h, x = ops.layer_norm(x, st, L + "/layer_norm1", eps, skip=True)
        qkv = ops.linear_multi(h, st, tuple(f"{L}/self_attn/{n}" for n in ("q_proj", "k_proj", "v_proj")))
        if qkv is not None:
            o = ops.attention_packed(qkv, None, heads, (d // heads) ** -0.5, causal=True)
        else:
            q, k, v = (ops.linear(hh, st, f"{L}/self_attn/{n}") for hh, n in zip(ops.fanout(h, 3), ("q_proj", "k_proj", "v_proj")))
            o = ops.attention(q, k, v, heads, (d // heads) ** -0.5, causal=True)
        x = ops.linear(o, st, L + "/self_attn/out_proj", residual=x)
        h, x = ops.layer_norm(x, st, L + "/layer_norm2", eps, skip=True)
        x = ops.linear(act(ops.linear(h, st, L + "/mlp/fc1")), st, L + "/mlp/fc2", residual=x)
    return ops.layer_norm(x, st, prefix + "text_model/final_layer_norm", eps)
