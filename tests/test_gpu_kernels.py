"""Kernel-level parity on a real MI355X: every HIP kernel, called through the C ABI (via the product's ops layer),
against a plain fp32 PyTorch reference of the same op on the same seeded inputs.  Tolerances are bf16-output
tolerances: a bf16 result carries 8 significant bits, so |err| <= ~2^-8 * |ref| per element plus accumulation noise;
the checks use relative L2 error (<= 6e-3 unless stated) plus a loose elementwise bound."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def rnd(shape, dev, seed, scale=1.0, dtype=BF):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


class FakeStore:
    """Minimal ParamStore-like holder for single-op tests."""

    def __init__(self, spec, dev, seed=0, trainable=True):
        from stable_diffusion_training_amd import nets, params
        self.st = params.ParamStore(spec, device=dev, quantise=False, trainable=trainable)
        self.w = nets.init_params(spec, seed)
        for k in self.w:
            if k.endswith("bias"):
                self.w[k] = self.w[k] * 10
        self.st.load(self.w)
        self.st.prepare()


# ------------------------------------------------------------------------------------------------ GEMM / Linear
@pytest.mark.parametrize("M,K,N", [(256, 320, 320), (4096, 320, 2560), (300, 768, 640), (77, 64, 8), (5, 1280, 1280), (1000, 136, 264),
                                   (33000, 72, 136),      # 128-tiles with 32-wide K-steps (three workgroups per CU), ragged K (72 = 2 x 32 + 8), M and N
                                   (16384, 320, 960)])    # the same kernel on the level-0 QKV shape; its input gradient (K = 960) too
def test_linear_fwd_bwd(dev, M, K, N):
    from stable_diffusion_training_amd import ops
    fs = FakeStore([("l/kernel", (K, N)), ("l/bias", (N,))], dev, seed=M)
    x = rnd((M, K), dev, 1).requires_grad_(True)
    res = rnd((M, N), dev, 2)
    y = ops.linear(x, fs.st, "l", residual=res)
    wq = fs.w["l/kernel"].to(dev).to(BF).float()
    ref = x.detach().float() @ wq + fs.w["l/bias"].to(dev) + res.float()
    assert rel_l2(y, ref) < 6e-3
    dy = rnd((M, N), dev, 3)
    y.backward(dy)
    assert rel_l2(x.grad, dy.float() @ wq.t()) < 6e-3
    assert rel_l2(fs.st.g("l/kernel"), x.detach().float().t() @ dy.float()) < 2e-3
    assert rel_l2(fs.st.g("l/bias"), dy.float().sum(0)) < 2e-3


def test_linear_wgrad_is_written_not_accumulated(dev):
    """sdt_gemm_tn_wgrad stores dW and the bias gradient (single writer per element, include/sdt.h): what the gradient buffer
    held before - a previous step, or garbage - does not enter the result, so the buffer needs no zero fill."""
    from stable_diffusion_training_amd import ops
    fs = FakeStore([("l/kernel", (64, 64)), ("l/bias", (64,))], dev)
    x = rnd((128, 64), dev, 1).requires_grad_(True)
    dy = rnd((128, 64), dev, 2)
    fs.st.grad.fill_(float("nan"))
    for _ in range(2):
        ops.linear(x, fs.st, "l").backward(dy)
    assert rel_l2(fs.st.g("l/kernel"), x.detach().float().t() @ dy.float()) < 2e-3
    assert rel_l2(fs.st.g("l/bias"), dy.float().sum(0)) < 2e-3


@pytest.mark.parametrize("M,K,N,taps", [(16384, 320, 320, 1), (4096, 640, 1920, 1), (16384, 320, 2560, 1), (4096, 320, 320, 9), (16384, 64, 64, 9)])
def test_wgrad_split_reduction(dev, M, K, N, taps):
    """Small weights at large M: the reduction over M is split across workgroups, every split publishes its fp32 partial tile
    and the split that arrives last adds them in split order.  Checked: the shape really takes that path, the result against
    fp32 torch, bit-identical results from launch to launch whatever the buffer held (no atomics), arrival counters back at zero."""
    from stable_diffusion_training_amd import _lib, ops
    from stable_diffusion_training_amd._lib import GATHER_FPROP, GATHER_PLAIN, SdtConvGeom
    conv = taps == 9
    if conv:
        side = int(round((M // 4) ** 0.5))
        geom = SdtConvGeom(4, side, side, side, side, 3, 3, 1, 1, 1)
        mode, gp = GATHER_FPROP, _lib.ctypes.addressof(geom)
        fs = FakeStore([("c/kernel", (3, 3, K, N)), ("c/bias", (N,))], dev, seed=3)
        x = rnd((4, side, side, K), dev, 1).requires_grad_(True)
        dy = rnd((4, side, side, N), dev, 2)
        name, op = "c", lambda: ops.conv2d(x, fs.st, "c")
    else:
        geom, mode, gp = None, GATHER_PLAIN, None
        fs = FakeStore([("l/kernel", (K, N)), ("l/bias", (N,))], dev, seed=3)
        x = rnd((M, K), dev, 1).requires_grad_(True)
        dy = rnd((M, N), dev, 2)
        name, op = "l", lambda: ops.linear(x, fs.st, "l")
    need = _lib.load().sdt_gemm_tn_workspace_bytes(M, K, N, taps, 0, mode, gp)
    assert need > 0, "shape no longer splits the reduction"
    outs = []
    for poison in (float("nan"), -3.0, 1e30):
        fs.st.grad.fill_(poison)
        op().backward(dy)
        torch.cuda.synchronize()
        outs.append((fs.st.g(name + "/kernel").clone(), fs.st.g(name + "/bias").clone()))
        cnt = ops._TN_WS[x.device][:65536].view(torch.int32)  # the fixed counter area at the head of the workspace
        assert int(cnt.abs().sum()) == 0, "arrival counters not reset"
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0]), "weight gradient not reproducible"
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][1], outs[2][1]), "bias gradient not reproducible"
    xf, dyf = x.detach().float(), dy.float()
    if conv:
        xr = xf.permute(0, 3, 1, 2).requires_grad_(True)
        w = torch.zeros(N, K, 3, 3, device=dev, requires_grad=True)
        torch.nn.functional.conv2d(xr, w, None, padding=1).backward(dyf.permute(0, 3, 1, 2))
        ref = w.grad.permute(2, 3, 1, 0)  # OIHW -> HWIO
        assert rel_l2(outs[0][0], ref) < 2e-3
        assert rel_l2(outs[0][1], dyf.sum((0, 1, 2))) < 2e-3
    else:
        assert rel_l2(outs[0][0], xf.t() @ dyf) < 2e-3
        assert rel_l2(outs[0][1], dyf.sum(0)) < 2e-3


@pytest.mark.parametrize("M,K,N", [(1024, 5120, 1280), (4096, 5120, 640), (308, 3072, 768), (200, 4096, 72)])
def test_splitk_leaves_workspace_zero(dev, M, K, N):
    """Split-K shapes: every split publishes its fp32 partial tile, the last-arriving split sums them in split order and
    finishes the tile in the same launch, handing the arrival counters back at zero (include/sdt.h contract), so back-to-back
    GEMMs of different shapes can share the workspace; ragged M/N tiles included; bit-identical results from launch to launch."""
    from stable_diffusion_training_amd import _lib, ops
    assert _lib.load().sdt_gemm_nt_workspace_bytes(M, N, K, 1) > 0, "shape no longer takes the split-K path"
    fs = FakeStore([("l/kernel", (K, N)), ("l/bias", (N,))], dev, seed=N)
    wq = fs.w["l/kernel"].to(dev).to(BF).float()
    for it in range(3):
        x = rnd((M, K), dev, 10 + it)
        res = rnd((M, N), dev, 20 + it)
        y = ops.linear(x, fs.st, "l", residual=res)
        y2 = ops.linear(x, fs.st, "l", residual=res)
        ref = x.float() @ wq + fs.w["l/bias"].to(dev) + res.float()
        assert rel_l2(y, ref) < 6e-3
        assert torch.equal(y, y2), "split-K result not reproducible"
        torch.cuda.synchronize()
        ws = ops._SPLITK_WS[x.device]
        assert int(torch.count_nonzero(ws[:65536])) == 0, "arrival counters not reset"  # the fixed counter area


def test_split_reduction_handoff_under_uneven_load(dev):
    """The slab hand-off of the split reductions (write-through stores, drained, then the ticket; the last arriver acquires and reads
    with sc1 loads: the CDNA4 guide's split-K recipe, no release fence) checked the way that guide asks for hand-offs: under UNEVEN load
    - a second stream keeps issuing fills and GEMMs of changing size, so splits of one tile land on busy and idle CUs of different XCDs
    at different times - every word of every result, hundreds of launches: split-K Dense forward, the split halo convolution, a split
    Dense weight gradient and a split 3x3 weight gradient must reproduce their first result bit for bit every time."""
    from stable_diffusion_training_amd import _lib, ops
    assert _lib.load().sdt_gemm_nt_workspace_bytes(1024, 1280, 5120, 1) > 0
    fs = FakeStore([("l/kernel", (5120, 1280)), ("l/bias", (1280,)), ("c/kernel", (3, 3, 1280, 1280)), ("c/bias", (1280,))], dev, seed=11)
    x = rnd((1024, 5120), dev, 1).requires_grad_(True)
    xc = rnd((4, 8, 8, 1280), dev, 2).requires_grad_(True)
    dy, dyc = rnd((1024, 1280), dev, 3), rnd((4, 8, 8, 1280), dev, 4)
    big = torch.empty(1 << 26, device=dev)
    ga, gb = rnd((4096, 4096), dev, 5), rnd((4096, 4096), dev, 6)
    side = torch.cuda.Stream()
    first = None
    for it in range(120):
        with torch.cuda.stream(side):  # the uneven background: 0 - 3 fills of changing length and sometimes a large GEMM
            for j in range(it % 4):
                big[: (1 << 20) * (1 + (7 * it + 3 * j) % 61)].fill_(float(it))
            if it % 3 == 0:
                n = 512 * (1 + it % 8)
                torch.mm(ga[:n], gb)
        x.grad = xc.grad = None
        y = ops.linear(x, fs.st, "l")
        y.backward(dy)
        yc = ops.conv2d(xc, fs.st, "c")
        yc.backward(dyc)
        cur = [y.detach().clone(), x.grad.clone(), fs.st.g("l/kernel").clone(), fs.st.g("l/bias").clone(), yc.detach().clone(), xc.grad.clone(),
               fs.st.g("c/kernel").clone(), fs.st.g("c/bias").clone()]
        if first is None:
            first = cur
            wq = fs.w["l/kernel"].to(dev).to(BF).float()
            assert rel_l2(y, x.detach().float() @ wq + fs.w["l/bias"].to(dev)) < 6e-3
        else:
            for k, (a, b) in enumerate(zip(cur, first)):
                assert torch.equal(a, b), f"launch {it}: result {k} changed ({(a.float() - b.float()).abs().max().item():.3e} max abs)"
    torch.cuda.synchronize()
    assert int(torch.count_nonzero(ops._SPLITK_WS[dev][:65536])) == 0


# ------------------------------------------------------------------------------------------------ Conv
CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 32, 32, 320, 320, 3, 1, 1),
    (1, 16, 16, 640, 320, 3, 1, 1),
    (2, 16, 16, 64, 128, 3, 2, 1),                   # UNet downsample
    (2, 16, 16, 64, 64, 3, 2, ((0, 1), (0, 1))),     # VAE downsample (asymmetric pad, VALID)
    (2, 256, 256, 64, 128, 3, 2, ((0, 1), (0, 1))),  # the same at a VAE level's size: gathered rows through the 32-wide-step 128-tile kernel
    (2, 8, 8, 128, 64, 1, 1, 0),                     # 1x1 shortcut
    (2, 12, 20, 8, 32, 3, 1, 1),                     # padded 4->8 input channels, non-square (eight taps per K-step, 64-tiles)
    (4, 64, 64, 8, 320, 3, 1, 1),                    # the UNet's conv_in at batch 4: eight taps per K-step, 128-tiles, ragged channel tile
    (1, 128, 256, 8, 128, 3, 1, 1),                  # the VAE's conv_in: one channel tile, 256 row tiles
    (2, 16, 16, 320, 8, 3, 1, 1),                    # conv_out style (4 -> 8 padded outputs)
    (3, 8, 8, 2560, 1280, 3, 1, 1),                  # deepest up-block shape (halo kernel: 4 images / tile, ragged group, split-K)
    (2, 64, 64, 320, 320, 3, 1, 1),                  # halo kernel, 4 x 64 tiles, N = 2.5 channel tiles
    (1, 8, 128, 128, 128, 3, 1, 1),                  # halo kernel, two 64-wide tiles per row (VAE-like wide image)
    (2, 32, 32, 640, 1280, 3, 1, 1),                 # halo kernel, 8 x 32 tiles, split over channel chunks
    (1, 16, 16, 1920, 640, 3, 1, 1),                 # halo kernel, 16 x 16 tile = one image
    (2, 36, 28, 64, 64, 3, 1, 1),                    # not tileable by the halo kernel: generic gather path
    (2, 24, 40, 64, 64, 3, 1, 1),                    # halo kernel, 8 x 8 tiles of four images, two of the four image slots empty
    (4, 72, 56, 128, 192, 3, 1, 1),                  # halo kernel, 8 x 8 x 4-image tiles over a 72 x 56 aspect bucket, 63 tiles per image group
    (3, 16, 24, 64, 128, 3, 1, 1),                   # halo kernel, 8 x 8 tiles, ragged image group (3 of 4)
    (1, 32, 48, 128, 64, 3, 1, 1),                   # halo kernel, 16 x 16 tiles, 2 x 3 per image
    (2, 48, 80, 64, 320, 3, 1, 1),                   # halo kernel, 16 x 16 tiles, N = 2.5 channel tiles
    (1, 256, 320, 64, 128, 3, 1, 1),                 # halo kernel, 320 tiles on 256 resident workgroups (second round ragged)
    (3, 128, 192, 64, 192, 3, 1, 1),                 # halo kernel, 576 tiles: three rounds, two channel tiles, 288 per XCD run
    (4, 16, 16, 128, 192, 3, 1, 1),                  # three-tap wgrad kernel: four image rows per 64-pixel chunk
    (16, 8, 8, 64, 72, 3, 1, 1),                     # three-tap wgrad kernel: one image per chunk, ragged channel tile
    (4, 8, 8, 128, 72, 3, 1, 1),                     # three-tap wgrad kernel at its smallest: 256 output pixels = four chunks, unsplit
    (1, 96, 96, 64, 64, 3, 1, 1),                    # three-tap wgrad kernel, width not a power of two (SD2.1-768 level 0): a chunk is 2/3 of a row
    (2, 48, 48, 128, 64, 3, 1, 1),                   # ... 48 wide: a chunk is a row and a third
    (4, 24, 24, 64, 136, 3, 1, 1),                   # ... 24 wide: 2 2/3 rows per chunk, 9 chunks per image, ragged channel tile
    (2, 40, 24, 64, 64, 3, 1, 1),                    # ... 24 wide, 40 tall: images end inside a chunk (960 pixels = 15 chunks per image)
    (8, 12, 12, 64, 64, 3, 1, 1),                    # 12 wide (not a multiple of 8): stays on the nine-tap kernel
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", CONV_CASES)
def test_conv2d_fwd_bwd(dev, B, H, W, Cin, Cout, k, stride, pad):
    from stable_diffusion_training_amd import ops
    cin_l = 4 if Cin == 8 else Cin      # logical channels (test the zero-padded paths)
    cout_l = 4 if Cout == 8 else Cout
    fs = FakeStore([("c/kernel", (k, k, cin_l, cout_l)), ("c/bias", (cout_l,))], dev, seed=Cin + Cout)
    x = rnd((B, H, W, Cin), dev, 1)
    if cin_l != Cin:
        x[..., cin_l:] = 0
    x.requires_grad_(True)
    y = ops.conv2d(x, fs.st, "c", stride=stride, pad=pad)
    wq = fs.w["c/kernel"].to(dev).to(BF).float().permute(3, 2, 0, 1)
    xr = x.detach().float()[..., :cin_l].permute(0, 3, 1, 2).requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    if isinstance(pad, int):
        yr = F.conv2d(xr, wr, fs.w["c/bias"].to(dev), stride=stride, padding=pad)
    else:
        (pt, pb), (pl, pr) = pad
        yr = F.conv2d(F.pad(xr, (pl, pr, pt, pb)), wr, fs.w["c/bias"].to(dev), stride=stride)
    assert y.shape[-1] == Cout and tuple(y.shape[1:3]) == tuple(yr.shape[2:])
    assert rel_l2(y[..., :cout_l], yr.permute(0, 2, 3, 1)) < 6e-3
    if cout_l != Cout:
        assert y[..., cout_l:].abs().max() == 0
    dy = rnd(tuple(y.shape), dev, 3)
    if cout_l != Cout:
        dy[..., cout_l:] = 0
    y.backward(dy)
    br = torch.zeros(cout_l, device=dev, requires_grad=True)
    yr.backward(dy.float()[..., :cout_l].permute(0, 3, 1, 2))
    assert rel_l2(x.grad[..., :cin_l], xr.grad.permute(0, 2, 3, 1)) < 6e-3
    assert rel_l2(fs.st.g("c/kernel"), wr.grad.permute(2, 3, 1, 0)) < 3e-3
    assert rel_l2(fs.st.g("c/bias"), dy.float()[..., :cout_l].sum((0, 1, 2))) < 3e-3
    del br


@pytest.mark.parametrize("B,H,W,Cin,Cout,exact", [
    (8, 64, 64, 320, 320, True),      # 5 chunks, N = 2.5 / 5 channel tiles, 384 / 640 tiles
    (4, 128, 128, 128, 128, True),    # 2 chunks, 256 / 512 tiles (VAE level shape)
    (2, 128, 192, 64, 192, True),     # one chunk, ragged channels at 128
    (64, 24, 40, 64, 64, True),       # 8 x 8 x 4-image tiles, exactly one 64-channel tile
    (2, 64, 64, 320, 320, False),     # few tiles: the planner splits the reduction (differently for the two widths)
    (2, 32, 32, 640, 1280, False),    # split over channel chunks (slab reduction)
    (3, 16, 24, 64, 128, False),      # ragged image group
    (1, 8, 128, 128, 72, False)])     # 64-wide tiles per row, ragged channels
def test_halo_conv_tile_widths_agree_bitwise(dev, monkeypatch, B, H, W, Cin, Cout, exact):
    """conv3x3_halo_kernel<BN = 64> (256 x 64 tiles, one halo buffer, two workgroups per CU) against <BN = 128> (one workgroup per
    CU): every output element is the same chain of MFMAs over (chunk, tap, k) in both, so forward and input gradient must agree
    bit for bit wherever neither splits the reduction (a split changes the summation order: tolerance there), with the fused
    row bias, residual and GroupNorm statistics."""
    from stable_diffusion_training_amd import ops
    fs = FakeStore([("c/kernel", (3, 3, Cin, Cout)), ("c/bias", (Cout,))], dev, seed=Cin)
    x = rnd((B, H, W, Cin), dev, 1).requires_grad_(True)
    rb, res, dy = rnd((B, Cout), dev, 2), rnd((B, H, W, Cout), dev, 3), rnd((B, H, W, Cout), dev, 4)

    def run(bn):
        monkeypatch.setenv("SDT_HALO_BN", str(bn))
        x.grad = None
        gn = 32 if Cout % 32 == 0 else 0
        out = ops.conv2d(x, fs.st, "c", rowbias=rb, residual=res, gn_groups=gn)
        y, stats = out if gn else (out, None)
        y.backward(dy)
        torch.cuda.synchronize()
        return y.detach().clone(), None if stats is None else stats.clone(), x.grad.clone()

    y0, s0, g0 = run(128)
    wq = fs.w["c/kernel"].to(dev).to(BF).float().permute(3, 2, 0, 1)
    ref = F.conv2d(x.detach().float().permute(0, 3, 1, 2), wq, fs.w["c/bias"].to(dev), padding=1).permute(0, 2, 3, 1)
    assert rel_l2(y0, ref + rb.float()[:, None, None, :] + res.float()) < 6e-3
    y1, s1, g1 = run(64)
    if not exact:
        assert rel_l2(y1, y0) < 1e-3 and rel_l2(g1, g0) < 1e-3
    else:
        assert torch.equal(y1, y0), "forward differs between the tile widths"
        assert torch.equal(g1, g0), "input gradient differs between the tile widths"
    if s0 is not None and s1 is not None:
        assert rel_l2(s1.sum(1), s0.sum(1)) < 1e-5  # (partial rows per tile: the two widths cut the channels differently)


def test_conv_rowbias_and_residual(dev):
    from stable_diffusion_training_amd import ops
    B, H, W, C = 3, 8, 8, 64
    fs = FakeStore([("c/kernel", (3, 3, C, C)), ("c/bias", (C,))], dev)
    x = rnd((B, H, W, C), dev, 1).requires_grad_(True)
    rb = rnd((B, C), dev, 2).requires_grad_(True)
    res = rnd((B, H, W, C), dev, 3).requires_grad_(True)
    y = ops.conv2d(x, fs.st, "c", rowbias=rb, residual=res)
    wq = fs.w["c/kernel"].to(dev).to(BF).float().permute(3, 2, 0, 1)
    ref = F.conv2d(x.detach().float().permute(0, 3, 1, 2), wq, fs.w["c/bias"].to(dev), padding=1).permute(0, 2, 3, 1)
    ref = ref + rb.detach().float()[:, None, None, :] + res.detach().float()
    assert rel_l2(y, ref) < 6e-3
    dy = rnd((B, H, W, C), dev, 4)
    y.backward(dy)
    assert rel_l2(rb.grad, dy.float().sum((1, 2))) < 6e-3
    assert torch.equal(res.grad, dy)


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, heads, scale, causal):
    B, Nq, C = q.shape
    Nk = k.shape[1]
    d = C // heads
    qh, kh, vh = (t.view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        s = s + torch.full((Nq, Nk), float("-inf"), device=q.device).triu(1)
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("B,H,Nq,Nk,D,causal", [
    (2, 8, 256, 256, 40, False), (1, 8, 1024, 1024, 40, False), (2, 8, 64, 77, 160, False), (2, 8, 256, 77, 80, False),
    (2, 5, 144, 144, 64, False), (3, 12, 77, 77, 64, True), (1, 3, 77, 77, 16, True), (1, 2, 200, 333, 128, False),
    (1, 8, 4096, 4096, 40, False),
    (1, 5, 9216, 9216, 64, False),  # SD2.1-768 first level (BASELINE configs[3]): the long-sequence self-attention, head dim 64
    (1, 10, 2304, 2304, 64, False), (2, 20, 1024, 1024, 64, False),  # its second level / the SDXL 32x32 level
    (2, 8, 2048, 77, 40, False), (1, 4, 1100, 100, 64, False), (4, 8, 1024, 77, 80, False)])  # few keys: query-split dK/dV pass
def test_attention_fwd_bwd(dev, B, H, Nq, Nk, D, causal):
    from stable_diffusion_training_amd import ops
    C = H * D
    q = rnd((B, Nq, C), dev, 1).requires_grad_(True)
    k = rnd((B, Nk, C), dev, 2).requires_grad_(True)
    v = rnd((B, Nk, C), dev, 3).requires_grad_(True)
    scale = D ** -0.5
    o = ops.attention(q, k, v, H, scale, causal)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    oref = attn_ref(qr, kr, vr, H, scale, causal)
    assert rel_l2(o, oref) < 8e-3
    do = rnd((B, Nq, C), dev, 4)
    o.backward(do)
    oref.backward(do.float())
    assert rel_l2(q.grad, qr.grad) < 1.5e-2
    assert rel_l2(k.grad, kr.grad) < 1.5e-2
    assert rel_l2(v.grad, vr.grad) < 1.5e-2


@pytest.mark.parametrize("B,H,Nq,Nk,D,packed", [(2, 8, 64, 77, 160, False), (2, 8, 64, 227, 160, True), (2, 2, 16, 77, 32, False),
                                                 (1, 8, 144, 231, 80, True), (4, 8, 1024, 77, 80, False), (2, 8, 2048, 77, 40, True)])
def test_attention_key_weights(dev, B, H, Nq, Nk, D, packed):
    """SdtAttnDesc.key_weight: P = softmax(s + ln w) with the multiplicities of the reference's clamped key chunks (w in {1, 2} when
    Nq < Nk does not divide Nk; the two large-Nq shapes get a synthetic weight vector so that the query-split dK/dV pass is covered),
    forward and the three gradients, plain and packed operands."""
    from stable_diffusion_training_amd import nets, ops
    C = H * D
    scale = D ** -0.5
    w = nets.key_chunk_weights(Nq, Nk, dev)
    if w is None:
        w = (1.0 + (torch.arange(Nk, device=dev) % 3 == 1).float()).contiguous()
    else:
        assert set(w.unique().tolist()) == {1.0, 2.0}
    q = rnd((B, Nq, C), dev, 1).requires_grad_(True)
    if packed:
        kv = rnd((B, Nk, 2 * C), dev, 2).requires_grad_(True)
        o = ops.attention_packed(q, kv, H, scale, key_weight=w)
        kvr = kv.detach().float().requires_grad_(True)
        kr, vr = kvr[..., :C], kvr[..., C:]
    else:
        k = rnd((B, Nk, C), dev, 2).requires_grad_(True)
        v = rnd((B, Nk, C), dev, 3).requires_grad_(True)
        o = ops.attention(q, k, v, H, scale, key_weight=w)
        kr, vr = k.detach().float().requires_grad_(True), v.detach().float().requires_grad_(True)
    qr = q.detach().float().requires_grad_(True)
    qh, kh, vh = (t.view(B, -1, H, D).transpose(1, 2) for t in (qr, kr, vr))
    sref = (qh @ kh.transpose(-1, -2)) * scale + torch.log(w)
    oref = (torch.softmax(sref, -1) @ vh).transpose(1, 2).reshape(B, Nq, C)
    assert rel_l2(o, oref) < 8e-3
    plain = (torch.softmax((qh @ kh.transpose(-1, -2)) * scale, -1) @ vh).transpose(1, 2).reshape(B, Nq, C)
    assert rel_l2(o, plain.detach()) > 3e-2  # the weights matter: this is not the exact softmax
    do = rnd((B, Nq, C), dev, 4)
    o.backward(do)
    oref.backward(do.float())
    assert rel_l2(q.grad, qr.grad) < 1.5e-2
    if packed:
        assert rel_l2(kv.grad, kvr.grad) < 1.5e-2
    else:
        assert rel_l2(k.grad, kr.grad) < 1.5e-2 and rel_l2(v.grad, vr.grad) < 1.5e-2
    with pytest.raises(ValueError):
        ops.attention(q, q, q, H, scale, key_weight=w[:5].contiguous())


@pytest.mark.parametrize("D", [64, 40, 80])  # 64: fp32 row sums; 40 / 80: row sum through the ones column of the P.V MFMAs
@pytest.mark.parametrize("spike", [6.0, 1.0, 0.35])
def test_attention_rescale_branch_forced(dev, D, spike):
    """A key spike in a LATER tile exercises both sides of the deferred online-softmax rescale (guide rule 26): spike 6
    raises the row maximum by far more than the 2^8 threshold (rescale taken mid-stream), the smaller ones raise it by
    less (the maximum stays stale and the probabilities exceed 1), each in one 32-row block only."""
    from stable_diffusion_training_amd import ops
    B, H, N = 1, 2, 320
    q = rnd((B, N, H * D), dev, 1)
    k = rnd((B, N, H * D), dev, 2)
    v = rnd((B, N, H * D), dev, 3)
    k[:, 200] = q[:, 7] * spike   # query 7: logit jump in the 4th key tile
    k[:, 290] = q[:, 150] * spike  # query 150 (second 128-row block): jump in the ragged last tile
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    o = ops.attention(q, k, v, H, D ** -0.5)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    oref = attn_ref(qr, kr, vr, H, D ** -0.5, False)
    assert rel_l2(o, oref) < 8e-3
    assert rel_l2(o[:, 7], oref[:, 7]) < 1e-2 and rel_l2(o[:, 150], oref[:, 150]) < 1e-2
    do = rnd((B, N, H * D), dev, 4)
    o.backward(do)
    oref.backward(do.float())
    for g, gr in ((q.grad, qr.grad), (k.grad, kr.grad), (v.grad, vr.grad)):  # the backward consumes the forward's LSE
        assert rel_l2(g, gr) < 1.5e-2


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("B,HW,C,silu", [(2, 64, 320, True), (2, 4096, 320, True), (3, 256, 2560, True), (2, 64, 1920, False), (2, 100, 32, True), (1, 1024, 960, True)])
def test_groupnorm_fwd_bwd(dev, B, HW, C, silu):
    from stable_diffusion_training_amd import ops
    fs = FakeStore([("n/scale", (C,)), ("n/bias", (C,))], dev, seed=C)
    x = (rnd((B, HW, C), dev, 1) * 2 + 0.5).requires_grad_(True)
    y = ops.group_norm(x, fs.st, "n", 32, 1e-5, silu=silu)
    xr = x.detach().float().requires_grad_(True)
    g, b = fs.w["n/scale"].to(dev).requires_grad_(True), fs.w["n/bias"].to(dev).requires_grad_(True)
    yr = F.group_norm(xr.transpose(1, 2), 32, g, b, 1e-5).transpose(1, 2)
    if silu:
        yr = F.silu(yr)
    assert rel_l2(y, yr) < 6e-3
    dy = rnd((B, HW, C), dev, 2)
    y.backward(dy)
    yr.backward(dy.float())
    assert rel_l2(x.grad, xr.grad) < 1e-2
    assert rel_l2(fs.st.g("n/scale"), g.grad) < 5e-3
    assert rel_l2(fs.st.g("n/bias"), b.grad) < 5e-3


@pytest.mark.parametrize("kind,shape", [("group", (4, 4096, 320)), ("group", (3, 256, 2560)), ("group", (4, 64, 1280)),
                                        ("layer", (16384, 320)), ("layer", (308, 768)), ("layer", (1024, 1280))])
def test_norms_are_bitwise_reproducible(dev, kind, shape):
    """No float atomics in the norm family: output, input gradient and the ACCUMULATED parameter gradients
    (many workgroups feed each element) come out bit for bit equal from two runs on the same inputs."""
    from stable_diffusion_training_amd import ops
    C = shape[-1]
    fs = FakeStore([("n/scale", (C,)), ("n/bias", (C,))], dev, seed=C)
    x0 = rnd(shape, dev, 1) * 2 + 0.5
    dy, dsk = rnd(shape, dev, 2), rnd(shape, dev, 3)
    runs = []
    for _ in range(2):
        fs.st.grad.zero_()
        x = x0.clone().requires_grad_(True)
        h = x * 1.0
        if kind == "group":
            y, xs = ops.group_norm(h, fs.st, "n", 32, 1e-5, silu=True, skip=True)
        else:
            y, xs = ops.layer_norm(h, fs.st, "n", skip=True)
        torch.autograd.backward([y, xs], [dy, dsk])
        torch.cuda.synchronize()
        runs.append((y.detach().clone(), x.grad.clone(), fs.st.grad.clone()))
    for a, b, what in zip(runs[0], runs[1], ("output", "input gradient", "parameter gradients")):
        assert torch.equal(a, b), f"{kind} norm {shape}: {what} differ between two identical launches"


@pytest.mark.parametrize("M,C", [(512, 320), (77 * 3, 768), (100, 1280), (64, 48), (33, 2048),
                                 (4096, 640), (2048, 2048)])  # many blocks: parameter gradients through the per-block partials
def test_layernorm_fwd_bwd(dev, M, C):
    from stable_diffusion_training_amd import ops
    fs = FakeStore([("n/scale", (C,)), ("n/bias", (C,))], dev, seed=C)
    x = (rnd((M, C), dev, 1) * 1.5 + 0.3).requires_grad_(True)
    y = ops.layer_norm(x, fs.st, "n", 1e-5)
    xr = x.detach().float().requires_grad_(True)
    g, b = fs.w["n/scale"].to(dev).requires_grad_(True), fs.w["n/bias"].to(dev).requires_grad_(True)
    yr = F.layer_norm(xr, (C,), g, b, 1e-5)
    assert rel_l2(y, yr) < 6e-3
    dy = rnd((M, C), dev, 2)
    y.backward(dy)
    yr.backward(dy.float())
    assert rel_l2(x.grad, xr.grad) < 1e-2
    assert rel_l2(fs.st.g("n/scale"), g.grad) < 5e-3
    assert rel_l2(fs.st.g("n/bias"), b.grad) < 5e-3


# ------------------------------------------------------------------------------------------------ elementwise
def test_activations_and_geglu(dev):
    from stable_diffusion_training_amd import ops
    x = (rnd((1000, 64), dev, 1) * 3).requires_grad_(True)
    dy = rnd((1000, 64), dev, 2)
    for fn, ref in ((ops.silu, F.silu), (ops.quick_gelu, lambda t: t * torch.sigmoid(1.702 * t)), (ops.gelu_erf, F.gelu)):
        x.grad = None
        y = fn(x)
        xr = x.detach().float().requires_grad_(True)
        yr = ref(xr)
        assert rel_l2(y, yr) < 6e-3
        y.backward(dy)
        yr.backward(dy.float())
        assert rel_l2(x.grad, xr.grad) < 8e-3
    h = (rnd((300, 2 * 640), dev, 3) * 2).requires_grad_(True)
    o = ops.geglu(h)
    hr = h.detach().float().requires_grad_(True)
    a, g = hr.chunk(2, -1)
    oref = a * F.gelu(g, approximate="tanh")
    assert rel_l2(o, oref) < 6e-3
    do = rnd((300, 640), dev, 4)
    o.backward(do)
    oref.backward(do.float())
    assert rel_l2(h.grad, hr.grad) < 8e-3


def test_upsample_concat_add(dev):
    from stable_diffusion_training_amd import ops
    x = rnd((2, 8, 12, 64), dev, 1).requires_grad_(True)
    y = ops.upsample2x(x)
    assert torch.equal(y, x.detach().repeat_interleave(2, 1).repeat_interleave(2, 2))
    dy = rnd((2, 16, 24, 64), dev, 2)
    y.backward(dy)
    ref = dy.float().view(2, 8, 2, 12, 2, 64).sum((2, 4))
    assert rel_l2(x.grad, ref) < 6e-3
    a, b = rnd((2, 5, 5, 64), dev, 3).requires_grad_(True), rnd((2, 5, 5, 128), dev, 4).requires_grad_(True)
    c = ops.concat_channels(a, b)
    assert torch.equal(c, torch.cat([a.detach(), b.detach()], -1))
    dc = rnd((2, 5, 5, 192), dev, 5)
    c.backward(dc)
    assert torch.equal(a.grad, dc[..., :64]) and torch.equal(b.grad, dc[..., 64:])
    s = ops.add(a.detach(), a.detach())
    assert rel_l2(s, 2 * a.detach().float()) < 4e-3


def test_scheduler_kernels_vs_oracle(dev):
    from oracle import schedulers as osched
    from stable_diffusion_training_amd import _lib, schedulers
    for sched_name, pred in (("scaled_linear", "epsilon"), ("zero_snr_scaled_linear", "v_prediction")):
        s = schedulers.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule=sched_name, prediction_type=pred)
        st = s.create_state(dev)
        ost = osched.create_state(sched_name)
        g = torch.Generator().manual_seed(0)
        lat, noise = torch.randn(4, 4, 16, 16, generator=g), torch.randn(4, 4, 16, 16, generator=g)
        t = torch.tensor([0, 1, 500, 999], dtype=torch.int32)
        noisy, target, noisy_nchw = s.add_noise_and_target(st, lat.to(dev), noise.to(dev), t.to(dev), want_noisy_nchw=True)
        ref = osched.add_noise(ost, lat.numpy(), noise.numpy(), t.numpy())
        np.testing.assert_allclose(noisy_nchw.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
        assert rel_l2(noisy[..., :4].permute(0, 3, 1, 2), torch.from_numpy(ref)) < 4e-3
        assert noisy[..., 4:].abs().max() == 0
        if pred == "v_prediction":
            np.testing.assert_allclose(target.cpu().numpy(), osched.get_velocity(ost, lat.numpy(), noise.numpy(), t.numpy()), rtol=1e-6, atol=1e-6)
        else:
            assert target.data_ptr() == noise.to(dev).data_ptr() or torch.equal(target.cpu(), noise)


def test_posterior_mse_timestep(dev):
    from oracle import nets as onets
    from stable_diffusion_training_amd import _lib, nets
    s = torch.cuda.current_stream().cuda_stream
    mom = rnd((2, 8, 8, 8), dev, 1) * 3
    eps = torch.randn(2, 8, 8, 4, generator=torch.Generator().manual_seed(1)).to(dev)
    lat = torch.empty(2, 4, 8, 8, device=dev)
    _lib.call("sdt_vae_posterior_sample", mom.data_ptr(), eps.data_ptr(), lat.data_ptr(), 2, 4, 8, 8, 8, 0.18215, s)
    ref = onets.vae_sample_latents(mom.float().cpu(), eps.cpu())
    np.testing.assert_allclose(lat.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=1e-6)
    pred = rnd((2, 8, 8, 8), dev, 2)
    tgt = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(2)).to(dev)
    w = torch.tensor([0.5, 2.0], device=dev)
    loss = torch.zeros(1, device=dev)
    dpred = torch.empty_like(pred)
    from stable_diffusion_training_amd import ops
    rws = ops.reduce_workspace(_lib.load().sdt_reduce_workspace_bytes(), dev)
    _lib.call("sdt_mse_loss_fwd_bwd", pred.data_ptr(), tgt.data_ptr(), w.data_ptr(), loss.data_ptr(), dpred.data_ptr(), 2, 4, 8, 8, 8,
              rws.data_ptr(), rws.numel(), s)
    p = pred.float()[..., :4].permute(0, 3, 1, 2).requires_grad_(True)
    lref = (((tgt - p) ** 2) * w[:, None, None, None]).mean()
    lref.backward()
    assert abs(loss.item() - lref.item()) < 1e-5 * max(1, abs(lref.item()))
    assert rel_l2(dpred[..., :4].permute(0, 3, 1, 2), p.grad) < 6e-3 and dpred[..., 4:].abs().max() == 0
    t = torch.tensor([0, 10, 999], dtype=torch.int32, device=dev)
    e = nets.timestep_embedding(t, 320)
    assert rel_l2(e, onets.timestep_embedding(t.cpu(), 320)) < 4e-3


def test_embedding_colsum_transpose_softmax(dev):
    from stable_diffusion_training_amd import _lib, ops
    s = torch.cuda.current_stream().cuda_stream
    fs = FakeStore([("t/embedding", (100, 48)), ("p/embedding", (77, 48))], dev)
    ids = torch.randint(0, 100, (3, 77), generator=torch.Generator().manual_seed(0), dtype=torch.int32).to(dev)
    anchor = torch.zeros(1, device=dev, requires_grad=True)
    out = ops.embedding(ids, fs.st, "t/embedding", "p/embedding", 77, anchor)
    ref = fs.w["t/embedding"].to(dev)[ids.long()] + fs.w["p/embedding"].to(dev)[None]
    assert rel_l2(out, ref) < 4e-3
    dout = rnd((3, 77, 48), dev, 1)
    out.backward(dout)
    gt = torch.zeros(100, 48, device=dev).index_add_(0, ids.view(-1).long(), dout.float().view(-1, 48))
    assert rel_l2(fs.st.g("t/embedding"), gt) < 1e-4 and rel_l2(fs.st.g("p/embedding"), dout.float().sum(0)) < 1e-4
    x = rnd((130, 200), dev, 2)
    y = torch.empty(200, 130, dtype=BF, device=dev)
    _lib.call("sdt_transpose_bf16", x.data_ptr(), y.data_ptr(), 1, 130, 200, s)
    assert torch.equal(y, x.t().contiguous())
    z = rnd((37, 500), dev, 3) * 4
    zr = torch.softmax(z.float() * 0.3, -1)
    _lib.call("sdt_softmax_rows_inplace", z.data_ptr(), 37, 500, 0.3, s)
    assert rel_l2(z, zr) < 6e-3


def test_param_prepare(dev):
    """fp32 master -> the bf16 compute copy W (Flax layout, zero-padded to multiples of 8 channels); there is no transposed copy."""
    fs = FakeStore([("a/kernel", (3, 3, 4, 320)), ("b/kernel", (130, 72)), ("b/bias", (72,)), ("c/kernel", (1, 1, 64, 64))], dev)
    for name in ("a", "b", "c"):
        W, lf = fs.st.wmat(name + "/kernel")
        src = fs.w[name + "/kernel"].to(dev).to(BF).reshape(lf.batch, lf.R, lf.C)
        assert torch.equal(W[:, :lf.R, :lf.C], src)
        assert W[:, lf.R:].abs().sum() == 0 and W[:, :, lf.C:].abs().sum() == 0
    assert not hasattr(fs.st, "wt")


def test_optimizer_keeps_the_bf16_mirror_current(dev):
    """The optimizer sweep writes W = bf16(master) for every unpadded matrix leaf (and the per-step prepare refreshes the padded
    ones), so the forward of the next step reads current weights without a conversion pass."""
    from stable_diffusion_training_amd import params
    spec = [("a/kernel", (64, 48)), ("a/bias", (48,)), ("conv_in/kernel", (3, 3, 4, 32)), ("t/kernel", (32, 64))]
    st = params.ParamStore(spec, device=dev, quantise=True, quant_excluded=("bias", "conv_in", "t"), block_size=16)
    g = torch.Generator().manual_seed(0)
    st.load({k: torch.randn(s, generator=g) for k, s in spec})
    for _ in range(2):
        st.set_grad_flat(torch.randn(st.total, generator=g))
        st.optimizer_step(lr=1e-2, wd=0.07)
        st.prepare()
    for k, _ in spec:
        if k.endswith("kernel"):
            W, lf = st.wmat(k)
            src = st.p(k).to(BF).reshape(lf.batch, lf.R, lf.C)
            assert torch.equal(W[:, :lf.R, :lf.C], src), k


# ------------------------------------------------------------------------------------------------ optimizer
@pytest.mark.parametrize("bs", [16, 64])
@pytest.mark.parametrize("gscale", [1e-4, 3.0])  # below / above the clip threshold
def test_lion8_step_vs_oracle(dev, bs, gscale):
    from oracle import lion8
    from stable_diffusion_training_amd import params
    spec = [("a/kernel", (64, 48)), ("a/bias", (48,)), ("n/scale", (48,)), ("conv_in/kernel", (3, 3, 4, 32))]
    st = params.ParamStore(spec, device=dev, quantise=True, quant_excluded=("bias", "scale", "conv_in"),
                           wd_excluded=("bias", "scale"), block_size=bs, with_ema=True)
    g = torch.Generator().manual_seed(bs)
    w = {k: torch.randn(s, generator=g) for k, s in spec}
    st.load(w)
    pn = {k: v.numpy().copy() for k, v in w.items()}
    state = lion8.init_state(pn, lion8.create_mask(pn, ["bias", "scale", "conv_in"]), bs)
    ema = {k: v.copy() for k, v in pn.items()}
    dmask = lion8.create_mask(pn, ["bias", "scale"])
    assert st.grad16 is not None and st.g("a/kernel").dtype == torch.bfloat16 and st.g("a/bias").dtype == torch.float32
    for step in range(3):
        grads = {k: torch.randn(s, generator=g) * gscale for k, s in spec}
        for k, v in grads.items():
            st.g(k).copy_(v.to(dev))
            grads[k] = st.g(k).float().cpu()  # what the store holds: bf16 for the quantised kernel leaves (ParamStore.grad16), widened
        st.optimizer_step(lr=1e-3, wd=0.07, ema_rate=0.999)
        pn, state, gn = lion8.lion_step(pn, {k: v.numpy() for k, v in grads.items()}, state, lr=1e-3, wd=0.07, block_size=bs, decay_mask=dmask)
        ema = lion8.ema_update(ema, pn, 0.999)
        assert abs(st.grad_norm() - float(gn)) <= 1e-6 * float(gn)
        got, mom, gema = st.export(), st.export_momentum(), st.export("ema")
        for k in pn:
            np.testing.assert_allclose(got[k].cpu().numpy(), pn[k], rtol=0, atol=2e-7, err_msg=f"{k} step {step}")
            np.testing.assert_allclose(gema[k].cpu().numpy(), ema[k], rtol=0, atol=2e-7)
            if isinstance(state["mu"][k], tuple):
                codes, inv = mom[k]
                # integer work: bit-exact (the codec settles every code against the host-built float32 thresholds, and the
                # global norm is the float32 rounding of a double sum on both sides), over three steps of carried state
                assert np.array_equal(codes.cpu().numpy(), state["mu"][k][0]), f"{k}: int8 codes differ at step {step}"
                assert np.array_equal(inv.cpu().numpy(), state["mu"][k][1]), f"{k}: inverse scales differ at step {step}"
            else:
                np.testing.assert_allclose(mom[k].cpu().numpy(), state["mu"][k], rtol=1e-6, atol=1e-9)


def test_lion8_quantize_dequantize_roundtrip(dev):
    from oracle import lion8
    from stable_diffusion_training_amd import _lib
    s = torch.cuda.current_stream().cuda_stream
    x = torch.randn(4096, generator=torch.Generator().manual_seed(0))
    x[:16] = 0
    xd = x.to(dev)
    codes = torch.empty(4096, dtype=torch.int8, device=dev)
    inv = torch.empty(256, device=dev)
    from stable_diffusion_training_amd import params
    _lib.call("sdt_lion8_quantize", xd.data_ptr(), codes.data_ptr(), inv.data_ptr(), 4096, 16, params.lion_thresholds(dev).data_ptr(), s)
    rc, ri = lion8.block_quantize(x.numpy(), 16)
    assert np.array_equal(codes.cpu().numpy().reshape(-1, 16), rc) and np.array_equal(inv.cpu().numpy().reshape(-1, 1), ri)
    assert (codes[:16] == 3).all() and inv[0] == 1.0
    back = torch.empty(4096, device=dev)
    _lib.call("sdt_lion8_dequantize", codes.data_ptr(), inv.data_ptr(), back.data_ptr(), 4096, 16, s)
    np.testing.assert_allclose(back.cpu().numpy(), lion8.block_dequantize((4096,), codes.cpu().numpy().reshape(-1, 16), inv.cpu().numpy().reshape(-1, 1)), rtol=1e-6, atol=1e-12)


def _adversarial_codec_values():
    """Values of y = x / absmax that put |y + offset| on, one float32 below and one above every decision threshold of the 8-bit
    codec (lion_quant.py:52-59), both signs - where a device power / log estimate and the host's float32 `power` could round apart."""
    from stable_diffusion_training_amd import lion_codec
    t = lion_codec.quantization_thresholds()[1:]  # T[c], c = 1 .. 127
    around = np.concatenate([t, np.nextafter(t, np.float32(0)), np.nextafter(t, np.float32(2)),
                             np.nextafter(np.nextafter(t, np.float32(0)), np.float32(0)), np.nextafter(np.nextafter(t, np.float32(2)), np.float32(2))]).astype(np.float32)
    off = lion_codec.OFFSET
    pos = (around - off).astype(np.float32)           # y + offset ~ +a
    neg = (-around - off).astype(np.float32)          # y + offset ~ -a
    y = np.concatenate([pos, neg, np.float32([0.0, -0.0, 1e-40, -1e-40, 1e-38, -1e-38, -off, -2 * off, off])])
    return y[np.abs(y) <= 1.0]


@pytest.mark.parametrize("scale", [1.0, 2.0 ** -10, 2.0 ** -20, 3e-7, 7.3e4])
def test_lion8_codec_at_every_threshold(dev, scale):
    """sdt_lion8_quantize == oracle.lion8.block_quantize on every code, for blocks built to sit on the rounding boundaries: element 0
    of each block is +-scale (the absmax), the others are scale * y for the adversarial y above.  Plus all-zero blocks, blocks with
    one non-zero element (tiny, normal, huge) and float32 denormals inside a normal block."""
    from oracle import lion8
    from stable_diffusion_training_amd import _lib, params
    bs = 16
    y = _adversarial_codec_values()
    nblk = -(-y.size // (bs - 1))
    yy = np.zeros(nblk * (bs - 1), np.float32)
    yy[: y.size] = y
    x = np.empty((nblk, bs), np.float32)
    x[:, 0] = np.float32(scale) * np.where(np.arange(nblk) % 2 == 0, 1.0, -1.0).astype(np.float32)
    x[:, 1:] = (np.float32(scale) * yy.reshape(nblk, bs - 1)).astype(np.float32)
    extra = np.zeros((8, bs), np.float32)   # 0: all zero; 1..: a single non-zero element
    for r, v in enumerate([0.0, 1e-30, -1e-30, 1.0, -3.5, 1e30, 1e-37, -2e-37]):
        extra[r, (3 * r) % bs] = v
    x = np.concatenate([x, extra]).astype(np.float32)
    n = x.size
    xd = torch.from_numpy(x.reshape(-1)).to(dev)
    codes = torch.empty(n, dtype=torch.int8, device=dev)
    inv = torch.empty(n // bs, device=dev)
    _lib.call("sdt_lion8_quantize", xd.data_ptr(), codes.data_ptr(), inv.data_ptr(), n, bs, params.lion_thresholds(dev).data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    rc, ri = lion8.block_quantize(x.reshape(-1), bs)
    got_c, got_i = codes.cpu().numpy().reshape(-1, bs), inv.cpu().numpy().reshape(-1, 1)
    assert np.array_equal(got_i, ri), "inverse block scales differ"
    bad = np.argwhere(got_c != rc)
    assert bad.size == 0, f"{len(bad)} codes differ; first: block {bad[0][0]} elem {bad[0][1]} x={x[bad[0][0], bad[0][1]]!r} hip {got_c[bad[0][0], bad[0][1]]} oracle {rc[bad[0][0], bad[0][1]]}"
    assert len(np.unique(rc)) >= 250  # the case really sweeps the code range (255 values, +-127 .. 0)


def test_lion8_million_heavy_tailed_elements_three_steps_exact(dev):
    """2^20 Student-t(2) distributed weights and gradients (heavy tails: block scales spread over many octaves, most codes small,
    a few at +-127), three carried optimizer steps: 8-bit codes, inverse scales AND fp32 masters equal the oracle (lion_quant.py:133-154
    via oracle.lion8.lion_step) exactly."""
    from oracle import lion8
    from stable_diffusion_training_amd import params
    bs = 16
    spec = [("a/kernel", (1024, 1024)), ("a/bias", (1024,))]
    st = params.ParamStore(spec, device=dev, quantise=True, quant_excluded=("bias",), wd_excluded=("bias",), block_size=bs, with_ema=True)
    rs = np.random.RandomState(20)
    w = {k: torch.from_numpy((0.05 * rs.standard_t(2, size=s)).astype(np.float32)) for k, s in spec}
    st.load(w)
    pn = {k: v.numpy().copy() for k, v in w.items()}
    state = lion8.init_state(pn, lion8.create_mask(pn, ["bias"]), bs)
    dmask = lion8.create_mask(pn, ["bias"])
    for step in range(3):
        grads = {k: torch.from_numpy((10.0 ** rs.uniform(-6, -2) * rs.standard_t(2, size=s)).astype(np.float32)) for k, s in spec}
        for k, v in grads.items():
            st.g(k).copy_(v.to(dev))
            grads[k] = st.g(k).float().cpu()  # the stored gradient (bf16 for the kernel leaf), widened: the oracle's input
        st.optimizer_step(lr=1e-3, wd=0.07, ema_rate=0.999)
        pn, state, gn = lion8.lion_step(pn, {k: v.numpy() for k, v in grads.items()}, state, lr=1e-3, wd=0.07, block_size=bs, decay_mask=dmask)
        assert abs(st.grad_norm() - float(gn)) <= 1e-6 * float(gn)
        got, mom = st.export(), st.export_momentum()
        codes, inv = mom["a/kernel"]
        assert np.array_equal(codes.cpu().numpy(), state["mu"]["a/kernel"][0]), f"int8 codes differ at step {step}"
        assert np.array_equal(inv.cpu().numpy(), state["mu"]["a/kernel"][1]), f"inverse scales differ at step {step}"
        assert np.array_equal(got["a/kernel"].cpu().numpy(), pn["a/kernel"]), f"fp32 master differs at step {step}"
    hist = np.bincount(state["mu"]["a/kernel"][0].astype(np.int32).reshape(-1) + 128, minlength=256)
    assert (hist > 0).sum() >= 200  # heavy tails: the whole code range is in use


# ------------------------------------------------------------------------------------------------ fused fan-in of gradients
@pytest.mark.parametrize("kind", ["layer", "group"])
def test_norm_skip_output_folds_residual_gradient(dev, kind):
    """norm(x, skip=True) -> (y, x'): the gradient arriving on x' is added inside the norm's backward kernel."""
    from stable_diffusion_training_amd import ops
    C = 320
    fs = FakeStore([("n/scale", (C,)), ("n/bias", (C,))], dev, seed=3)
    x = rnd((2, 64, C), dev, 1).requires_grad_(True)
    h = x * 1.0
    if kind == "layer":
        y, xs = ops.layer_norm(h, fs.st, "n", skip=True)
    else:
        y, xs = ops.group_norm(h, fs.st, "n", 32, 1e-5, silu=True, skip=True)
    assert xs.data_ptr() == h.data_ptr()
    w1, w2 = rnd((2, 64, C), dev, 2), rnd((2, 64, C), dev, 3)
    ((y * w1).sum() + (xs * w2).sum()).backward()
    xr = x.detach().float().requires_grad_(True)
    g, b = fs.w["n/scale"].to(dev), fs.w["n/bias"].to(dev)
    if kind == "layer":
        yr = F.layer_norm(xr, (C,), g, b, 1e-5)
    else:
        yr = F.silu(F.group_norm(xr.transpose(1, 2), 32, g, b, 1e-5).transpose(1, 2))
    ((yr * w1.float()).sum() + (xr * w2.float()).sum()).backward()
    assert rel_l2(x.grad, xr.grad) < 1e-2


@pytest.mark.parametrize("n,used", [(3, 3), (5, 3), (40, 40)])
def test_fanout_sums_gradients_in_one_pass(dev, n, used):
    from stable_diffusion_training_amd import ops
    x = rnd((4, 77, 768), dev, 1).requires_grad_(True)
    outs = ops.fanout(x * 1.0, n)
    ws = [rnd((4, 77, 768), dev, 10 + i) for i in range(used)]
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()  # aliases beyond `used` get no gradient
    ref = sum(w.float() for w in ws)
    assert rel_l2(x.grad, ref) < 6e-3


# ------------------------------------------------------------------------------------------------ merged projections
@pytest.mark.parametrize("M,K,N,n,bias", [(4096, 320, 320, 3, False), (308, 768, 768, 3, True), (308, 768, 640, 2, False), (100, 64, 64, 3, True)])
def test_linear_multi_matches_separate_linears(dev, M, K, N, n, bias):
    """ops.linear_multi: n Dense layers sharing their input as ONE GEMM each for forward, dgrad (plain multi-segment
    reduction) and wgrad (segmented output + fused bias gradient), against the same layers applied one by one."""
    from stable_diffusion_training_amd import ops
    names = [f"p{i}" for i in range(n)]
    # kernels back to back, then biases back to back: the order the (quantised?, decayed?) segments give the real stores
    spec = [(nm + "/kernel", (K, N)) for nm in names] + ([(nm + "/bias", (N,)) for nm in names] if bias else [])
    fs = FakeStore(spec, dev, seed=K + N)
    x = rnd((M, K), dev, 1).requires_grad_(True)
    y = ops.linear_multi(x, fs.st, names)
    assert y is not None and y.shape == (M, n * N)
    dy = rnd((M, n * N), dev, 2)
    y.backward(dy)
    dx_ref = 0
    for i, nm in enumerate(names):
        wq = fs.w[nm + "/kernel"].to(dev).to(BF).float()
        ref = x.detach().float() @ wq
        if bias:
            ref = ref + fs.w[nm + "/bias"].to(dev)
        assert rel_l2(y[:, i * N:(i + 1) * N], ref) < 6e-3
        dyi = dy[:, i * N:(i + 1) * N].float()
        dx_ref = dx_ref + dyi @ wq.t()
        assert rel_l2(fs.st.g(nm + "/kernel"), x.detach().float().t() @ dyi) < 2e-3
        if bias:
            assert rel_l2(fs.st.g(nm + "/bias"), dyi.sum(0)) < 2e-3
    assert rel_l2(x.grad, dx_ref) < 6e-3


@pytest.mark.parametrize("B,C,hw,n", [(4, 320, 16, 3), (2, 1280, 8, 5)])
def test_time_embedding_projections_as_one_gemm(dev, B, C, hw, n):
    """nets._time_emb_projections: the Dense(silu(temb)) of n ResBlocks of one width run as ONE GEMM whose column slices are the
    blocks' row biases (strided `rowbias`, ld_rowbias = n*C, added inside the convolution's epilogue); against the same layers
    applied one by one (reference: diffusers FlaxResnetBlock2D, temb = time_emb_proj(silu(temb)); hidden += temb[:, None, None])."""
    from stable_diffusion_training_amd import nets, ops
    T = 1280
    names = [f"r{i}" for i in range(n)]
    spec = nets._Spec()
    for nm in names:
        spec.dense(nm + "/time_emb_proj", T, C)
        spec.conv(nm + "/conv1", C, C)
    # grouped layout (what unet_spec emits) and the interleaved one (per-block fallback)
    grouped = ([(nm + "/conv1/" + l, sh) for nm in names for l, sh in (("kernel", (3, 3, C, C)), ("bias", (C,)))]
               + [(nm + "/time_emb_proj/kernel", (T, C)) for nm in names] + [(nm + "/time_emb_proj/bias", (C,)) for nm in names])
    out = []
    for layout in (grouped, list(spec)):
        fs = FakeStore(layout, dev, seed=3)
        temb = rnd((B, T), dev, 1).requires_grad_(True)
        xs = [rnd((B, hw, hw, C), dev, 10 + i) for i in range(n)]
        rbs = nets._time_emb_projections(fs.st, ops.silu(temb))
        assert set(rbs) == set(names) and all(rb.shape == (B, C) for rb in rbs.values())
        if layout is grouped:
            assert rbs[names[0]].stride(0) == n * C  # column slices of one GEMM's output
        ys = [ops.conv2d(x, fs.st, nm + "/conv1", rowbias=rbs[nm]) for x, nm in zip(xs, names)]
        torch.autograd.backward(ys, [rnd((B, hw, hw, C), dev, 20 + i) for i in range(n)])
        out.append((ys, temb.grad, {k: fs.st.g(k).clone() for k, _ in layout if "time_emb_proj" in k}))
        for i, nm in enumerate(names):  # absolute check of the forward
            wq = fs.w[nm + "/time_emb_proj/kernel"].to(dev).to(BF).float()
            ref = torch.nn.functional.silu(temb.detach().float()).to(BF).float() @ wq + fs.w[nm + "/time_emb_proj/bias"].to(dev)
            assert rel_l2(rbs[nm], ref) < 6e-3
    (ya, ga, wa), (yb, gb, wb) = out
    for a, b in zip(ya, yb):
        assert rel_l2(a, b) < 2e-3
    assert rel_l2(ga, gb) < 6e-3
    for k in wa:
        assert rel_l2(wa[k], wb[k]) < 2e-3, k


@pytest.mark.parametrize("B,C,heads,Nq,Nk,n", [(4, 320, 8, 256, 77, 3), (2, 1280, 8, 64, 77, 2)])
def test_context_projections_as_one_gemm(dev, B, C, heads, Nq, Nk, n):
    """nets._context_projections: to_k / to_v of n cross-attention blocks of one width as ONE GEMM over the text context, each
    block's attention reading its [k|v] as a column slice (row pitch n*2C) and its gradient gathered back into one tensor for
    ONE input-gradient and ONE weight-gradient GEMM; against the per-block path (interleaved layout).  Reference: diffusers
    FlaxAttention (attention_flax.py), key = to_k(context), value = to_v(context)."""
    from stable_diffusion_training_amd import nets, ops
    cd = 768
    names = [f"b{i}/attn2" for i in range(n)]
    per_block = [(nm + "/" + l + "/kernel", (cd if l in ("to_k", "to_v") else C, C)) for nm in names for l in ("to_q", "to_k", "to_v")]
    grouped = ([(nm + "/to_q/kernel", (C, C)) for nm in names]
               + [(nm + "/" + l + "/kernel", (cd, C)) for nm in names for l in ("to_k", "to_v")])
    out = []
    for layout in (grouped, per_block):
        fs = FakeStore(layout, dev, seed=4)
        ctx = rnd((B, Nk, cd), dev, 1).requires_grad_(True)
        xs = [rnd((B, Nq, C), dev, 10 + i) for i in range(n)]
        kv = nets._context_projections(fs.st, ctx)
        assert set(kv) == set(names)
        if layout is grouped:
            assert all(isinstance(v, nets._PackedKV) and v.kv.stride(1) == n * 2 * C for v in kv.values())
        else:
            assert not any(isinstance(v, nets._PackedKV) for v in kv.values())
        ys = []
        for x, nm in zip(xs, names):
            if layout is grouped:
                ys.append(ops.attention_packed(ops.linear(x, fs.st, nm + "/to_q"), kv[nm].kv, heads, (C // heads) ** -0.5))
            else:
                pk = ops.linear_multi(kv[nm], fs.st, (nm + "/to_k", nm + "/to_v"))
                assert pk is not None
                ys.append(ops.attention_packed(ops.linear(x, fs.st, nm + "/to_q"), pk, heads, (C // heads) ** -0.5))
        torch.autograd.backward(ys, [rnd((B, Nq, C), dev, 20 + i) for i in range(n)])
        out.append((ys, ctx.grad, {k: fs.st.g(k).clone() for k, _ in layout}))
    (ya, ga, wa), (yb, gb, wb) = out
    for a, b in zip(ya, yb):
        assert rel_l2(a, b) < 2e-3
    assert rel_l2(ga, gb) < 6e-3
    for k in wa:
        assert rel_l2(wa[k], wb[k]) < 2e-3, k


@pytest.mark.parametrize("B,H,Nq,Nk,D,causal,cross", [(2, 8, 256, 256, 40, False, False), (2, 8, 256, 77, 80, False, True),
                                                      (3, 12, 77, 77, 64, True, False), (2, 8, 4096, 77, 40, False, True)])
def test_attention_packed(dev, B, H, Nq, Nk, D, causal, cross):
    from stable_diffusion_training_amd import ops
    C = H * D
    scale = D ** -0.5
    if cross:
        a = rnd((B, Nq, C), dev, 1).requires_grad_(True)
        b = rnd((B, Nk, 2 * C), dev, 2).requires_grad_(True)
        q, k, v = a, b[..., :C], b[..., C:]
    else:
        a = rnd((B, Nq, 3 * C), dev, 1).requires_grad_(True)
        b = None
        q, k, v = a[..., :C], a[..., C:2 * C], a[..., 2 * C:]
    o = ops.attention_packed(a, b, H, scale, causal)
    ar = a.detach().float().requires_grad_(True)
    br = None if b is None else b.detach().float().requires_grad_(True)
    if cross:
        qr, kr, vr = ar, br[..., :C], br[..., C:]
    else:
        qr, kr, vr = ar[..., :C], ar[..., C:2 * C], ar[..., 2 * C:]
    oref = attn_ref(qr, kr, vr, H, scale, causal)
    assert rel_l2(o, oref) < 8e-3
    do = rnd((B, Nq, C), dev, 4)
    o.backward(do)
    oref.backward(do.float())
    assert rel_l2(a.grad, ar.grad) < 1.5e-2
    if cross:
        assert rel_l2(b.grad, br.grad) < 1.5e-2


# ------------------------------------------------------------------------------------------------ grouped weight gradients
def test_grouped_dense_weight_gradients_match_single_launches(dev):
    """ops.wgrad_grouping(): the weight gradients of Dense layers / 1x1 convolutions / merged projections queued and issued by
    sdt_gemm_tn_wgrad_group (one launch per tile size) against the same layers with one launch each: the same sums up to the
    fp32 order of the split reductions, bit for bit equal between two grouped runs, and grad_ready fires for every leaf."""
    from stable_diffusion_training_amd import ops
    shapes = [("a", 4096, 640, 640), ("b", 16384, 320, 320), ("c", 308, 768, 3072), ("d", 1024, 1280, 1280), ("e", 4096, 640, 5120),
              ("f", 256, 1280, 1280), ("g", 100, 64, 136), ("h", 16384, 320, 2560), ("i", 308, 3072, 768), ("j", 4, 1280, 320),
              ("k", 1024, 5120, 1280), ("l", 4096, 2560, 640), ("m", 77, 768, 768), ("n", 2048, 320, 320)]
    spec = []
    for n, M, K, N in shapes:
        spec += [(f"{n}/kernel", (K, N)), (f"{n}/bias", (N,))]
    spec += [("q0/kernel", (320, 320)), ("q1/kernel", (320, 320)), ("q2/kernel", (320, 320)), ("cv/kernel", (1, 1, 640, 320)), ("cv/bias", (320,))]
    fs = FakeStore(spec, dev, seed=11)
    xs = {n: rnd((M, K), dev, i).requires_grad_(True) for i, (n, M, K, N) in enumerate(shapes)}
    dys = {n: rnd((M, N), dev, 100 + i) for i, (n, M, K, N) in enumerate(shapes)}
    xq, dyq = rnd((4096, 320), dev, 50).requires_grad_(True), rnd((4096, 960), dev, 51)
    xc, dyc = rnd((2, 32, 32, 640), dev, 52).requires_grad_(True), rnd((2, 32, 32, 320), dev, 53)
    ready = []
    fs.st.grad_ready = ready.append

    def run(grouped):
        fs.st.grad.zero_()
        del ready[:]
        outs = [ops.linear(xs[n], fs.st, n) for n, *_ in shapes]
        outs.append(ops.linear_multi(xq, fs.st, ("q0", "q1", "q2")))
        outs.append(ops.conv2d(xc, fs.st, "cv", pad=0))
        gs = [dys[n] for n, *_ in shapes] + [dyq, dyc]
        if grouped:
            with ops.wgrad_grouping():
                torch.autograd.backward(outs, gs)
                assert len(ready) < len(fs.st.leaves)  # held back ...
        else:
            torch.autograd.backward(outs, gs)
        torch.cuda.synchronize()
        assert sorted(ready) == sorted(fs.st.leaves)   # ... and all reported once the context has closed
        return fs.st.grad.clone()

    g0 = run(False)
    g1 = run(True)
    g2 = run(True)
    assert torch.equal(g1, g2), "grouped weight gradients differ between two launches"
    for n, M, K, N in shapes:
        lf = fs.st.leaves[f"{n}/kernel"]
        a, b = g1[lf.offset: lf.offset + lf.numel], g0[lf.offset: lf.offset + lf.numel]
        assert rel_l2(a, b) < 1e-5, n
        ref = xs[n].detach().float().t() @ dys[n].float()
        assert rel_l2(a.view(K, N), ref) < 2e-3, n
    assert rel_l2(g1, g0) < 1e-5


def test_grouped_conv_weight_gradients_match_single_launches(dev):
    """3x3 convolution weight gradients queued by ops.wgrad_grouping() and issued by sdt_conv_wgrad_group (shared launches for the
    three-taps-per-workgroup kernel's shapes, single launches for the rest: stride 2, 8x8 images, odd widths) against one launch
    each: equal up to the fp32 order of the split reductions, bitwise equal between two grouped runs."""
    from stable_diffusion_training_amd import ops
    convs = [("a", 4, 64, 64, 320, 320, 1), ("b", 4, 32, 32, 640, 640, 1), ("c", 4, 16, 16, 1280, 1280, 1), ("d", 4, 8, 8, 1280, 1280, 1),
             ("e", 2, 32, 32, 320, 320, 2), ("f", 4, 64, 64, 640, 320, 1), ("g", 2, 24, 40, 64, 64, 1), ("h", 4, 64, 64, 8, 320, 1),
             ("i", 4, 32, 32, 1280, 640, 1), ("j", 4, 64, 64, 320, 8, 1)]
    spec = []
    for n, B, H, W, Ci, Co, st in convs:
        spec += [(f"{n}/kernel", (3, 3, Ci if Ci != 8 else 4, Co if Co != 8 else 4)), (f"{n}/bias", (Co if Co != 8 else 4,))]
    fs = FakeStore(spec, dev, seed=21)
    xs = {n: rnd((B, H, W, Ci), dev, i).requires_grad_(True) for i, (n, B, H, W, Ci, Co, st) in enumerate(convs)}

    def run(grouped):
        fs.st.grad.zero_()
        outs = [ops.conv2d(xs[n], fs.st, n, stride=st, pad=1) for n, B, H, W, Ci, Co, st in convs]
        gs = [rnd(tuple(o.shape), dev, 40 + i) for i, o in enumerate(outs)]
        if grouped:
            with ops.wgrad_grouping():
                torch.autograd.backward(outs, gs)
        else:
            torch.autograd.backward(outs, gs)
        torch.cuda.synchronize()
        return fs.st.grad.clone()

    g0, g1, g2 = run(False), run(True), run(True)
    assert torch.equal(g1, g2)
    for n, *_ in convs:
        lf = fs.st.leaves[f"{n}/kernel"]
        assert rel_l2(g1[lf.offset: lf.offset + lf.numel], g0[lf.offset: lf.offset + lf.numel]) < 1e-5, n
        lb = fs.st.leaves[f"{n}/bias"]
        assert rel_l2(g1[lb.offset: lb.offset + lb.numel], g0[lb.offset: lb.offset + lb.numel]) < 1e-5, n


def test_deferred_layernorm_parameter_gradients(dev):
    """Inside ops.wgrad_grouping() the dgamma / dbeta sums of LayerNorms are held back and issued as ONE launch
    (sdt_norm_param_grads_group): the same partial rows added in the same order - bit for bit the immediate result."""
    from stable_diffusion_training_amd import ops
    shapes = [("a", (16384, 320)), ("b", (4096, 640)), ("c", (308, 768)), ("d", (1024, 1280)), ("e", (33, 2048))]
    spec = []
    for n, (M, C) in shapes:
        spec += [(f"{n}/scale", (C,)), (f"{n}/bias", (C,))]
    fs = FakeStore(spec, dev, seed=4)
    xs = {n: rnd(shp, dev, i) * 2 + 0.3 for i, (n, shp) in enumerate(shapes)}
    dys = {n: rnd(shp, dev, 20 + i) for i, (n, shp) in enumerate(shapes)}
    ready = []
    fs.st.grad_ready = ready.append

    def run(grouped):
        fs.st.grad.zero_()
        del ready[:]
        ins = {n: xs[n].clone().requires_grad_(True) for n, _ in shapes}
        outs = [ops.layer_norm(ins[n] * 1.0, fs.st, n) for n, _ in shapes]
        if grouped:
            with ops.wgrad_grouping():
                torch.autograd.backward(outs, [dys[n] for n, _ in shapes])
                assert not ready
        else:
            torch.autograd.backward(outs, [dys[n] for n, _ in shapes])
        torch.cuda.synchronize()
        assert sorted(ready) == sorted(fs.st.leaves)
        return fs.st.grad.clone(), [ins[n].grad.clone() for n, _ in shapes]

    g0, dx0 = run(False)
    g1, dx1 = run(True)
    assert torch.equal(g0, g1) and all(torch.equal(a, b) for a, b in zip(dx0, dx1))
    for n, (M, C) in shapes:
        xr = xs[n].float().requires_grad_(True)
        g, b = fs.w[f"{n}/scale"].to(dev).requires_grad_(True), fs.w[f"{n}/bias"].to(dev).requires_grad_(True)
        F.layer_norm(xr, (C,), g, b, 1e-5).backward(dys[n].float())
        assert rel_l2(fs.st.g(f"{n}/scale"), g.grad) < 5e-3 and rel_l2(fs.st.g(f"{n}/bias"), b.grad) < 5e-3


# ------------------------------------------------------------------------------------------------ fused GroupNorm statistics
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,pad,res", [
    (2, 64, 64, 320, 320, 3, 1, True),     # halo kernel, normal epilogue, N = 2.5 channel tiles
    (2, 32, 32, 640, 1280, 3, 1, False),   # halo kernel, split over channel chunks (last arriver finishes the tile)
    (2, 16, 16, 128, 256, 1, 0, True),     # 1x1: generic kernel
    (4, 16, 16, 1280, 1280, 3, 1, False),  # halo kernel 16 x 16 tile = one image, split
    (3, 8, 8, 1280, 640, 3, 1, False),     # 64-pixel images: fusable only with 64-row tiles, else the standalone pass
])
def test_conv_epilogue_groupnorm_statistics(dev, B, H, W, Cin, Cout, k, pad, res):
    """conv2d(gn_groups=32) hands the next GroupNorm its {sum, sumsq}: group_norm(stats=) must equal the standalone path."""
    from stable_diffusion_training_amd import ops
    fs = FakeStore([("c/kernel", (k, k, Cin, Cout)), ("c/bias", (Cout,)), ("n/scale", (Cout,)), ("n/bias", (Cout,))], dev, seed=Cout)
    x = rnd((B, H, W, Cin), dev, 1)
    r = rnd((B, H, W, Cout), dev, 2) if res else None
    y, stats = ops.conv2d(x, fs.st, "c", pad=pad, residual=r, gn_groups=32)
    y0 = ops.conv2d(x, fs.st, "c", pad=pad, residual=r)
    assert rel_l2(y, y0) < 1e-3  # split-K summation order may differ between two launches
    a0 = ops.group_norm(y, fs.st, "n", 32, 1e-5, silu=True)
    if stats is None:
        pytest.skip("shape not fusable with the chosen tiling (falls back to the standalone statistics pass)")
    cpg = Cout // 32
    yf = y.float().view(B, H * W, 32, cpg)
    ref = torch.stack([yf.sum(dim=(1, 3)), (yf * yf).sum(dim=(1, 3))], dim=-1)
    assert stats.dim() == 4 and tuple(stats.shape[2:]) == (32, 2)  # (B, partial rows, G, 2): written, not accumulated
    assert rel_l2(stats.sum(1), ref) < 1e-4
    a1 = ops.group_norm(y, fs.st, "n", 32, 1e-5, silu=True, stats=stats)
    assert rel_l2(a1, a0) < 2e-3
    # no atomics anywhere on the path: a second launch gives the same bits (partials, normalised output)
    y2, stats2 = ops.conv2d(x, fs.st, "c", pad=pad, residual=r, gn_groups=32)
    assert torch.equal(y2, y) and torch.equal(stats2, stats)
    assert torch.equal(ops.group_norm(y, fs.st, "n", 32, 1e-5, silu=True, stats=stats2), a1)
    assert torch.equal(ops.group_norm(y, fs.st, "n", 32, 1e-5, silu=True), a0)


def test_linear_epilogue_groupnorm_statistics(dev):
    from stable_diffusion_training_amd import ops
    B, HW, K, N = 2, 1024, 640, 640
    fs = FakeStore([("l/kernel", (K, N)), ("l/bias", (N,)), ("n/scale", (N,)), ("n/bias", (N,))], dev, seed=5)
    x = rnd((B, HW, K), dev, 1)
    r = rnd((B, HW, N), dev, 2)
    y, stats = ops.linear(x, fs.st, "l", residual=r, gn_groups=32)
    assert stats is not None
    yf = y.float().view(B, HW, 32, N // 32)
    ref = torch.stack([yf.sum(dim=(1, 3)), (yf * yf).sum(dim=(1, 3))], dim=-1)
    assert rel_l2(stats.sum(1), ref) < 1e-4


def test_lion_quant_facade_matches_oracle(dev):
    """lion_quant.lion_8bit (the reference's optimizer contract) against the NumPy oracle: two updates, quantised and
    fp32 leaves, decayed and not, no clipping (the bare transformation)."""
    from oracle import lion8
    from stable_diffusion_training_amd import lion_quant
    g = torch.Generator().manual_seed(3)
    params = {"a/kernel": torch.randn(48, 32, generator=g) * 0.05, "a/bias": torch.randn(32, generator=g) * 0.05,
              "b/kernel": torch.randn(3, 3, 16, 16, generator=g) * 0.05, "n/scale": torch.ones(16)}
    qmask = {"a/kernel": True, "a/bias": False, "b/kernel": True, "n/scale": False}
    dmask = {"a/kernel": True, "a/bias": False, "b/kernel": True, "n/scale": False}
    lr, wd = 1e-3, 0.07
    tx = lion_quant.lion_8bit(lr, block_size=16, weight_decay=wd, mask=dmask, excluded_layer_mask=qmask)
    p_dev = {k: v.to(dev) for k, v in params.items()}
    state = tx.init(p_dev)
    assert isinstance(state, lion_quant.ScaleBy8bitLionState) and state.count == 0 and state.mu_quant_flag == qmask
    assert isinstance(state.mu_quant["a/kernel"], tuple) and int(state.mu_quant["a/kernel"][0].float().abs().max()) == 3
    p_ref = {k: v.numpy() for k, v in params.items()}
    s_ref = lion8.init_state(p_ref, qmask, 16)
    for step in range(2):
        grads = {k: torch.randn(v.shape, generator=g) * 1e-2 for k, v in params.items()}
        upd, state = tx.update({k: v.to(dev) for k, v in grads.items()}, state, p_dev)
        p_dev = {k: p_dev[k] + upd[k] for k in p_dev}
        p_ref, s_ref, _ = lion8.lion_step(p_ref, {k: v.numpy() for k, v in grads.items()}, s_ref, lr=lr, wd=wd, block_size=16,
                                          decay_mask=dmask, clip=None)
        assert state.count == step + 1
        for k in params:
            # +-lr steps: identical except where the interpolated momentum is ~0 (sign of a rounding-noise value)
            diff = (p_dev[k].cpu().numpy() - p_ref[k])
            assert (abs(diff) > 1e-7).mean() < 5e-3, (k, step)
        assert np.array_equal(state.mu_quant["a/kernel"][0].cpu().numpy(), s_ref["mu"]["a/kernel"][0])  # int8 codes: exact
    with pytest.raises(ValueError):
        tx.update({k: v.to(dev) for k, v in grads.items()}, state, None)


@pytest.mark.parametrize("M,C", [(16384, 320), (4096, 640), (1024, 1280), (16000, 320), (300, 64)])
def test_feed_forward_geglu_fused_epilogues_equal_the_separate_ops(dev, M, C):
    """ops.feed_forward_geglu (GEGLU inside the FF1 epilogue) against linear -> geglu -> linear: the same bf16 rounding points, so
    outputs, input gradients and all four parameter gradients are equal bit for bit; (300, 64) is not served by the fused kernel and
    must take the fallback."""
    from stable_diffusion_training_amd import _lib, ops
    F = 4 * C
    spec = [("ff/net_0/proj/kernel", (C, 2 * F)), ("ff/net_0/proj/bias", (2 * F,)), ("ff/net_2/kernel", (F, C)), ("ff/net_2/bias", (C,))]
    fa, fb = FakeStore(spec, dev, seed=3), FakeStore(spec, dev, seed=3)
    assert bool(_lib.load().sdt_ff_geglu_supported(M, F, C)) == (M >= 1024)
    xa = rnd((M, C), dev, 1).requires_grad_(True)
    xb = xa.detach().clone().requires_grad_(True)
    res = rnd((M, C), dev, 2)
    ya = ops.feed_forward_geglu(xa, fa.st, "ff/net_0/proj", "ff/net_2", residual=res)
    yb = ops.linear(ops.geglu(ops.linear(xb, fb.st, "ff/net_0/proj")), fb.st, "ff/net_2", residual=res)
    assert torch.equal(ya, yb)
    wq1, wq2 = fa.w["ff/net_0/proj/kernel"].to(dev).to(BF).float(), fa.w["ff/net_2/kernel"].to(dev).to(BF).float()
    hr = xa.detach().float() @ wq1 + fa.w["ff/net_0/proj/bias"].to(dev)
    ref = (hr[:, :F] * F_gelu_tanh(hr[:, F:])) @ wq2 + fa.w["ff/net_2/bias"].to(dev) + res.float()
    assert rel_l2(ya, ref) < 8e-3
    dy = rnd((M, C), dev, 3)
    ya.backward(dy)
    yb.backward(dy)
    assert torch.equal(xa.grad, xb.grad)
    for k in ("ff/net_0/proj/kernel", "ff/net_0/proj/bias", "ff/net_2/kernel", "ff/net_2/bias"):
        assert torch.equal(fa.st.g(k), fb.st.g(k)), k


def F_gelu_tanh(x):
    return F.gelu(x, approximate="tanh")
