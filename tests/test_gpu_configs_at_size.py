"""BASELINE.json configs[3] (SD2.1-768, v-prediction, 9216-token self-attention) and configs[4] (SDXL-base 1024x1024, 2.57 B
parameter UNet, Lion-8bit state) AT SIZE on one MI355X: one full HIP train_step each, checked (i) against the fp32 CPU oracle's
forward on the same weights and inputs (the oracle's backward at these sizes does not fit the test budget; gradients are
oracle-checked on the same architectures at reduced resolution in test_gpu_model.py) and (ii) through size-independent
properties: finite loss in the band random-init weights give, gradient norms, 8-bit state that moved, parameters that moved by
exactly +-lr(1 + wd p), memory high-water.  The reference's own train_step cannot drive SDXL (no added_cond_kwargs, one text
encoder: SURVEY.md §8(d) note); its oracle is the restatement with explicit micro-conditioning inputs."""
import numpy as np
import pytest
import torch

from tests.helpers import build_hip_states, make_case, rel_l2, to_dev

pytestmark = pytest.mark.gpu


def _step_and_check(dev, case, pred_type, vae_scale, tag):
    from oracle import train_step as ots
    from stable_diffusion_training_amd import training_utils as tu
    torch.zeros(1, device=dev)  # (the allocator must exist before its statistics can be reset)
    torch.cuda.reset_peak_memory_stats(dev)
    tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev, prediction_type=pred_type, ema=True)
    w0 = us.store.master.clone()
    aux = {}
    out = tu.train_step(us, ts, ue, te, to_dev(case["batch"], dev), torch.Generator(device=dev), vae, sc,
                        strip_bos_eos_token=False, ema_rate=0.999, rand=to_dev(case["rand"], dev), aux=aux, vae_scale=vae_scale)
    loss = out[4]["loss"].item()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    gn_u, gn_t = us.store.grad_norm(), ts.store.grad_norm()
    print(f"[{tag}] loss {loss:.4f}  |g_unet| {gn_u:.4f}  |g_text| {gn_t:.4f}  HBM high-water {peak:.1f} GiB "
          f"({us.store.total / 1e6:.0f} M UNet + {ts.store.total / 1e6:.0f} M text parameters)")
    # ---- properties that hold at any size
    assert np.isfinite(loss) and 0.2 < loss < 5.0, loss           # unit-variance target, random-init prediction
    assert np.isfinite(gn_u) and gn_u > 0 and np.isfinite(gn_t) and gn_t > 0
    assert bool(torch.isfinite(us.store.grad_flat()).all()) and bool(torch.isfinite(ts.store.grad_flat()).all())
    # Lion: every weight-decayed parameter moves by lr * (+-1 + wd * p) exactly (sign(0) = 0 where the interpolated momentum is 0)
    lr, wd = us.hyper["lr"], us.hyper["wd"]
    lf = us.store.leaves["mid_block/resnets_0/conv1/kernel"]
    p0, p1 = w0[lf.offset: lf.offset + lf.numel], us.store.master[lf.offset: lf.offset + lf.numel]
    step = (p0 - p1 - lr * wd * p0) / lr
    assert float((step.abs() - 1).abs().max()) < 5e-2, "update is not -lr (sign(c) + wd p)"  # (fp32 ulp of p ~ lr / 100)
    codes = us.store.codes[lf.offset: lf.offset + lf.numel]
    assert int((codes != 3).sum()) > 0.9 * lf.numel, "8-bit momentum did not move off quantise(0) = 3"
    assert float((us.store.ema[lf.offset: lf.offset + lf.numel] - (0.999 * p0 + 0.001 * p1)).abs().max()) < 1e-6
    assert us.step == 1 and ts.step == 1
    # ---- forward parity against the fp32 oracle at full size
    with torch.no_grad():
        loss_ref, aux_ref = ots.compute_loss(case["weights"]["unet"], case["weights"]["clip"], case["weights"]["vae"],
                                             case["sched_state"], case["cfgs"], case["batch"], case["rand"],
                                             prediction_type=pred_type, vae_scale=vae_scale, return_aux=True)
    e_m, e_c = rel_l2(aux["moments"], aux_ref["moments"]), rel_l2(aux["ctx"], aux_ref["ctx"])
    e_p = rel_l2(aux["pred"][..., :4].permute(0, 3, 1, 2), aux_ref["pred"])
    print(f"[{tag}] vs fp32 oracle: moments {e_m:.2e}  context {e_c:.2e}  prediction {e_p:.2e}  loss {loss:.5f} / {float(loss_ref):.5f}")
    # bf16 tolerance: SURVEY.md §8(d) starts from rel-L2 <= 2e-2 on the prediction at SD1.5 size; these graphs are deeper (23 / 44
    # text layers, up to 70 transformer blocks) and their bf16 rounding-noise floor - the distance between two bf16 evaluations of
    # the network that round at different points, DESIGN.md section 2 - is 1.4 - 1.8e-2 already at SD1.5 size, so 3e-2 here;
    # |dloss|/loss <= 1e-2 as everywhere
    assert e_m < 3e-2 and e_c < 2e-2 and e_p < 3e-2
    assert abs(loss - float(loss_ref)) / float(loss_ref) < 1e-2


def test_sd21_768_train_step_at_size(dev):
    """configs[3]: latents 96x96 -> 9216 query/key tokens per head (d = 64) in the first level, zero-terminal-SNR v-prediction."""
    case = make_case("sd21", B=1, image=768, sched="zero_snr_scaled_linear")
    _step_and_check(dev, case, "v_prediction", 0.18215, "sd21-768")


def test_sdxl_1024_train_step_at_size(dev):
    """configs[4]: 2.57 B UNet parameters (text_time micro-conditioning, transformer depth 1/2/10), latents 128x128, two text towers,
    Lion-8bit state on everything but the excluded leaves."""
    case = make_case("sdxl", B=1, image=1024)
    assert sum(v.numel() for v in case["weights"]["unet"].values()) == 2_567_463_684
    _step_and_check(dev, case, "epsilon", 0.13025, "sdxl-1024")
