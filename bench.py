"""Headline benchmark: training images/sec of the SD1.5 512x512 bf16 DDPM train_step (BASELINE.json configs[1]:
batch 4 per MI355X; weak scaling, global batch 4*N) on synthetic 512x512 images + 77-token captions with
random-init weights of the SD1.5 UNet / VAE / CLIP-L architectures.

  python bench.py --gpus 1 --steps K --warmup W                          (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = the whole reference train_step (training_utils.py:504-762) on one batch: VAE encode -> posterior sample ->
noise/timestep draw -> add_noise -> CLIP text encoder -> UNet forward -> MSE -> backward through UNet + CLIP ->
(N>1: bucketed RCCL all-reduce overlapped with backward) -> clip_by_global_norm -> Lion-8bit -> EMA.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra legs (N=1, rank 0 only, outside the timed region):
  roofline      HIP-event timing (on the launch stream) of every launch of the dominant kernel family,
                gemm_nt_kernel (MFMA implicit-GEMM conv / linear fwd+dgrad), during one extra step
  cpu_baseline  the CPU oracle (fp32 PyTorch-CPU restatement of the same step, kind "port") timed on the host cores
                for ONE step at batch 1 (BASELINE.json configs[0]); the reference's own Flax/XLA path is not
                installable offline
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"

# --config: the BASELINE.json configurations that fit one GPU.  The default (the driver's line) is configs[1]; the other two
# are the long-sequence and the large-model configurations, benchmarked on request (same step, same timing contract).
# tflop: algorithmic TFLOP per image of the whole step (SURVEY.md §8(d) work table).
CONFIGS = {
    "sd15_512": dict(unet="sd15", clip="clip_l", image=512, batch=4, pred="epsilon", sched="scaled_linear", vae_scale=0.18215, tflop=3.57,
                     metric="training images/sec, SD1.5 512×512 bf16, ε-pred MSE parity; 1/2/4/8 GPUs",
                     label="SD1.5 512x512 train_step: VAE encode + CLIP-L fwd/bwd + UNet fwd/bwd + clip + Lion-8bit + EMA"),
    "sd21_768": dict(unet="sd21", clip="openclip_h", image=768, batch=4, pred="v_prediction", sched="zero_snr_scaled_linear",
                     vae_scale=0.18215, tflop=9.2,
                     metric="training images/sec, SD2.1-768 v-prediction 768×768 bf16 (BASELINE configs[3])",
                     label="SD2.1-768 train_step (v-prediction, zero-terminal-SNR, 9216-token self-attention): VAE encode + OpenCLIP-H "
                           "fwd/bwd + UNet fwd/bwd + clip + Lion-8bit + EMA"),
    "sdxl_1024": dict(unet="sdxl", clip="dual", image=1024, batch=2, pred="epsilon", sched="scaled_linear", vae_scale=0.13025, tflop=25.4,
                      metric="training images/sec, SDXL-base 1024×1024 bf16, Lion-8bit state (BASELINE configs[4])",
                      label="SDXL 1024x1024 train_step: VAE encode + CLIP-L and OpenCLIP-bigG fwd/bwd + UNet (text_time "
                            "micro-conditioning) fwd/bwd + clip + Lion-8bit + EMA"),
}
XGMI_PEAK_GBPS = 7 * 153.0         # 7 point-to-point links per GPU (MI355X_MICROARCH.md); SURVEY §8(d): busbw over 7 x 153 GB/s


def build_states(dev, per_gpu_batch, ema=True, config="sd15_512"):
    import torch
    from stable_diffusion_training_amd import nets
    from stable_diffusion_training_amd import training_utils as tu
    c = CONFIGS[config]
    cfgs = dict(unet=nets.unet_config(c["unet"]), vae=nets.vae_config("sd"),
                clip=nets.dual_clip_config() if c["clip"] == "dual" else nets.clip_config(c["clip"]))
    weights = dict(unet=nets.init_params(nets.unet_spec(cfgs["unet"]), 1), vae=nets.init_params(nets.vae_encoder_spec(cfgs["vae"]), 2),
                   clip=nets.init_params(nets.clip_text_spec(cfgs["clip"]), 3))
    tc = tu.TrainingConfig(
        model_path="synthetic-" + config, batch_size=per_gpu_batch, learning_rate=1e-6, unet_learning_rate=1e-6,
        text_encoder_learning_rate=1e-6, lr_scheduler="constant", adam_to_lion_scale_factor=7.0, compilation_cache_path="",
        keep_compiled_fn_in_cache=False, text_encoder_context_window=77, context_window_concatenation_count=1, aot_compile=True,
        strip_bos_eos_token=False, offset_noise_magnitude=0.0, min_snr_gamma_magnitude=0.0, perturbation_noise_magnitude=0.0,
        image_area_root=[c["image"]], minimum_axis_length=[c["image"]], beta_scheduler=c["sched"], prediction_type=c["pred"],
        excluded_layer_pattern_from_weight_decay=["bias", "scale", "embedding"],
        excluded_layer_from_quantization=["bias", "scale", "embedding", "conv_in", "conv_out", "time_embedding", "embeddings", "time_emb_proj"],
        quant_block_size=16, quantize_unet_state=True, quantize_text_encoder_state=True, accumulate_unet_ema=ema,
        accumulate_text_encoder_ema=ema, ema_rate=0.99998)
    models = {"unet": {"unet_params": weights["unet"], "config": cfgs["unet"]}, "vae": {"vae_params": weights["vae"], "config": cfgs["vae"]},
              "text_encoder": {"text_encoder_params": weights["clip"], "config": cfgs["clip"]}}
    states = tu.on_device_model_training_state(tc, models, device=dev)
    return tc, cfgs, weights, states


def synthetic_batch(dev, B, rank, config="sd15_512"):
    import torch
    c = CONFIGS[config]
    g = torch.Generator().manual_seed(1234 + rank)
    px = torch.rand(B, 3, c["image"], c["image"], generator=g) * 2 - 1
    dual = c["clip"] == "dual"
    ids = torch.randint(0, 49406, (B, 2, 77) if dual else (B, 77), generator=g, dtype=torch.int32)
    ids[..., 0] = 49406
    ids[..., -1] = 49407
    batch = {"pixel_values": px.to(dev), "input_ids": ids.to(dev), "attention_mask": torch.ones(B, 77, dtype=torch.int32, device=dev)}
    if dual:  # SDXL micro-conditioning (explicit synthetic inputs: SURVEY.md §8(d) note on configs[4])
        batch["text_embeds"] = torch.randn(B, 1280, generator=g).to(dev)
        batch["time_ids"] = torch.tensor([[c["image"], c["image"], 0, 0, c["image"], c["image"]]] * B, dtype=torch.int32).to(dev)
    return batch


def state_digest(stores):
    """sha256 over fp32 masters, 8-bit momentum codes, block scales, fp32 momenta and EMA of the stores, in flat-buffer order: what two
    runs that claim the same training trajectory must share bit for bit."""
    import hashlib
    h = hashlib.sha256()
    for st in stores:
        for name in ("master", "codes", "inv_scale", "mom", "ema"):
            t = getattr(st, name, None)
            if t is None:
                continue
            flat = t.detach().reshape(-1).view(__import__("torch").uint8)
            for i in range(0, flat.numel(), 1 << 28):  # 256 MiB host chunks
                h.update(flat[i: i + (1 << 28)].cpu().numpy().tobytes())
        h.update(str(int(st.count)).encode())
    return h.hexdigest()


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(weights, cfgs, full=False):
    """The CPU restatement of the reference train_step (oracle/, kind "port": the reference's own Flax/XLA path is not installable
    offline) timed on this box's host cores at BASELINE configs[0]: SD1.5, 512x512 (64x64 latent), batch 1.
    Default: ONE fp32 step, no warm-up - a bounded sample (~1 min of CPU work) that keeps the default command within minutes.
    full (--cpu-baseline-full): SURVEY.md §8(d)'s protocol - 1 warm-up + 3 timed steps, fp32 and bf16-autocast variants (takes
    ~10 minutes; its result for this round is committed under profiles/)."""
    import torch
    from oracle import schedulers as osched
    from oracle import train_step as ots
    g = torch.Generator().manual_seed(7)
    batch = dict(pixel_values=torch.rand(1, 3, 512, 512, generator=g) * 2 - 1, input_ids=torch.randint(0, 49406, (1, 77), generator=g))
    rand = dict(posterior_eps=torch.randn(1, 64, 64, 4, generator=g), noise=torch.randn(1, 4, 64, 64, generator=g),
                timesteps=torch.randint(0, 1000, (1,), generator=g))
    cores = torch.get_num_threads()
    sched = osched.create_state("scaled_linear")

    def one(autocast):
        t0 = time.time()
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = ots.train_step(weights["unet"], weights["clip"], weights["vae"], sched, cfgs, batch, rand, dict(ots.DEFAULT_OPT))
        dt = time.time() - t0
        if full:  # progress for the ~10-minute protocol (a silent run is taken to be hung)
            print(f"[cpu_baseline] {'bf16 autocast' if autocast else 'fp32'} step: {dt:.1f} s", file=sys.stderr, flush=True)
        return dt, out["loss"]

    res = {"unit": "images/sec", "cores": cores, "cpu": _cpu_model(), "kind": "port",
           "note": "CPU restatement of the reference train_step (PyTorch-CPU), not JAX/XLA"}
    if not full:
        dt, loss = one(False)
        res.update(value=1.0 / dt, sample=f"1 fp32 oracle train_step (no warm-up), SD1.5 512x512, batch 1 (BASELINE configs[0]), {dt:.1f} s, loss {loss:.4f}")
        return res
    variants = {}
    for name, ac in (("fp32", False), ("bf16_autocast", True)):
        one(ac)  # warm-up
        times = [one(ac)[0] for _ in range(3)]
        variants[name] = {"images_per_sec": 1.0 / (sum(times) / 3), "step_s": times}
    res.update(value=variants["fp32"]["images_per_sec"], variants=variants,
               sample="1 warm-up + 3 timed oracle train_steps per variant (fp32, bf16 autocast), SD1.5 512x512, batch 1 (SURVEY.md 8(d))")
    return res


def _latest_profile(suffix):
    """The newest committed profiles/rNN_<suffix> (bench.py cannot read hardware counters or rocprofv3 output itself)."""
    import glob
    hits = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_" + suffix)))
    return hits[-1] if hits else None


def pmc_traffic(kernel_prefixes):
    """Average HBM bytes per launch of the dominant kernel family from the committed PMC passes (profiles/, collected with
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in runs of their own)."""
    path = _latest_profile("pmc_traffic.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError, TypeError):
        return None, None
    tot = n = 0.0
    for name, v in k.items():
        if name.startswith(kernel_prefixes):
            tot += v["total_bytes_per_launch"] * v["launches"]
            n += v["launches"]
    return (tot / n if n else None), "profiles/" + os.path.basename(path)


def rocprof_roofline():
    """The same fraction from rocprofv3's kernel durations of the committed profile (tools/make_profile_summary.py)."""
    path = _latest_profile("roofline_rocprof.json")
    try:
        with open(path) as f:
            r = json.load(f)
        return r["frac"], "profiles/" + os.path.basename(path)
    except (OSError, ValueError, KeyError, TypeError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="sd15_512", help="BASELINE.json configuration (default: configs[1], the metric's)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 4 for SD1.5 / SD2.1, 2 for SDXL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="SURVEY 8(d) protocol: 1 warm-up + 3 timed steps, fp32 and bf16 autocast (~10 min)")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.batch = args.batch or cfg["batch"]

    import torch
    import torch.distributed as dist
    from stable_diffusion_training_amd import _lib, dp, ops
    from stable_diffusion_training_amd import training_utils as tu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    _lib.require_device()
    # developer rehearsal of the N>1 launch on a one-GPU box: SDT_BENCH_BACKEND=gloo SDT_BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and exchanges through gloo (RCCL refuses two ranks on one device); never used by the driver's runs
    backend = os.environ.get("SDT_BENCH_BACKEND", "nccl")
    if os.environ.get("SDT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SDT_DP_FORCE (developer switch, one-rank torchrun launch): 1 = the multi-rank launch structure (two graphs around the exchange,
    # event nodes, stream waits) without the identity collectives; 2 = with RCCL's one-rank all-reduce kernels as well
    force_dp = os.environ.get("SDT_DP_FORCE") in ("1", "2") and "RANK" in os.environ
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=dp.rccl_group_options())
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    bucket_mb = int(os.environ.get("SDT_DP_BUCKET_MB", "96"))
    multi = world > 1 or force_dp
    # N > 1 [r4]: BOTH exchange modes are timed in one invocation - the all-reduce + replicated optimizer sweep (the reference's
    # semantics, training_utils.py:835-932), then the sharded optimizer (reduce-scatter + sliced sweep + all-gather of the bf16
    # mirrors, dp.GradReducer) when the world size can be sliced (dp.shardable_world) and the in-place collective forms it needs
    # pass their self-test - from the same weights, batches and generator seeds; the headline is the faster one, both are in
    # `exchange_modes`, and the two final states are compared: a digest of fp32 masters, 8-bit codes, scales and EMA must be
    # IDENTICAL ON EVERY RANK inside a mode (a hard error otherwise: replicas that drift apart are not data parallelism), and
    # is reported across the modes (`digest_match`, `final_loss_rel_diff`; the modes average the ranks' gradients with different RCCL
    # collectives and add the norm in another order, so across modes nothing is fatal unless SDT_BENCH_STRICT=1).
    # SDT_DP_SHARD=0 / 1 runs only that mode.
    shard_env = os.environ.get("SDT_DP_SHARD")
    modes = [False]
    shard_note = None
    if multi:
        if shard_env == "1":
            modes = [True]
        elif shard_env == "0":
            modes = [False]
        elif not dp.shardable_world(world):
            shard_note = f"world size {world} cannot be sliced (8 / world must be whole): all-reduce only"
        elif not dp.inplace_collectives_ok(dev):
            shard_note = "in-place reduce-scatter / all-gather self-test failed on this node: all-reduce only"
        else:
            modes = [False, True]
    batch = synthetic_batch(dev, args.batch, rank, args.config)

    def run_mode(shard):
        tc, cfgs, weights, (us, ts, ue, te, vae, sched, _) = build_states(dev, args.batch, config=args.config)
        reducer = (dp.GradReducer([us.store, ts.store], bucket_bytes=bucket_mb << 20, force=force_dp, shard=shard,
                                  skip_self=os.environ.get("SDT_DP_FORCE") == "1") if multi else None)
        table = tu.dp_compile_all_unique_resolution(us, ts, ue, te, vae, sched, tc, reducer=reducer, per_device_batch=args.batch,
                                                    step_overrides={"vae_scale": cfg["vae_scale"]})
        step_fn = table[tuple(batch["pixel_values"].shape)]
        rng = torch.Generator(device=dev)
        rng.manual_seed(1000 + rank)
        st = dict(us=us, ts=ts, ue=ue, te=te, rng=rng)

        def run(n, fn=None, note=None):
            loss = None
            for i in range(n):
                st["us"], st["ts"], st["ue"], st["te"], metrics, st["rng"] = (fn or step_fn)(st["us"], st["ts"], st["ue"], st["te"], batch, st["rng"], vae, sched)
                loss = metrics["loss"]
                if note and multi and rank == 0:  # untimed phases of a multi-rank run: a line per step (a gloo rehearsal takes a minute per step)
                    print(f"[bench] {'sharded' if shard else 'all-reduce'} {note} step {i + 1}/{n}", file=sys.stderr, flush=True)
            return loss

        # single-process runs replay the step as one HIP graph; building it (2 eager steps that size the workspaces + the capture)
        # is set-up, like the reference's per-resolution jit compile, and is kept out of the W warm-up / K timed steps
        graphed = isinstance(step_fn, tu._GraphedStep)
        setup_steps = step_fn.warmup + 1 if graphed else 0
        run(setup_steps, note="set-up")
        run(args.warmup, note="warm-up")
        if reducer is not None:
            reducer.timing = []  # HIP events around each timed step's exchange (two records per step)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = run(args.steps)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if dist.is_initialized():
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        out = dict(dt=float(tmax.item()), loss=float(loss.item()), graphed=graphed, setup_steps=setup_steps, reducer=reducer, run=run,
                   step_fn=step_fn, us=us, ts=ts, weights=weights, cfgs=cfgs, peak=torch.cuda.max_memory_allocated(dev) / 2 ** 30)
        if multi:
            if rank == 0:
                print(f"[bench] {'sharded' if shard else 'all-reduce'}: {args.steps} timed steps in {out['dt']:.2f} s; comparing the ranks' states", file=sys.stderr, flush=True)
            if reducer.timing:
                out["exchange_times"] = reducer.exchange_times_ms()
            reducer.timing = None
            reducer.gather_state()  # collective (a no-op for the replicated optimizer): every rank holds the whole state
            torch.cuda.synchronize()
            out["digest"] = state_digest([us.store, ts.store])
            every = [None] * world
            dist.all_gather_object(every, (out["digest"], out["loss"], rank, torch.cuda.current_device()))
            out["ranks_agree"] = all(e[0] == every[0][0] for e in every)
            out["rank_devices"] = [e[3] for e in sorted(every, key=lambda e: e[2])]
            if not out["ranks_agree"] and os.environ.get("SDT_BENCH_ALLOW_DRIFT") != "1":
                raise SystemExit(f"[bench] exchange mode {'sharded' if shard else 'all-reduce'}: the ranks' parameter / optimizer state differ after "
                                 f"{setup_steps + args.warmup + args.steps} steps (digests {[e[0][:12] for e in every]}): the gradient exchange is broken")
        return out

    def exchange_info(o):
        reducer, us, ts = o["reducer"], o["us"], o["ts"]
        payload = us.store.grad_bytes() + ts.store.grad_bytes()  # kernel gradients in bf16, the rest float32 (ParamStore.grad16)
        wire = 2.0 * (world - 1) / world * payload
        if reducer.shard:  # scattered part: reduce-scatter of the kernel gradients + all-gather of bf16 mirrors; the rest all-reduced
            quant = us.store.quant_total + ts.store.quant_total
            gq = sum((2 if st.grad16 is not None else 4) * st.quant_total for st in (us.store, ts.store))
            wire = (world - 1) / world * (gq + 2 * quant) + 2.0 * (world - 1) / world * (payload - gq)
        info = {"mode": "reduce-scatter + sharded optimizer + all-gather" if reducer.shard else "all-reduce",
                "ms_per_step": 1000.0 * o["dt"] / args.steps, "final_loss": o["loss"], "state_digest": o.get("digest"),
                "ranks_agree": o.get("ranks_agree"), "payload_bytes": payload, "buckets": len(reducer.buckets), "wire_bytes_per_gpu": wire,
                "xgmi_peak_GBps": XGMI_PEAK_GBPS}
        if o.get("exchange_times"):
            # the gradient exchange of the timed steps on this rank: the span from the first bucket's collective to the last one's end
            # (buckets wait for their gradients, so busbw is a LOWER bound), and what the compute stream still waited for after the
            # backward (the exposed part)
            span_ms, exposed_ms = o["exchange_times"]
            busbw = wire / (span_ms * 1e-3) / 1e9 if span_ms > 0 else 0.0
            info.update(span_ms=span_ms, exposed_ms=exposed_ms, busbw_GBps_lower_bound=busbw, frac_lower_bound=busbw / XGMI_PEAK_GBPS)
        return info

    runs = []
    for shard in modes:
        o = run_mode(shard)
        runs.append(o)
        if len(modes) > 1 and shard is not modes[-1]:  # free the first mode's states before the second is built
            o["info"] = exchange_info(o)
            for k in ("reducer", "run", "step_fn", "us", "ts"):
                o.pop(k)
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    best = min(runs, key=lambda o: o["dt"])
    last = runs[-1]
    dt, loss_val, graphed, setup_steps = best["dt"], best["loss"], best["graphed"], best["setup_steps"]
    weights, cfgs = last["weights"], last["cfgs"]
    run, step_fn = last.get("run"), last.get("step_fn")

    result = None
    if rank == 0:
        gb = args.batch * world
        result = {
            "metric": cfg["metric"],
            "value": gb * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{cfg['label']}, batch {args.batch}/GPU, 77-token captions, random-init weights",
                       "name": args.config, "global_batch": gb, "latent": f"{cfg['image'] // 8}x{cfg['image'] // 8}x4", "parallelism": f"dp{world}",
                       "step_tflop_per_image": cfg["tflop"], "step_mfma_frac": cfg["tflop"] * gb * args.steps / dt / MFMA_BF16_PEAK_TFLOPS / world,
                       "peak_hbm_GiB": max(o["peak"] for o in runs),
                       "launch": "hip_graph" if graphed else "eager", "setup_steps": setup_steps},
            "final_loss": loss_val,
        }
        if multi:
            infos = [o["info"] if "info" in o else exchange_info(o) for o in runs]
            result["exchange"] = infos[runs.index(best)]
            result["exchange_modes"] = {("sharded" if "sharded" in i["mode"] else "all_reduce"): i for i in infos}
            result["dist"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rank_devices": last["rank_devices"],
                              "sharded_mode_note": shard_note}
            if len(runs) == 2:
                la, lb = runs[0]["loss"], runs[1]["loss"]
                result["exchange_modes"]["digest_match"] = runs[0]["digest"] == runs[1]["digest"]
                result["exchange_modes"]["final_loss_rel_diff"] = abs(la - lb) / max(abs(la), 1e-30)
    if multi and len(runs) == 2:
        # across the modes the bits need not agree (another order of double additions in the norm, possibly another summation order inside
        # RCCL's reduce-scatter than inside its all-reduce), and bf16 training steps amplify a last bit: the loss difference is REPORTED
        # (exchange_modes.final_loss_rel_diff) and only SDT_BENCH_STRICT=1 turns a difference beyond 1e-3 into a failure
        la, lb = runs[0]["loss"], runs[1]["loss"]
        if abs(la - lb) > 1e-3 * max(abs(la), 1e-30):
            msg = f"[bench] the two exchange modes disagree: final loss {la} (all-reduce) vs {lb} (sharded)"
            if os.environ.get("SDT_BENCH_STRICT") == "1":
                raise SystemExit(msg)
            if rank == 0:
                print(msg + " - reported, not fatal (SDT_BENCH_STRICT=1 makes it fatal)", file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_roofline:
        ops.GEMM_NT_TIMER = ops.KernelTimer()
        ops.GEMM_TN_TIMER = ops.KernelTimer()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(1, step_fn.fn if graphed else None)  # eager: every GEMM launch bracketed by HIP events on its stream
        torch.cuda.synchronize()
        inst_ms = 1000 * (time.perf_counter() - t1)
        nt, tn = ops.GEMM_NT_TIMER.summary(), ops.GEMM_TN_TIMER.summary()
        if os.environ.get("SDT_BENCH_SHAPES"):
            with open(os.environ["SDT_BENCH_SHAPES"], "w") as f:
                for name, tm in (("nt", ops.GEMM_NT_TIMER), ("tn", ops.GEMM_TN_TIMER)):
                    for shape, n, ms, tf in tm.by_shape():
                        f.write(f"{name} {shape} calls={n} ms={ms:.3f} TF={tf:.1f}\n")
        ops.GEMM_NT_TIMER = ops.GEMM_TN_TIMER = None
        # headline: the RAW HIP-event time of every launch (each reading includes the ~5 us the event pair itself takes, so it
        # under-states the kernels: rocprofv3's kernel durations of the same build, profiles/, sit between the two figures);
        # the pair-overhead-corrected figure is kept beside it as *_calibrated
        ach = nt["flops"] / (nt["raw_ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(("gemm_nt_kernel", "conv3x3_halo_kernel"))
        result["roofline"] = {"bound": "mfma", "kernel": "sdt_gemm_nt_bf16 (gemm_nt_kernel + conv3x3_halo_kernel)", "achieved": ach,
                              "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                              "frac_rocprof": rocprof_roofline()[0], "frac_rocprof_source": rocprof_roofline()[1],
                              "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                              "traffic_measured_in_run": False,
                              "launches_per_step": nt["launches"], "avg_launch_us": 1000 * nt["raw_ms"] / max(nt["launches"], 1),
                              "kernel_ms_per_step": nt["raw_ms"], "achieved_calibrated": nt["flops"] / (nt["ms"] * 1e-3) / 1e12,
                              "kernel_ms_per_step_calibrated": nt["ms"],
                              "event_pair_overhead_us": nt["event_overhead_us"], "algorithmic_tflop_per_step": nt["flops"] / 1e12,
                              "instrumented_step_ms": inst_ms,
                              "wgrad_kernel": {"kernel": "gemm_tn_kernel + conv_wgrad3_kernel", "achieved": tn["flops"] / (tn["raw_ms"] * 1e-3) / 1e12,
                                               "achieved_calibrated": tn["flops"] / (tn["ms"] * 1e-3) / 1e12,
                                               "kernel_ms_per_step": tn["raw_ms"], "launches_per_step": tn["launches"]}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "sd15_512":
        for o in runs:
            for k in ("reducer", "run", "step_fn", "us", "ts"):
                o.pop(k, None)
        run = step_fn = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        result["cpu_baseline"] = cpu_baseline(weights, cfgs, full=args.cpu_baseline_full)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
