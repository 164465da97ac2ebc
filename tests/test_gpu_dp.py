"""Data-parallel train_step on the GPU code path: two ranks share cuda:0 (the box has one GPU; RCCL refuses two ranks
on one device, so the collective backend is gloo on device tensors) - exercises the bucket reducer's stream/event
logic with the HIP kernels, checks that both ranks end with identical parameters and that the averaged gradient
equals the single-process gradient of the 2x larger batch."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _graph_worker(rank, world, port, q):
    """Captured step with the exchange between two graphs: ranks see different shards, so their parameters stay identical
    only if every bucket is all-reduced after ITS gradients are complete and before the optimizer graph reads them."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from stable_diffusion_training_amd import dp
        from stable_diffusion_training_amd import training_utils as tu
        from tests.helpers import build_hip_states, make_case, to_dev
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        case = make_case("tiny", B=2, image=64)
        sl = slice(rank, rank + 1)
        batch = to_dev({k: v[sl] for k, v in case["batch"].items()}, dev)
        rand = to_dev({k: v[sl] for k, v in case["rand"].items()}, dev)
        res = {}
        for mode in ("eager", "graph"):
            tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev, quantize=False)
            red = dp.GradReducer([us.store, ts.store], bucket_bytes=1 << 16)

            def bound(us, ts, ue, te, batch, rng, vae, sched, **extra):
                return tu.train_step(us, ts, ue, te, batch, rng, vae, sched, strip_bos_eos_token=False, reducer=red, **extra)

            step = tu._GraphedStep(bound, warmup=1, reducer=red) if mode == "graph" else bound
            rng = torch.Generator(device=dev)
            losses = []
            for _ in range(4):  # graph: 1 eager warm-up, then capture + 3 replays
                out = step(us, ts, None, None, batch, rng, vae, sc, rand=rand)
                losses.append(float(out[4]["loss"].item()))
            torch.cuda.synchronize()
            if mode == "graph":
                assert step.graph_b is not None and not step.disabled and len(step.plan.items) == len(red.buckets)
            res[mode] = (us.store.master.detach().cpu().numpy().copy(), ts.store.master.detach().cpu().numpy().copy(), losses)
        q.put((rank, "ok", res))
        dist.barrier()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + repr(e) + traceback.format_exc()[-1500:], None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _worker(rank, world, port, q, quantize=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from stable_diffusion_training_amd import dp
        from stable_diffusion_training_amd import training_utils as tu
        from tests.helpers import build_hip_states, make_case, to_dev
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        case = make_case("tiny", B=2, image=64)
        sl = slice(rank, rank + 1)
        batch = {k: v[sl] for k, v in case["batch"].items()}
        rand = {k: v[sl] for k, v in case["rand"].items()}
        tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev, quantize=quantize)
        assert (us.store.grad16 is not None) == quantize  # quantised stores keep (and exchange) the kernel gradients in bf16
        red = dp.GradReducer([us.store, ts.store], bucket_bytes=1 << 16)
        assert len(red.buckets) > 4
        out = tu.train_step(us, ts, None, None, to_dev(batch, dev), torch.Generator(device=dev), vae, sc,
                            strip_bos_eos_token=False, rand=to_dev(rand, dev), reducer=red)
        torch.cuda.synchronize()
        g = us.store.grad_flat().detach().cpu().clone()
        p = us.store.master.detach().cpu().numpy().copy()  # by value: a torch tensor travels as a shm handle the exiting worker may take with it
        loss = float(out[4]["loss"].item())
        if rank == 0:
            # single-process reference with the full batch
            tc2, (us2, ts2, _, _, vae2, sc2, _) = build_hip_states(case, dev, quantize=False)  # float32 gradients, no exchange
            out2 = tu.train_step(us2, ts2, None, None, to_dev(case["batch"], dev), torch.Generator(device=dev), vae2, sc2,
                                 strip_bos_eos_token=False, rand=to_dev(case["rand"], dev))
            torch.cuda.synchronize()
            g2 = us2.store.grad_flat().detach().cpu()
            # (a quantised store lays its leaves out in other segments than an unquantised one: compare leaf by leaf, same order)
            ga = torch.cat([g[lf.offset: lf.offset + lf.numel] for lf in us.store.leaves.values()])
            gb = torch.cat([g2[us2.store.leaves[pth].offset: us2.store.leaves[pth].offset + lf.numel] for pth, lf in us.store.leaves.items()])
            cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
            # per leaf (the gates of test_sd15_full_size_gradient_parity_per_leaf): cosine and relative norm of every kernel leaf of size
            worst_cos, worst_norm = 1.0, 0.0
            for pth, lf in us.store.leaves.items():
                lf2 = us2.store.leaves[pth]
                a, b = g[lf.offset: lf.offset + lf.numel], g2[lf2.offset: lf2.offset + lf2.numel]
                if lf.numel >= 256 and float(b.norm()) > 1e-3 * float(g2.norm()) / len(us.store.leaves) ** 0.5:
                    worst_cos = min(worst_cos, float(torch.dot(a, b) / (a.norm() * b.norm())))
                    worst_norm = max(worst_norm, abs(float(a.norm() / b.norm()) - 1.0))
            q.put((rank, "ok", p, loss, cos, float(out2[4]["loss"].item()), worst_cos, worst_norm))
        else:
            q.put((rank, "ok", p, loss, None, None, None, None))
        dist.barrier()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + repr(e) + traceback.format_exc()[-1500:], None, None, None, None, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("quantize", [False, True])
def test_dp_two_ranks_one_gpu(quantize):
    """quantize=True: the stores keep the kernel leaves' gradients in bf16 and the buckets are all-reduced IN bf16 (half the payload;
    the reference all-reduces float32-widened values, training_utils.py:709, 835-932).  Held against the float32, exchange-free
    gradient of the whole batch with the same per-leaf gates as the float32 exchange: a bf16 sum of two ranks is one more rounding
    of a value that already carries bf16 precision."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (1 if quantize else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, quantize)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    assert (res[0][2] == res[1][2]).all(), "ranks diverged after the optimizer step"
    assert abs(res[0][3] - res[1][3]) < 1e-6  # the reduced (mean) loss is identical on both ranks
    assert res[0][4] > 0.999, f"DP-averaged gradient vs full-batch gradient cosine {res[0][4]}"
    assert res[0][6] > 0.995 and res[0][7] < 0.03, f"worst kernel leaf: cosine {res[0][6]}, norm off by {res[0][7]} (bf16 exchange: {quantize})"
    # B=1 per rank and B=2 in one process take different GEMM tilings / split-K plans: bf16 rounding differs (512-element loss)
    assert abs(res[0][3] - res[0][5]) / res[0][5] < 1e-2


def test_dp_captured_step_two_ranks_one_gpu():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_graph_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    r0, r1 = res[0][2], res[1][2]
    for mode in ("eager", "graph"):
        assert (r0[mode][0] == r1[mode][0]).all() and (r0[mode][1] == r1[mode][1]).all(), f"{mode}: ranks diverged"
        assert np.allclose(r0[mode][2], r1[mode][2], rtol=0, atol=1e-6)
    # same data, same draws: the captured run follows the eager one (fp32 atomics order differs; Lion steps are +-lr)
    assert np.allclose(r0["graph"][2], r0["eager"][2], rtol=2e-2), (r0["graph"][2], r0["eager"][2])
    d = np.abs(r0["graph"][0] - r0["eager"][0])
    assert np.mean(d > 0) < 0.2, np.mean(d > 0)


def _rccl_one_rank_worker(port, q):
    """The RCCL side of the exchange on one GPU: a one-rank nccl group, the exchange forced on (the collectives really run:
    RCCL's one-rank reduce kernel), eager and as two graphs around it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from stable_diffusion_training_amd import dp
        from stable_diffusion_training_amd import training_utils as tu
        from tests.helpers import build_hip_states, make_case, to_dev
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=dp.rccl_group_options())
        case = make_case("tiny", B=2, image=64)
        batch, rand = to_dev(case["batch"], dev), to_dev(case["rand"], dev)
        res = {}
        for mode in ("plain", "eager", "graph", "plain_shard", "eager_shard", "graph_shard"):
            shard = mode.endswith("_shard")  # RCCL reduce-scatter / all-gather (in place on slices of the flat buffers) + sliced sweep
            tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev, quantize=shard)  # (8-bit state: scattered buckets exist)
            red = None if mode.startswith("plain") else dp.GradReducer([us.store, ts.store], bucket_bytes=1 << 16, force=True, shard=shard)
            if red is not None:
                assert red.active and red.native_avg and len(red.buckets) > 4 and red.shard == shard

            def bound(us, ts, ue, te, batch, rng, vae, sched, **extra):
                return tu.train_step(us, ts, ue, te, batch, rng, vae, sched, strip_bos_eos_token=False, reducer=red, **extra)

            step = tu._GraphedStep(bound, warmup=1, reducer=red) if mode.startswith("graph") else bound
            losses = []
            rng = torch.Generator(device=dev)  # a captured step is bound to the objects it was captured with
            for _ in range(3):
                out = step(us, ts, None, None, batch, rng, vae, sc, rand=rand)
                losses.append(float(out[4]["loss"].item()))
            torch.cuda.synchronize()
            if mode.startswith("graph"):
                assert step.graph_b is not None and not step.disabled and len(step.plan.items) == len(red.buckets)
            res[mode] = (losses, us.store.master.detach().cpu().numpy().copy())
        q.put(("ok", res))
        dist.barrier()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("ERR " + repr(e) + traceback.format_exc()[-1500:], None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_exchange_one_rank_eager_and_captured():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(37500 + (os.getpid() % 2000), q))
    p.start()
    status, res = q.get(timeout=600)
    p.join(120)
    assert status == "ok", status
    for mode in ("eager", "graph", "eager_shard", "graph_shard"):  # one rank: same trajectory as without the exchange
        base = res["plain_shard" if mode.endswith("_shard") else "plain"]
        assert np.allclose(res[mode][0], base[0], rtol=2e-2), (mode, res[mode][0], base[0])
        assert np.mean(np.abs(res[mode][1] - base[1]) > 0) < 0.2


def _shard_worker(rank, world, port, q):
    """Sharded optimizer against the replicated one, on the optimizer arithmetic itself: identical weights, per-rank gradients
    injected into the flat buffer (so both runs exchange and sweep exactly the same numbers), three steps of carried 8-bit state."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from stable_diffusion_training_amd import dp, nets, params
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        spec = nets.unet_spec(nets.unet_config("tiny"))
        weights = nets.init_params(spec, 1)
        # the reference's default exclusion list (reference training_utils.py:79): conv_in / conv_out - the zero-padded kernels whose
        # compute copies sit past the mirrored range - are quantised, i.e. scattered: their padded copies must follow the gathered
        # mirror on the ranks that do not own them (w below includes the padded slots)
        excl = ("bias", "scale", "embedding")
        res = {}
        for shard in (False, True):
            st = params.ParamStore(spec, device=dev, quantise=True, quant_excluded=excl, wd_excluded=("bias", "scale"), block_size=16, with_ema=True)
            st.load(weights)
            red = dp.GradReducer([st], bucket_bytes=1 << 16, shard=shard)
            assert red.shard == shard and len(red.buckets) > 4
            if shard:
                assert any(b["scatter"] for b in red.buckets) and any(not b["scatter"] for b in red.buckets)
                pieces = red.shard_pieces(st)[0]
                assert sum(b - a for a, b, q, d in pieces if q) * world == st.quant_total
                assert sum(b - a for a, b, q, d in pieces if not q) == st.total - st.quant_total
            for step in range(3):
                g = torch.Generator().manual_seed(1000 * step + rank)
                st.set_grad_flat(torch.randn(st.total, generator=g) * (0.3 if step else 1e-4))  # below, then above the clip norm
                red.begin_step()
                for p in st.leaves:
                    st.grad_ready(p)
                red.finish()
                st.optimizer_step(lr=1e-3, wd=0.07, ema_rate=0.999, shard=red.shard_pieces(st))
                red.after_optimizer()
                red.wait_gathered()  # the mirror all-gather runs on the communication stream: the next reader waits for it
                st.prepare()
            torch.cuda.synchronize()
            gn = st.grad_norm()
            red.gather_state()  # collective: masters / EMA / momentum of the scattered slices become whole on every rank
            res[shard] = {k: v.detach().cpu().numpy().copy() for k, v in
                          dict(master=st.master, ema=st.ema, codes=st.codes, inv=st.inv_scale, mom=st.mom, w=st.w.view(torch.int16)).items()}
            res[shard]["gnorm"] = gn
        q.put((rank, "ok", res))
        dist.barrier()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + repr(e) + traceback.format_exc()[-1500:], None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_sharded_optimizer_equals_replicated_two_ranks_one_gpu():
    """reduce-scatter + per-slice clip / Lion-8bit / EMA + all-gather of the bf16 mirrors gives, bit for bit, the
    parameters, int8 momentum codes, scales, EMA and compute copies of the replicated optimizer, on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 39500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    r0, r1 = res[0][2], res[1][2]
    for k in ("master", "ema", "codes", "inv", "mom", "w"):
        assert (r0[True][k] == r1[True][k]).all(), f"sharded: ranks differ in {k}"
        assert (r0[True][k] == r0[False][k]).all(), f"sharded != replicated in {k}"
    assert abs(r0[True]["gnorm"] - r0[False]["gnorm"]) <= 1e-6 * r0[False]["gnorm"] and r0[True]["gnorm"] == r1[True]["gnorm"]


def _shard_step_worker(rank, world, port, q):
    """The whole train_step with the sharded optimizer, eager and captured (graph A | exchange + slice norms | graph B | all-gather)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from stable_diffusion_training_amd import dp
        from stable_diffusion_training_amd import training_utils as tu
        from tests.helpers import build_hip_states, make_case, to_dev
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        case = make_case("tiny", B=2, image=64)
        sl = slice(rank, rank + 1)
        batch = to_dev({k: v[sl] for k, v in case["batch"].items()}, dev)
        rand = to_dev({k: v[sl] for k, v in case["rand"].items()}, dev)
        res = {}
        for mode in ("replicated", "eager", "graph"):
            tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev, quantize=True, ema=True, quant_excluded=["bias", "scale", "embedding"])
            assert us.store.leaves["conv_in/kernel"].quantised and us.store.leaves["conv_in/kernel"].w_off >= us.store.total
            red = dp.GradReducer([us.store, ts.store], bucket_bytes=1 << 16, shard=mode != "replicated")

            def bound(us, ts, ue, te, batch, rng, vae, sched, **extra):
                return tu.train_step(us, ts, ue, te, batch, rng, vae, sched, strip_bos_eos_token=False, ema_rate=0.999, reducer=red, **extra)

            step = tu._GraphedStep(bound, warmup=1, reducer=red) if mode == "graph" else bound
            rng = torch.Generator(device=dev)
            losses = []
            for _ in range(4):
                out = step(us, ts, ue, te, batch, rng, vae, sc, rand=rand)
                losses.append(float(out[4]["loss"].item()))
            torch.cuda.synchronize()
            if mode == "graph":
                assert step.graph_b is not None and not step.disabled and step.plan.post
            if mode != "replicated":
                # the export guard on both launch paths: a sharded sweep - Python call or graph replay - leaves the store
                # not whole; gather_state() (collective) makes it whole; the next step un-wholes it again; a load makes it whole
                for st in (us.store, ts.store):
                    assert st.sharded and not st.state_whole
                    with pytest.raises(RuntimeError, match="gather_state"):
                        st.export()
                red.gather_state()
                assert us.store.state_whole and ts.store.state_whole
                us.store.export()
                step(us, ts, ue, te, batch, rng, vae, sc, rand=rand)  # (graph mode: a replay)
                torch.cuda.synchronize()
                assert not us.store.state_whole and not ts.store.state_whole, mode
                with pytest.raises(RuntimeError, match="gather_state"):
                    us.store.export_host()
                red.gather_state()
                whole = us.store.export()
                us.store.load(whole, init_ema=False)  # every rank loads the whole tree: whole again, nothing pending
                assert us.store.state_whole
                us.store.export()
            snap = {}
            red.gather_state()
            for name, st in (("unet", us.store), ("text", ts.store)):
                snap[name] = (st.master.detach().cpu().numpy().copy(), st.codes.detach().cpu().numpy().copy(), st.w.view(torch.int16).detach().cpu().numpy().copy())
            res[mode] = (snap, losses)
        q.put((rank, "ok", res))
        dist.barrier()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + repr(e) + traceback.format_exc()[-1500:], None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_sharded_train_step_two_ranks_one_gpu():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 41500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_step_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    r0, r1 = res[0][2], res[1][2]
    for mode in ("replicated", "eager", "graph"):
        for name in ("unet", "text"):
            for a, b in zip(r0[mode][0][name], r1[mode][0][name]):
                assert (a == b).all(), f"{mode}/{name}: ranks hold different state after the gather"
        assert np.allclose(r0[mode][1], r1[mode][1], rtol=0, atol=1e-6)
    # same data and draws in every mode: the loss trajectories agree to the bf16 noise of the step (the exchange adds the two ranks'
    # gradients in another order than one rank's sums: a last bit of a bf16 weight, then the network's rounding noise)
    for mode in ("eager", "graph"):
        assert np.allclose(r0[mode][1], r0["replicated"][1], rtol=2e-2), (mode, r0[mode][1], r0["replicated"][1])
    # eager and replayed sharded steps are the same arithmetic (5 steps each, the export-guard step included)
    for name in ("unet", "text"):
        for a, b in zip(r0["eager"][0][name], r0["graph"][0][name]):
            assert (a == b).all(), f"{name}: eager and captured sharded steps differ"


def test_eight_way_slices_of_the_sharded_sweep_equal_the_replicated_sweep():
    """The 8-rank slicing of the sharded optimizer without a process group: eight 'virtual ranks' each sweep their slices of every
    scattered bucket (as GradReducer.shard_pieces cuts them for world = 8) on their own copy of the state; stitched together, the
    slices must equal the replicated sweep bit for bit - masters, 8-bit codes, scales, EMA, bf16 mirrors - over three carried steps.
    (What a multi-GPU node would add is only RCCL's in-place reduce-scatter / all-gather themselves.)"""
    from stable_diffusion_training_amd import _lib, nets, params
    _lib.require_device()
    dev = torch.device("cuda:0")
    world = 8
    spec = nets.unet_spec(nets.unet_config("tiny"))
    weights = nets.init_params(spec, 1)
    kw = dict(device=dev, quantise=True, quant_excluded=("bias", "scale", "embedding"), wd_excluded=("bias", "scale"), block_size=16, with_ema=True)
    ref = params.ParamStore(spec, **kw)
    ref.load(weights)
    ranks = []
    for r in range(world):
        st = params.ParamStore(spec, **kw)
        st.load(weights)
        st.sharded = True
        ranks.append(st)
    buckets = ref.shard_buckets(world, 1 << 16)
    assert sum(1 for a, b, q, d in buckets if q) > 4 and all((b - a) % (world * 256) == 0 for a, b, q, d in buckets if q)

    def pieces(r):
        out = []
        for a, b, q, d in buckets:
            n = (b - a) // world
            out.append((a + r * n, a + (r + 1) * n, q, d) if q else (a, b, q, d))
        return out

    names = ("master", "codes", "inv_scale", "ema", "w")
    for step in range(3):
        g = (torch.randn(ref.total, generator=torch.Generator().manual_seed(step)) * (0.3 if step else 1e-4)).to(dev)
        ref.set_grad_flat(g)
        ref.optimizer_step(lr=1e-3, wd=0.07, ema_rate=0.999)
        for r, st in enumerate(ranks):
            st.set_grad_flat(g)
            # the scattered part of the squared norm (GradReducer._shard_norms + its all-reduce): sum over every rank's slices
            st.sqnorm.zero_()
            for rr in range(world):
                for a, b, q, d in pieces(rr):
                    if q:
                        st.sqnorm_accumulate(a, b)
            st.optimizer_step(lr=1e-3, wd=0.07, ema_rate=0.999, shard=(pieces(r), True))
        # "all-gather": every rank's slices of every state buffer into every rank (what dp._gather_buffers moves over RCCL)
        for name in names:
            per = ref.block_size if name == "inv_scale" else 1
            for a, b, q, d in buckets:
                if not q:
                    continue
                n = (b - a) // world
                for r, src in enumerate(ranks):
                    lo, hi = (a + r * n) // per, (a + (r + 1) * n) // per
                    for dst in ranks:
                        if dst is not src:
                            getattr(dst, name)[lo:hi].copy_(getattr(src, name)[lo:hi])
        torch.cuda.synchronize()
        for st in ranks[:2] + ranks[-1:]:
            for name in names:
                a_, b_ = getattr(st, name), getattr(ref, name)
                n = ref.total if name != "inv_scale" else b_.numel()
                assert torch.equal(a_[:n].view(torch.int16 if name == "w" else a_.dtype), b_[:n].view(torch.int16 if name == "w" else b_.dtype)), (step, name)
            assert abs(st.grad_norm() - ref.grad_norm()) <= 1e-6 * ref.grad_norm()
