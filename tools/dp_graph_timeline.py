"""Developer tool: where the time of a split-graph data-parallel step goes.  One rank, RCCL group of size 1, exchange forced on;
prints host time and device time (HIP events) of graph A / the exchange / graph B per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29791")

import torch
import torch.distributed as dist

import bench
from stable_diffusion_training_amd import dp
from stable_diffusion_training_amd import training_utils as tu

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev, "pg_options": dp.rccl_group_options()} if backend == "nccl" else {}))
tc, cfgs, weights, (us, ts, ue, te, vae, sched, _) = bench.build_states(dev, 4)
red = dp.GradReducer([us.store, ts.store], force=True)
table = tu.dp_compile_all_unique_resolution(us, ts, ue, te, vae, sched, tc, reducer=red, per_device_batch=4, use_graph=True)
batch = bench.synthetic_batch(dev, 4, 0)
step = table[tuple(batch["pixel_values"].shape)]
rng = torch.Generator(device=dev)
rng.manual_seed(1)
for _ in range(4):
    step(us, ts, ue, te, batch, rng, vae, sched)
torch.cuda.synchronize()
assert step.graph_b is not None
mode = sys.argv[2] if len(sys.argv) > 2 else "rccl"
marks = []


def exchange_probe(plan):
    """run_exchange with a time stamp after every bucket on the communication stream (mode mul: a plain elementwise kernel
    stands in for the collective)."""
    from stable_diffusion_training_amd import _lib
    cs = red.comm_stream
    handles, stamps = [], []
    for ev, view, bk in plan.items:
        if ev is not None:
            _lib.call("sdt_stream_wait_event", cs.cuda_stream, ev)
        with torch.cuda.stream(cs):
            if mode == "mul":
                view.mul_(1.0)
            elif mode == "none":
                pass
            else:
                h = dist.all_reduce(view, op=red.op, async_op=True)
                h.wait()
            t = torch.cuda.Event(enable_timing=True)
            t.record()
            stamps.append(t)
    torch.cuda.current_stream().wait_stream(cs)
    marks.append(stamps)


rows = []
evs = []
for it in range(6):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    h = [time.perf_counter()]
    e[0].record()
    step.graph.replay()
    h.append(time.perf_counter())
    e[1].record()
    exchange_probe(step.plan)
    h.append(time.perf_counter())
    e[2].record()
    step.graph_b.replay()
    h.append(time.perf_counter())
    e[3].record()
    evs.append(e)
    rows.append(h)
t0 = time.perf_counter()
torch.cuda.synchronize()
print("final sync wait ms", (time.perf_counter() - t0) * 1e3)
for e, h in zip(evs, rows):
    print("host ms: A %.2f exch %.2f B %.2f | device ms: A %.2f exch %.2f B %.2f" % (
        (h[1] - h[0]) * 1e3, (h[2] - h[1]) * 1e3, (h[3] - h[2]) * 1e3,
        e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])))
print("bucket completion after graph A start, ms:", " ".join("%.1f" % evs[-1][0].elapsed_time(t) for t in marks[-1]))
for a, b in zip(evs[:-1], evs[1:]):
    print("step-to-step device ms %.2f" % a[0].elapsed_time(b[0]))
dist.destroy_process_group()
