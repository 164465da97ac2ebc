"""The hand-off between a captured step and the gradient exchange that stays outside the graph (dp.ExchangePlan):
an event-record NODE inside a HIP graph must order a stream outside the graph behind the node's predecessors of THIS
launch - and only behind those, not behind the rest of the graph."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_external_event_node_orders_outside_stream():
    from stable_diffusion_training_amd import _lib
    _lib.require_device()
    dev = torch.device("cuda:0")
    ev = ctypes.c_void_p()
    _lib.call("sdt_event_create", ctypes.byref(ev))
    a = torch.zeros(1 << 16, device=dev)
    b = torch.zeros_like(a)
    big = torch.zeros(1 << 28, device=dev)  # 1 GiB: each fill is a ~0.3 ms kernel
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            big.fill_(1.0)
        a.add_(1.0)
        _lib.call("sdt_event_record", ev, 1, torch.cuda.current_stream().cuda_stream)
        for _ in range(60):
            big.fill_(2.0)
    torch.cuda.synchronize()
    gaps = []
    for it in range(4):
        # as dp.GradReducer: a high-priority stream (a default-priority one can share the graph's hardware queue).  Which hardware
        # queue the runtime gives a stream is its choice; a fresh stream per launch makes the overlap observable on most of them.
        side = torch.cuda.Stream(priority=-1)
        g.replay()
        _lib.call("sdt_stream_wait_event", side.cuda_stream, ev)
        with torch.cuda.stream(side):
            b.copy_(a)
            t_side = torch.cuda.Event(enable_timing=True)
            t_side.record()
        t_main = torch.cuda.Event(enable_timing=True)
        t_main.record()
        torch.cuda.synchronize()
        # ORDERING (what correctness rests on): the copy saw this launch's increment although ~6 ms of fills precede it in the graph
        assert float(b[0]) == it + 1 and float(b[-1]) == it + 1
        gaps.append(t_side.elapsed_time(t_main))
    # OVERLAP (a performance property): the outside stream did not wait for the ~18 ms of fills that follow the node.  It holds
    # whenever the runtime puts the side stream on a hardware queue of its own; when every launch of this process happened to
    # share the graph's queue there is nothing to assert about the library, so report instead of failing.
    _lib.call("sdt_event_destroy", ev)
    if max(gaps) <= 5.0:
        pytest.skip(f"ordering verified on 4 launches; the runtime serialised the side stream behind the graph (gaps {gaps} ms): overlap not observable here")


def test_external_event_wait_node_waits_for_the_latest_record():
    """The reverse hand-off (sdt_stream_wait_event_external): a wait NODE inside a captured graph must hold the graph's later nodes
    until the work another stream recorded into the event BEFORE this launch has finished - every launch anew (the sharded
    optimizer's all-gather of the weight mirrors, consumed by the next step behind its VAE encode)."""
    from stable_diffusion_training_amd import _lib
    _lib.require_device()
    dev = torch.device("cuda:0")
    ev = ctypes.c_void_p()
    _lib.call("sdt_event_create", ctypes.byref(ev))
    src = torch.zeros(1 << 16, device=dev)
    seen = torch.zeros_like(src)
    big = torch.zeros(1 << 28, device=dev)
    big2 = torch.zeros(1 << 28, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        big.fill_(1.0)                                   # something ahead of the wait (the VAE encode)
        _lib.call("sdt_stream_wait_event_external", torch.cuda.current_stream().cuda_stream, ev)
        seen.copy_(src)                                  # the first reader of what the other stream produces
    torch.cuda.synchronize()
    g.replay()                                           # never recorded: the node must not block
    torch.cuda.synchronize()
    assert float(seen[0]) == 0.0
    side = torch.cuda.Stream(priority=-1)
    for it in range(1, 5):
        with torch.cuda.stream(side):
            for _ in range(30):                          # ~10 ms of fills in front of the value: far longer than the graph's own prefix
                big2.fill_(float(it))
            src.fill_(float(it))
        _lib.call("sdt_event_record", ev, 0, side.cuda_stream)
        g.replay()
        torch.cuda.synchronize()
        assert float(seen[0]) == it and float(seen[-1]) == it, (it, float(seen[0]))
    _lib.call("sdt_event_destroy", ev)
