"""volta_amd/streams.py: helper streams are chosen by PROBING that they neither delay the engine's compute / side streams nor are delayed by
them (hardware queue and command-processor pipe sharing)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_independent_stream_is_clear_of_the_engine_streams():
    import volta_amd  # noqa: F401
    from volta_amd import streams as S
    torch.zeros(1, device="cuda")
    own, side = S.engine_streams()
    assert side and side != own, "the executor's side stream exists and is not the compute stream"
    s = S.independent_stream()
    assert s.cuda_stream not in (own, side)
    for victim in (own, side):
        busy, idle = S.active_cost(victim, s)
        assert busy <= 1.25 * idle + 50.0, ("a kernel on the chosen stream slows launches on an engine stream", victim, busy, idle)
    # the probe tells a stream that shares the compute stream's hardware queue from one that does not: the compute stream itself is the limit case
    busy, idle = S.active_cost(own, torch.cuda.ExternalStream(side))
    assert busy <= 1.25 * idle + 50.0, "the side stream must not share the compute stream's queue or pipe"


def test_gate_and_flag_order_two_streams_without_an_event():
    """vk_store_u64 on one stream releases vk_gate_value on another; a gate whose flag never comes gives up and raises the error word."""
    import ctypes as C
    from volta_amd import _lib as L
    from volta_amd import streams as S
    a = torch.cuda.Stream()
    b = S.independent_stream(avoid=[a])          # a gate on the hardware queue of the stream that releases it would wait for its own releaser
    words = torch.zeros(4, dtype=torch.int64, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(b):
        L.check(L.lib.vk_gate_value(C.c_void_p(words.data_ptr()), 7, 2000000, L.ptr(err), L.stream_ptr()))
        out.add_(1.0)
    with torch.cuda.stream(a):
        L.check(L.lib.vk_hold_cus(1, 300, 0, L.stream_ptr()))
        L.check(L.lib.vk_store_u64(C.c_void_p(words.data_ptr()), 7, L.stream_ptr()))
    torch.cuda.synchronize()
    assert float(out) == 1.0 and int(err) == 0 and int(words[0]) == 7
    with torch.cuda.stream(b):
        L.check(L.lib.vk_gate_value(C.c_void_p(words.data_ptr() + 8), 9, 2000, L.ptr(err), L.stream_ptr()))      # nobody stores 9: gives up after 2 ms
    torch.cuda.synchronize()
    assert int(err) == 2
