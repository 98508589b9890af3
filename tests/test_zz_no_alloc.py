"""The allocation-free dB post-pass (sed_db_normalize_static) against the entry point it is a variant of
(sed_db_normalize), through the C ABI, bit for bit, on two calls in a row, the second with a smaller input."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sound-event-detection_amd"))

pytestmark = pytest.mark.gpu


def _both(lib, x, groups, elems, subtract_max, top_db, stream):
    a, b = x.clone(), x.clone()
    rc = lib.sed_db_normalize(C.c_void_p(a.data_ptr()), groups, elems, subtract_max, C.c_float(top_db), C.c_void_p(stream))
    assert rc == 0
    rc = lib.sed_db_normalize_static(C.c_void_p(b.data_ptr()), groups, elems, subtract_max, C.c_float(top_db), C.c_void_p(stream))
    assert rc == 0
    torch.cuda.synchronize()
    return a, b


def test_static_db_normalize_equals_the_allocating_entry_point():
    import sed_amd._lib as L
    lib = L.lib
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device=dev).manual_seed(5)
    # first call: 3 groups of 1 000 003 floats (62 pieces each, the two-pass form in both entry points)
    x = -70.0 * torch.rand((3, 1000003), generator=g, device=dev)
    x[1, 777777] = 3.5
    a, b = _both(lib, x, 3, 1000003, 1, 80.0, stream)
    assert torch.equal(a, b)
    want = np.maximum(x.cpu().numpy() - x.cpu().numpy().max(axis=1, keepdims=True), -80.0)
    assert np.array_equal(a.cpu().numpy(), want.astype(np.float32))
    # second call, smaller: one group of 5000 floats, top_db only (the per-group kernel in the allocating entry point);
    # the scratch still holds the first call's partial maxima
    y = -90.0 * torch.rand((5000,), generator=g, device=dev)
    a, b = _both(lib, y, 1, 5000, 0, 30.0, stream)
    assert torch.equal(a, b)
    assert np.array_equal(a.cpu().numpy(), np.maximum(y.cpu().numpy(), y.cpu().numpy().max() - np.float32(30.0)))
    # a third one in between sizes, many small groups, no floor
    z = -50.0 * torch.rand((40, 70001), generator=g, device=dev)
    a, b = _both(lib, z, 40, 70001, 1, -1.0, stream)
    assert torch.equal(a, b)
    # beyond the scratch: a clear error, nothing written
    big = torch.zeros(8, device=dev)
    rc = lib.sed_db_normalize_static(C.c_void_p(big.data_ptr()), 20000, 16385, 1, C.c_float(80.0), C.c_void_p(stream))
    assert rc == -3
    assert "capacity" in L.lib.sed_status_string(rc).decode()


def test_static_db_normalize_under_graph_capture():
    """After one eager call on the device (which looks the scratch address up), the entry point is stream operations
    only: captured in a HIP graph and replayed on new data it gives the eager result."""
    import sed_amd._lib as L
    lib = L.lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9)
    buf = -60.0 * torch.rand((2, 300001), generator=g, device=dev)
    warm = buf.clone()
    assert lib.sed_db_normalize_static(C.c_void_p(warm.data_ptr()), 2, 300001, 1, C.c_float(80.0),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 0
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = lib.sed_db_normalize_static(C.c_void_p(buf.data_ptr()), 2, 300001, 1, C.c_float(80.0),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0
    for seed in (1, 2):
        fresh = -60.0 * torch.rand((2, 300001), generator=torch.Generator(device=dev).manual_seed(seed), device=dev)
        fresh[seed - 1, 12345 * seed] = 1.25
        buf.copy_(fresh)
        graph.replay()
        torch.cuda.synchronize()
        want = fresh.clone()
        assert lib.sed_db_normalize(C.c_void_p(want.data_ptr()), 2, 300001, 1, C.c_float(80.0),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 0
        torch.cuda.synchronize()
        assert torch.equal(buf, want)
