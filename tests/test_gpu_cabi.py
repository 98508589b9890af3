"""The C ABI called directly through ctypes on the GPU box: status codes for every class of bad call, and the stream /
plan-sharing contract (asynchronous on the caller's stream, plans immutable and shareable between streams)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    import sed_amd
    from sed_amd import _lib as L
    assert torch.cuda.is_available()
    return sed_amd, L


def _plan(L, cfg):
    h = C.c_void_p()
    c = cfg.to_c()
    rc = L.lib.sed_plan_create(C.byref(c), C.byref(h))
    return rc, h


def test_status_codes_of_bad_calls():
    sed_amd, L = _lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    good = sed_amd.FrontendConfig.htsat()
    rc, plan = _plan(L, good)
    assert rc == 0 and plan.value
    # plan creation
    import dataclasses
    for bad, want in ((dict(n_fft=768, win_length=768), L.SED_ERR_UNSUPPORTED_NFFT), (dict(win_length=2000), L.SED_ERR_BAD_ARG),
                      (dict(hop=0), L.SED_ERR_BAD_ARG), (dict(hop=1025), L.SED_ERR_BAD_ARG), (dict(n_mels=0), L.SED_ERR_BAD_ARG),
                      (dict(n_mels=129), L.SED_ERR_BAD_ARG), (dict(fmin=9000.0, fmax=100.0), L.SED_ERR_BAD_ARG),
                      (dict(spectrum=7), L.SED_ERR_BAD_ARG), (dict(pcm_kind=1, pcm_divisor=0.0), L.SED_ERR_BAD_ARG),
                      (dict(preemphasis=0.97), L.SED_ERR_BAD_ARG)):           # per-frame conditioning needs uncentred frames
        rc, h = _plan(L, dataclasses.replace(good, **bad))
        assert rc == want and not h.value, bad
    assert L.lib.sed_plan_create(None, None) == L.SED_ERR_BAD_ARG
    # frame counts and short input
    assert L.lib.sed_plan_num_frames(plan, 320000) == 1001
    assert L.lib.sed_plan_num_frames(plan, 512) == L.SED_ERR_SHORT_INPUT       # reflect padding needs L > n_fft / 2
    assert L.lib.sed_plan_num_frames(plan, -1) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_plan_num_frames(None, 1000) == L.SED_ERR_BAD_ARG
    # forward
    x = torch.zeros(2, 4096, device=dev)
    y = torch.empty(2, 13, 64, device=dev)
    s = L.stream_ptr(dev)
    assert L.lib.sed_logmel_forward(plan, x.data_ptr(), 2, 4096, 4096, y.data_ptr(), 13 * 64, s) == 0
    assert L.lib.sed_logmel_forward(None, x.data_ptr(), 2, 4096, 4096, y.data_ptr(), 13 * 64, s) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_logmel_forward(plan, None, 2, 4096, 4096, y.data_ptr(), 13 * 64, s) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_logmel_forward(plan, x.data_ptr(), 2, 4096, 4096, None, 13 * 64, s) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_logmel_forward(plan, x.data_ptr(), -1, 4096, 4096, y.data_ptr(), 13 * 64, s) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_logmel_forward(plan, x.data_ptr(), 2, 4096, 4000, y.data_ptr(), 13 * 64, s) == L.SED_ERR_BAD_ARG   # rows overlap
    assert L.lib.sed_logmel_forward(plan, x.data_ptr(), 2, 4096, 4096, y.data_ptr(), 100, s) == L.SED_ERR_BAD_ARG       # output rows overlap
    assert L.lib.sed_logmel_forward(plan, x.data_ptr(), 2, 300, 4096, y.data_ptr(), 13 * 64, s) == L.SED_ERR_SHORT_INPUT
    assert L.lib.sed_logmel_forward(plan, x.data_ptr(), 0, 4096, 4096, y.data_ptr(), 13 * 64, s) == 0                   # empty batch
    # segments of a resident buffer, zero-padded rows and rows of their own; the image from rows of their own
    off = torch.tensor([0, 4096], dtype=torch.int64, device=dev)
    ln = torch.tensor([4096, 1000], dtype=torch.int32, device=dev)
    for entry in (L.lib.sed_logmel_forward_segments, L.lib.sed_logmel_forward_segments_own):
        assert entry(plan, x.data_ptr(), off.data_ptr(), ln.data_ptr(), 2, 4096, y.data_ptr(), 13 * 64, s) == 0
        assert entry(plan, x.data_ptr(), None, ln.data_ptr(), 2, 4096, y.data_ptr(), 13 * 64, s) == L.SED_ERR_BAD_ARG
        assert entry(plan, x.data_ptr(), off.data_ptr(), None, 2, 4096, y.data_ptr(), 13 * 64, s) == L.SED_ERR_BAD_ARG
        assert entry(None, x.data_ptr(), off.data_ptr(), ln.data_ptr(), 2, 4096, y.data_ptr(), 13 * 64, s) == L.SED_ERR_BAD_ARG
        assert entry(plan, x.data_ptr(), off.data_ptr(), ln.data_ptr(), 2, 4096, y.data_ptr(), 100, s) == L.SED_ERR_BAD_ARG
        assert entry(plan, x.data_ptr(), off.data_ptr(), ln.data_ptr(), 0, 4096, y.data_ptr(), 13 * 64, s) == 0          # no segment
    torch.cuda.synchronize()
    assert not y[1, 4:].any() and y[1, :4].abs().sum() > 0                      # 1000 samples: 1 + 1000 // 320 rows of its own
    fr = torch.tensor([13, 4], dtype=torch.int32, device=dev)
    img = torch.empty(2, 1, 64, 16, device=dev)
    assert L.lib.sed_htsat_wav2img_ragged(y.data_ptr(), fr.data_ptr(), 2, 13, 64, 16, 1, img.data_ptr(), s) == 0
    assert L.lib.sed_htsat_wav2img_ragged(y.data_ptr(), None, 2, 13, 64, 16, 1, img.data_ptr(), s) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_htsat_wav2img_ragged(y.data_ptr(), fr.data_ptr(), 2, 13, 64, 12, 1, img.data_ptr(), s) == L.SED_ERR_BAD_ARG   # fewer columns than rows
    assert L.lib.sed_htsat_wav2img_ragged(y.data_ptr(), fr.data_ptr(), 2, 13, 64, 16, 3, img.data_ptr(), s) == L.SED_ERR_BAD_ARG   # 16 % 3
    assert L.lib.sed_htsat_wav2img_ragged(None, fr.data_ptr(), 2, 13, 64, 16, 1, img.data_ptr(), s) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_htsat_wav2img_ragged(y.data_ptr(), fr.data_ptr(), 0, 13, 64, 16, 1, img.data_ptr(), s) == 0
    # decoders
    p = torch.rand(3, 50, 2, device=dev)
    runs = torch.empty(3, 2, 8, 2, dtype=torch.int32, device=dev)
    counts = torch.empty(3, 2, dtype=torch.int32, device=dev)
    assert L.lib.sed_threshold_runs(p.data_ptr(), 3, 50, 2, 0.5, 0, runs.data_ptr(), counts.data_ptr(), 8, s) == 0
    assert L.lib.sed_threshold_runs(p.data_ptr(), 3, 50, 2, 0.5, 9, runs.data_ptr(), counts.data_ptr(), 8, s) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_threshold_runs(p.data_ptr(), 3, 50, 2, 0.5, 0, runs.data_ptr(), counts.data_ptr(), 0, s) == L.SED_ERR_BAD_ARG
    assert L.lib.sed_threshold_runs(None, 3, 50, 2, 0.5, 0, runs.data_ptr(), counts.data_ptr(), 8, s) == L.SED_ERR_BAD_ARG
    table = torch.arange(52, dtype=torch.float64, device=dev)
    packed = torch.empty(6, 1 + 2 * 8, dtype=torch.int32, device=dev)
    assert L.lib.sed_threshold_merge_runs(p.data_ptr(), 3, 50, 2, 0.5, 0, table.data_ptr(), 52, 0.5, packed.data_ptr(), 8, s) \
        == L.SED_ERR_CAPACITY                                                    # 50 frames can hold 26 runs > cap 8
    torch.cuda.synchronize()
    for code in (0, L.SED_ERR_BAD_ARG, L.SED_ERR_UNSUPPORTED_NFFT, L.SED_ERR_CAPACITY, L.SED_ERR_SHORT_INPUT,
                 L.SED_ERR_WRONG_DEVICE, -1000 - 2):
        assert L.status_string(code)
    L.lib.sed_plan_destroy(plan)
    L.lib.sed_plan_destroy(None)                                                 # a no-op, like free(NULL)


def test_one_plan_on_two_streams_and_a_side_stream():
    """Calls are asynchronous on the stream they are given; a plan holds read-only tables, so two streams may use it at
    the same time.  Results equal the default-stream result bit for bit."""
    sed_amd, L = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    xa = torch.randn(64, 320000, generator=g, device=dev) * 0.1
    xb = torch.randn(64, 320000, generator=g, device=dev) * 0.1
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    want_a, want_b = fe(xa), fe(xb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs = []
    for _ in range(4):                                                           # interleaved launches on both streams
        with torch.cuda.stream(s1):
            ya = fe(xa)
        with torch.cuda.stream(s2):
            yb = fe(xb)
        outs.append((ya, yb))
    s1.synchronize()
    s2.synchronize()
    for ya, yb in outs:
        assert torch.equal(ya, want_a) and torch.equal(yb, want_b)
    from sed_amd import decode as D
    det = torch.rand(256, 20, 1, generator=g, device=dev)
    table = torch.tensor([k * 0.48 for k in range(22)], dtype=torch.float64, device=dev)
    want = D.threshold_merge_runs(det, 0.5, table, 0.6, "median3")
    with torch.cuda.stream(s1):
        got = D.threshold_merge_runs(det, 0.5, table, 0.6, "median3")
    s1.synchronize()
    g_, w_ = got.cpu().numpy().reshape(256, -1), want.cpu().numpy().reshape(256, -1)
    assert np.array_equal(g_[:, 0], w_[:, 0]) and g_[:, 0].max() > 0
    for a, b in zip(g_, w_):                                                     # slots past the count are not written
        assert np.array_equal(a[1:1 + 2 * a[0]], b[1:1 + 2 * b[0]])


def test_step_is_capturable_in_a_hip_graph():
    """The front-end and the fused decode launch make no synchronising or allocating HIP call of their own, so a step
    can be captured once (torch.cuda.graph = hipGraph on ROCm) and replayed; replays give the eager results."""
    sed_amd, L = _lib()
    from sed_amd import decode as D
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(4)
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat())
    x = torch.randn(32, 64000, generator=g, device=dev) * 0.1
    det = torch.rand(32, 20, 1, generator=g, device=dev)
    table = torch.tensor([k * 0.48 for k in range(22)], dtype=torch.float64, device=dev)

    def step():
        return fe(x), D.threshold_merge_runs(det, 0.5, table, 0.6, "median3")

    y0, b0 = step()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        step()                                                                   # warm-up outside the capture
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        yg, bg = step()
    for _ in range(3):
        x.mul_(-1.0)                                                             # new input in the captured buffer
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y0)                                                   # the power spectrum is even in the sign
    assert torch.equal(bg[..., 0], b0[..., 0])
    x2 = torch.randn(32, 64000, generator=g, device=dev) * 0.1
    want = fe(x2)
    x.copy_(x2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, want)
