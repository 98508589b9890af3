"""GPU side of the multi-GPU path: the record packer against its oracle, the exchange over RCCL on a one-rank group,
and the per-GPU share of the 4096-stream server workload (512 streams of 10 s) end to end: front-end, detector
decode, records, gather.  Event records are bit exact; features follow the front-end tolerance."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import decode as od
from oracle import frontend as of

pytestmark = pytest.mark.gpu


def _dev():
    import sed_amd  # noqa: F401  (fails loudly if the extension is missing)
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def rccl_group():
    """A one-rank process group on backend nccl (= RCCL on ROCm): the collective path of the exchange runs for real."""
    import torch.distributed as dist
    dev = _dev()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dist.group.WORLD
    dist.destroy_process_group()


def _oracle_runs(probs, thr, smoothing):
    S, T, C = probs.shape
    cap_in = T // 2 + 1
    sm = od.median3_time(probs, 1) if smoothing == "median3" else probs
    act = (sm > np.float32(thr)).astype(int)
    runs = np.zeros((S, C, cap_in, 2), np.int32)
    counts = np.zeros((S, C), np.int32)
    for s in range(S):
        for c in range(C):
            reg = od.contiguous_regions(act[s, :, c])
            counts[s, c] = len(reg)
            for k, (a, b) in enumerate(reg):
                runs[s, c, k] = (a, b)
    return runs, counts


def test_pack_events_against_oracle_both_input_layouts():
    import sed_amd
    from sed_amd import decode as D
    dev = _dev()
    rng = np.random.default_rng(11)
    for (S, T, C, cap) in ((37, 20, 1, 64), (9, 41, 6, 5), (130, 96, 5, 16), (1, 3, 2, 1), (5, 20, 3, 64)):
        probs = rng.random((S, T, C), dtype=np.float32)
        probs[0] = 0.0                                                    # a silent stream: no records
        if S > 2:
            probs[2] = 1.0                                                # always active: one run per class
        runs, counts = _oracle_runs(probs, 0.5, None)
        want = od.pack_events(runs, counts, cap, first_stream=1000)
        # (a) the buffers of threshold_runs
        g_runs, g_counts = D.threshold_runs(torch.from_numpy(probs).to(dev), 0.5, None)
        assert np.array_equal(g_counts.cpu().numpy(), counts)
        got = sed_amd.sharding.pack_events(g_runs, g_counts, cap, first_stream=1000)
        assert got.shape == (S, cap + 1, 4) and got.dtype == torch.int32
        assert np.array_equal(got.cpu().numpy(), want), (S, T, C, cap)
        # (b) the packed buffer of the fused threshold + merge launch (gap 0 on an index table merges nothing that is
        # not adjacent: use a gap below one frame)
        table = torch.arange(T + 2, dtype=torch.float64, device=dev)
        buf = D.threshold_merge_runs(torch.from_numpy(probs).to(dev), 0.5, table, 0.5, None)
        got2 = sed_amd.sharding.pack_events(buf, None, cap, first_stream=1000)
        assert np.array_equal(got2.cpu().numpy(), want), (S, T, C, cap)
        assert sed_amd.sharding.unpack_events(got2)[0] == []
    with pytest.raises(RuntimeError):
        sed_amd.sharding.pack_events(torch.zeros((2, 1, 3, 2), dtype=torch.int32), torch.zeros((2, 1), dtype=torch.int32), 4)
    with pytest.raises(ValueError):
        sed_amd.sharding.pack_events(torch.zeros((2, 1, 4), dtype=torch.int32, device=dev), None, 4)


def test_records_in_one_launch_equal_the_two_step_path():
    """sed_threshold_merge_records against sed_threshold_merge_runs + sed_pack_events (bit exact) and against the oracle's
    median / threshold / regions / merge / pack chain, several classes, caps that overflow, traces at the LDS limit."""
    import sed_amd
    from sed_amd import decode as D
    dev = _dev()
    rng = np.random.default_rng(23)
    for (S, T, C, cap, gap, smoothing) in ((37, 20, 1, 16, 0.6, "median3"), (9, 41, 6, 5, 1.0, None), (130, 96, 5, 16, 0.3, "median3"),
                                           (1, 3, 2, 1, 0.6, None), (5, 510, 3, 64, 0.5, "gauss"), (3, 1, 1, 4, 0.6, "median3"),
                                           (4, 862, 5, 64, 0.3, "gauss"), (3, 4096, 6, 300, 0.05, "median3"),
                                           (2, 8190, 2, 4200, 0.0, None), (3, 513, 2, 300, 0.5, None)):
        probs = rng.random((S, T, C), dtype=np.float32)
        probs[0] = 0.0
        if S > 2:
            probs[2] = 1.0
        hop = 0.48 if T < 100 else 0.02
        table_host = [k * hop for k in range(T + 2)]
        table = torch.tensor(table_host, dtype=torch.float64, device=dev)
        pt = torch.from_numpy(probs).to(dev)
        got = D.threshold_merge_records(pt, 0.5, table, gap, smoothing, cap, first_stream=700)
        two = sed_amd.sharding.pack_events(D.threshold_merge_runs(pt, 0.5, table, gap, smoothing), None, cap, first_stream=700)
        assert got.shape == (S, cap + 1, 4) and torch.equal(got, two), (S, T, C, cap)
        if smoothing != "gauss":
            sm = od.median3_time(probs, 1) if smoothing == "median3" else probs
            act = (sm > np.float32(0.5)).astype(int)
            cap_in = T // 2 + 1
            runs = np.zeros((S, C, cap_in, 2), np.int32)
            counts = np.zeros((S, C), np.int32)
            for s_ in range(S):
                for c in range(C):
                    ev = [{"file": "f", "event_onset": table_host[a], "event_offset": table_host[b], "event_label": "e", "k": (a, b)}
                          for a, b in od.contiguous_regions(act[s_, :, c])]
                    merged = od.merge_close_events(ev, gap)
                    counts[s_, c] = len(merged)
                    for k, m in enumerate(merged):
                        runs[s_, c, k] = (table_host.index(m["event_onset"]), table_host.index(m["event_offset"]))
            assert np.array_equal(got.cpu().numpy(), od.pack_events(runs, counts, cap, first_stream=700)), (S, T, C, cap)
    with pytest.raises(ValueError):                                    # 8192 frames: one raw run too many for the LDS buffer
        D.threshold_merge_records(torch.zeros((1, 8192, 1), device=dev), 0.5, torch.zeros(8194, dtype=torch.float64, device=dev), 0.5)
    with pytest.raises(RuntimeError):
        D.threshold_merge_records(torch.zeros((1, 4, 1)), 0.5, torch.zeros(6, dtype=torch.float64), 0.5)


def test_decode_and_exchange_crnn_traces_against_the_oracle_chain():
    """decode_and_exchange on CRNN-shaped probabilities (500 files x 862 frames x 5 classes, src/models/crnn/crnn.py:181-182):
    gaussian smoothing + threshold (crnn.py:418-425), region decode with round(k / fps, 3) times (crnn/utils.py:91-134) and
    the gap merge (src/utils/merge_events.py:4-60) at the three gaps of the reference's result files, against the oracle's
    chain of the same steps, as (stream, class, onset frame, offset frame) records."""
    import sed_amd
    dev = _dev()
    rng = np.random.default_rng(418)
    S, T, C, fps = 500, 862, 5, 44100 / 1024
    # smoothed noise through a sigmoid: a few events per class and file, like the reference's result files
    z = rng.standard_normal((S, T + 40, C)).astype(np.float32)
    z = np.stack([np.convolve(z[s, :, c], np.ones(41, np.float32) / 6.0, mode="valid") for s in range(S) for c in range(C)])
    probs = np.ascontiguousarray((1.0 / (1.0 + np.exp(-1.5 * z))).astype(np.float32).reshape(S, C, T).transpose(0, 2, 1))
    probs[3] = 1.0
    probs[4] = 0.0
    labels = {0: "car_horn", 1: "cough", 2: "dog_bark", 3: "siren", 4: "gun_shot"}
    files = [f"f{i}.wav" for i in range(S)]
    table_host = [round(np.int64(k) * (1.0 / fps), 3) for k in range(T + 2)]
    frame_of = {t: k for k, t in enumerate(table_host)}
    assert len(frame_of) == len(table_host)
    table = torch.tensor(table_host, dtype=torch.float64, device=dev)
    raw = od.decode_predictions(od.smooth_and_threshold(probs), files, labels, fps)
    class_of = {v: k for k, v in labels.items()}
    pt = torch.from_numpy(probs).to(dev)
    n_events = 0
    for gap in (0.3, 0.6, 1.0):
        cap = 64
        got = sed_amd.sharding.decode_and_exchange(pt, table, gap, cap=cap, first_stream=0, threshold=0.5, smoothing="gauss")
        got = got.cpu().numpy()
        assert got.shape == (S, cap + 1, 4)
        for i, f in enumerate(files):
            merged = od.merge_close_events(raw[f], gap)
            # records are class-major and onset-ascending; the reference's merged list is onset-sorted across classes
            want = sorted((class_of[e["event_label"]], frame_of[e["event_onset"]], frame_of[e["event_offset"]]) for e in merged)
            n = got[i, 0, 0]
            assert got[i, 0, 1] == 0 and n == len(want), (gap, f)
            assert [tuple(r[1:]) for r in got[i, 1:1 + n]] == want and np.all(got[i, 1:1 + n, 0] == i), (gap, f)
            n_events += n
    assert n_events > 3 * S


def test_decode_and_exchange_beyond_the_one_launch_limit():
    """Traces longer than the LDS buffers of the one-launch kernel go through threshold_runs + merge_runs + pack_events inside
    decode_and_exchange; at the limit both paths exist and give the same records."""
    import sed_amd
    from sed_amd import decode as D
    dev = _dev()
    rng = np.random.default_rng(8)
    for T in (8190, 8191, 20000):
        probs = torch.from_numpy(rng.random((3, T, 2), dtype=np.float32)).to(dev)
        probs[1] = 0.0
        table = torch.arange(T + 2, dtype=torch.float64, device=dev) * 0.01
        got = sed_amd.sharding.decode_and_exchange(probs, table, 0.015, cap=T // 2 + 1, first_stream=5, smoothing="median3")
        runs, counts = D.merge_runs(*D.threshold_runs(probs, 0.5, "median3"), table, 0.015)
        want = sed_amd.sharding.pack_events(runs, counts, T // 2 + 1, 5)
        assert got.shape == (3, T // 2 + 2, 4) and torch.equal(got, want), T
        assert int(got[0, 0, 0]) > 100 and int(got[1, 0, 0]) == 0


def test_serving_step_under_graph_capture():
    """One rank's serving step (fused log-mel kernel of the detector features + one decode kernel with the event records)
    captured in a HIP graph after one eager warm-up and replayed on new audio and new detector outputs: the entry points
    are stream operations only (no allocation or synchronisation at call time beyond torch's graph-private pool), so the
    replays equal the eager results bit for bit."""
    import sed_amd
    from sed_amd import decode as D
    dev = _dev()
    S, n = 24, 48000
    fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.yamnet(), layout="btm")
    table = torch.tensor([k * 0.48 for k in range(43)], dtype=torch.float64, device=dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    pcm = 0.1 * torch.randn((S, n), generator=gen, device=dev)
    det = torch.rand((S, 41, 1), generator=gen, device=dev)
    fe(pcm)
    D.threshold_merge_records(det, 0.5, table, 0.6, "median3", 16, 0)             # warm-up: kernel attributes, plan upload
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_mel = fe(pcm)
        g_rec = D.threshold_merge_records(det, 0.5, table, 0.6, "median3", 16, 0)
    for seed in (11, 12, 13):
        gen = torch.Generator(device=dev).manual_seed(seed)
        pcm.copy_(0.1 * torch.randn((S, n), generator=gen, device=dev))
        det.copy_(torch.rand((S, 41, 1), generator=gen, device=dev))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(g_mel, fe(pcm))
        assert torch.equal(g_rec, D.threshold_merge_records(det, 0.5, table, 0.6, "median3", 16, 0))


def test_exchange_over_rccl_one_rank_group(rccl_group):
    """The function bench.py times (decode -> records -> all-gather), with the collective forced on a one-rank RCCL group."""
    import sed_amd
    dev = _dev()
    rng = np.random.default_rng(5)
    S, T = 256, 20
    probs = rng.random((S, T, 1), dtype=np.float32)
    table_host = [k * 0.48 for k in range(T + 2)]
    table = torch.tensor(table_host, dtype=torch.float64, device=dev)
    out, work = sed_amd.sharding.decode_and_exchange(torch.from_numpy(probs).to(dev), table, 0.6, cap=16, first_stream=512,
                                                     group=rccl_group, async_op=True, force_collective=True)
    work.wait()
    torch.cuda.synchronize()
    assert out.shape == (S, 17, 4)
    events = sed_amd.sharding.unpack_events(out)
    for s in range(S):
        # frame-index form of the same events: the oracle's median + threshold + regions, merged with the 0.6 s gap
        runs, counts = _oracle_runs(probs[s:s + 1], 0.5, "median3")
        ev = [{"file": "f", "event_onset": table_host[a], "event_offset": table_host[b], "event_label": "e", "a": a, "b": b}
              for a, b in runs[0, 0, :counts[0, 0]].tolist()]
        merged = od.merge_close_events(ev, 0.6)
        want = [(512 + s, 0, table_host.index(m["event_onset"]), table_host.index(m["event_offset"])) for m in merged]
        assert events[s] == want[:16], s


def _check_near(hip, gold, tol, near_span=60.0):
    peak = gold.max(axis=-1, keepdims=True)
    near = gold >= peak - near_span
    assert float(np.abs(hip - gold)[near].max()) <= tol


@pytest.mark.parametrize("flavour", ["yamnet", "htsat"])
def test_server_shard_512_streams_end_to_end(rccl_group, flavour):
    """One GPU's share of the two-stage server workload: 512 concurrent 10 s streams through the front-end (YAMNet
    features at 16 kHz, HTS-AT log-mel at 32 kHz), the detector decode of 512 probability traces, records, gather."""
    import sed_amd
    dev = _dev()
    S = 512
    sr = 16000 if flavour == "yamnet" else 32000
    L = 10 * sr
    g = torch.Generator(device=dev).manual_seed(7)
    x = (0.1 * torch.randn((S, L), generator=g, device=dev)).clamp_(-1, 1)
    x[3, 20000:60000] = 0.0
    if flavour == "yamnet":
        fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.yamnet(), layout="btm")
        gold_fn = lambda v: of.yamnet_logmel(v, mode="gold")[0]
        near, tol = 6.9, 1e-3                      # nepers: 60 dB = 6.9 Np
    else:
        fe = sed_amd.LogMelFrontend(sed_amd.FrontendConfig.htsat(), layout="btm")
        gold_fn = lambda v: of.htsat_logmel(v, mode="gold")[:, 0]
        near, tol = 60.0, 1e-3
    y = fe(x)
    T = y.shape[1]
    assert y.shape == (S, T, 64)
    # features: the oracle on a few streams, batch invariance (bit exact) on all of them
    pick = [0, 3, 255, 511]
    gold = np.asarray(gold_fn(x[pick].cpu().numpy()), np.float64).reshape(len(pick), T, 64)
    _check_near(y[pick].cpu().numpy().astype(np.float64), gold, tol, near)
    for lo in (0, 128, 384):
        assert torch.equal(fe(x[lo:lo + 128]), y[lo:lo + 128])
    assert torch.equal(fe(x[511:512]), y[511:512])
    # detector decode of the same 512 streams (20 patches of 0.48 s per 10 s stream), records, gather
    probs = torch.rand((S, 20, 1), generator=g, device=dev)
    table_host = [k * 0.48 for k in range(22)]
    table = torch.tensor(table_host, dtype=torch.float64, device=dev)
    out = sed_amd.sharding.decode_and_exchange(probs, table, 0.6, cap=16, first_stream=3 * S, group=rccl_group,
                                               force_collective=True)
    torch.cuda.synchronize()
    events = sed_amd.sharding.unpack_events(out)
    pn = probs.cpu().numpy()
    for s in list(range(0, S, 37)) + [S - 1]:
        runs, counts = _oracle_runs(pn[s:s + 1], 0.5, "median3")
        ev = [{"file": "f", "event_onset": table_host[a], "event_offset": table_host[b], "event_label": "e"}
              for a, b in runs[0, 0, :counts[0, 0]].tolist()]
        want = [(3 * S + s, 0, table_host.index(m["event_onset"]), table_host.index(m["event_offset"]))
                for m in od.merge_close_events(ev, 0.6)]
        assert events[s] == want, s
    # the same records from four quarter batches: stream-wise independence of the whole tail
    for q in range(4):
        part = sed_amd.sharding.decode_and_exchange(probs[128 * q:128 * (q + 1)], table, 0.6, cap=16,
                                                    first_stream=3 * S + 128 * q)
        assert torch.equal(part, out[128 * q:128 * (q + 1)])
