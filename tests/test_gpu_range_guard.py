"""The split16 range guard (csrc/spconv.h split16_track / split16_report, eyoc_model_range_check): the safety net under
the default arithmetic of large batches.  A checkpoint is doctored so that ONE stored activation tensor - the output of a
chosen layer, hence of a chosen kernel's epilogue - reaches 1e5 while everything else stays O(10); the oracle
(``oracle/resunet.py``, which records every tensor a fused implementation stores) says what the largest |activation| is
and where.  Checked per case: the activation probe reports the oracle's maximum, ``eyoc_model_range_check`` answers
EYOC_ERR_RANGE, explicit split16 raises and leaves NaN features, automatic mode re-runs in fp32 and matches the oracle at
the forward's usual 1e-4 bar, the sticky flag is reported once, forwards pipelined behind an overflowing one are judged on
their own, and ``RegistrationPipeline`` falls back for good.  Ref: model/resunet.py:142-193 (the tensors), lib/trainer.py
never checks ranges - fp32 has none to check."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL = 1e-4
BIG = 1.0e5          # where the doctored activation peaks (the guard trips at 6e4, fp16 ends at 65504)


def _lib():
    from eyoc_amd import _lib as L
    return L, L.load()


@pytest.fixture(scope="module")
def cloud():
    """Two 12k-voxel clouds in one batch: >= 8192 rows, so the automatic arithmetic is split16 on Z-ordered rows (the
    production path of the bench: staged stride-1 / transposed kernels, staged first convolution, fused tail)."""
    from eyoc_amd import synthetic as syn
    from oracle import coords as oc
    from oracle import resunet as orr
    p = syn.make_pair(3, beams=32, azimuths=1000, band=None)
    coords = syn.batch_coords([p["coords0"], p["coords1"]])
    feats = np.ones((len(coords), 1), np.float32)
    sd = syn.make_weights()
    maps = oc.build_maps(coords, 5)
    _, inter, _ = orr.resunet_forward(sd, coords, feats, maps=maps, return_intermediate=True)
    base = {k: v.numpy() for k, v in inter["stored"].items()}
    assert len(coords) >= 8192 and max(np.abs(v).max() for v in base.values()) < 100
    return dict(coords=coords, feats=feats, sd=sd, maps=maps, base=base)


def _scale_norm(sd, norm, s, shift=0.0):
    """The batch norm ``norm`` answers ``s * (y - shift)`` where it answered ``y``."""
    sd[f"{norm}.bn.weight"] = sd[f"{norm}.bn.weight"] * np.float32(s)
    sd[f"{norm}.bn.bias"] = (sd[f"{norm}.bn.bias"] - np.float32(shift)) * np.float32(s)


def doctor(sd, base, case):
    """-> (state dict, name of the layer whose stored output peaks at BIG)."""
    sd = copy.deepcopy(sd)
    if case in ("conv1", "conv2", "conv4_tr"):
        # the layer's output X also feeds a residual add: make it s * (X - c) <= 0 everywhere, so the block output
        # relu(... + X') stays small, and divide the consuming convolution by s
        norm, consumer = {"conv1": ("norm1", "block1.conv1"), "conv2": ("norm2", "block2.conv1"),
                          "conv4_tr": ("norm4_tr", "block4_tr.conv1")}[case]
        x = base[case]
        c = float(x.max()) + 1.0
        s = BIG / (c - float(x.min()))
        _scale_norm(sd, norm, s, c)
        sd[f"{consumer}.kernel"] = sd[f"{consumer}.kernel"] / np.float32(s)
        return sd, case
    if case == "block1.conv1":            # staged stride-1 layer; its output feeds block1.conv2 only
        s = BIG / float(base[case].max())
        _scale_norm(sd, "block1.norm1", s)
        sd["block1.conv2.kernel"] = sd["block1.conv2.kernel"] / np.float32(s)
        return sd, case
    if case == "block2_tr.conv2":         # the last stored tensor in front of the fused 1x1 tail (columns 0..63 of its input)
        s = 1.5 * BIG / float(base[case].max())     # relu(s bn2(.) + x) is not s relu(bn2(.) + x): aim high
        _scale_norm(sd, "block2_tr.norm2", s)
        k = sd["conv1_tr.kernel"].copy()
        k[:64] /= np.float32(s)
        sd["conv1_tr.kernel"] = k
        return sd, case
    if case == "conv1_tr":                # the tail's own 64-channel intermediate (registers only)
        s = BIG / float(base[case].max())
        sd["conv1_tr.kernel"] = sd["conv1_tr.kernel"] * np.float32(s)
        sd["final.kernel"] = sd["final.kernel"] / np.float32(s)
        return sd, case
    raise KeyError(case)


def make_model(sd, math="auto"):
    import eyoc_amd
    m = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    m.spconv_math = math
    return m


def forward(model, cloud, feats=None):
    import eyoc_amd
    f = cloud["feats"] if feats is None else feats
    return model(eyoc_amd.SparseTensor(torch.from_numpy(f).cuda(), coordinates=torch.from_numpy(cloud["coords"]).cuda())).F


CASES = ["conv1", "block1.conv1", "conv2", "conv4_tr", "block2_tr.conv2", "conv1_tr"]


@pytest.fixture
def class_major(request):
    """True: the transposed convolutions run on spconv_upc.hip (class-major tiles) although the batch is small."""
    L, lib = _lib()
    prev = L.knob("eyoc_spconv_upc_min_rows", 0 if request.param else 1 << 30)
    yield request.param
    L.knob("eyoc_spconv_upc_min_rows", prev)


@pytest.fixture
def tail_in_epilogue(request):
    """True: 64-channel workgroups although the batch is small, so that block2_tr.conv2 carries the 1x1 tail in its epilogue (round 6,
    eyoc_model_fuse_tail 2 = the default of large batches)."""
    L, lib = _lib()
    prev = L.knob("eyoc_spconv_st_split_below", 0 if request.param else 1024)
    yield request.param
    L.knob("eyoc_spconv_st_split_below", prev)


@pytest.mark.parametrize("case,class_major,tail_in_epilogue",
                         [(c, False, False) for c in CASES] + [("conv4_tr", True, False), ("block2_tr.conv2", False, True), ("conv1_tr", False, True)],
                         indirect=["class_major", "tail_in_epilogue"])
def test_overflow_in_one_layer_is_seen_reported_and_recovered_from(cloud, case, class_major, tail_in_epilogue):
    from oracle import resunet as orr
    L, lib = _lib()
    sd, where = doctor(cloud["sd"], cloud["base"], case)
    want, inter, _ = orr.resunet_forward(sd, cloud["coords"], cloud["feats"], maps=cloud["maps"], return_intermediate=True)
    want = want.numpy()
    peaks = {k: float(v.abs().max()) for k, v in inter["stored"].items()}
    top = max(peaks, key=peaks.get)
    others = max(v for k, v in peaks.items() if k != where)
    assert top == where and peaks[where] > 7e4 and others < 6e3, (top, peaks[where], others)   # the doctoring isolates the layer

    # ---- explicit split16: the check raises EYOC_ERR_RANGE, the probe has the oracle's maximum, the features are NaN
    m = make_model(sd, "split16")
    m.probe_activations(True)
    with pytest.raises(L.EyocError) as ei:
        forward(m, cloud)
    assert ei.value.code == L.ERR_RANGE and "split16" in str(ei.value)
    assert m.last_spconv_math == "split16"
    got_peak = m.last_max_activation
    assert got_peak is not None and abs(got_peak - peaks[where]) <= 1e-5 * peaks[where], (got_peak, peaks[where])
    assert m.check_range() is not None            # reported once: the sticky flag is clear again (and the probe keeps its maximum)
    m.range_check = False                          # a pipelined caller: no check inside the call
    F = forward(m, cloud).cpu().numpy()
    if case == "conv1_tr":
        # the tail's intermediate never reaches memory: the rows it overflowed in answer NaN, every other row is right
        h = inter["stored"]["conv1_tr"].abs().max(1).values.numpy()
        assert np.isnan(F[h >= 6.001e4]).all() and (h >= 6.001e4).sum() > 0
        fine = h < 5.999e4
        assert np.isfinite(F[fine]).all() and np.abs(F[fine] - want[fine]).max() <= REL * np.abs(want).max()
    else:
        assert np.isnan(F).all()
    with pytest.raises(L.EyocError) as ei:
        m.check_range()
    assert ei.value.code == L.ERR_RANGE

    # ---- automatic mode: the same forward is re-run with fp32 MFMAs and meets the forward's bar
    a = make_model(sd, "auto")
    got = forward(a, cloud).cpu().numpy()
    assert a.last_spconv_math == "fp32" and a.spconv_math == "auto"
    err = float(np.abs(got - want).max() / np.abs(want).max())
    cos = float((got * want).sum(1).min())
    print(f"range guard [{case}]: oracle peak {peaks[where]:.6g} probe {got_peak:.6g}; fp32 fallback vs oracle {err:.2e} cos {cos:.8f}")
    assert err < REL and cos > 1 - 1e-6
    a.check_range()                                # nothing pending after the fallback


def test_a_forward_behind_an_overflowing_one_is_judged_on_its_own(cloud):
    """Pipelined use (``range_check = False``, one check after several forwards): the overflow flag the last layer poisons
    on belongs to ONE forward, the flag ``check_range`` reads is sticky.  Round 3's single flag made every forward between
    an overflow and its check NaN - and a caller who caught the error and carried on had those pass a later, clean check."""
    L, lib = _lib()
    base = cloud["base"]["block1.conv1"]
    sd = copy.deepcopy(cloud["sd"])
    s = 2.0e4 / float(base.max())                  # peaks at 2e4 on unit features ...
    _scale_norm(sd, "block1.norm1", s)
    sd["block1.conv2.kernel"] = sd["block1.conv2.kernel"] / np.float32(s)
    m = make_model(sd, "split16")
    clean = forward(m, cloud).cpu().numpy()        # (checked inside the call: no overflow)
    assert np.isfinite(clean).all()
    m.range_check = False
    hot = cloud["feats"] * np.float32(8.0)         # ... and well above 6e4 on features of 8 (the first convolution is linear in them)
    F_hot = forward(m, cloud, hot)
    F_after = forward(m, cloud)
    F_hot, F_after = F_hot.cpu().numpy(), F_after.cpu().numpy()
    assert np.isnan(F_hot).all()
    np.testing.assert_array_equal(F_after, clean)  # bit-identical to the forward that ran alone
    with pytest.raises(L.EyocError) as ei:
        m.check_range()
    assert ei.value.code == L.ERR_RANGE
    m.check_range()
    np.testing.assert_array_equal(forward(m, cloud).cpu().numpy(), clean)
    m.check_range()


def test_only_a_range_error_triggers_the_fp32_fallback(cloud, monkeypatch):
    """``forward`` in automatic mode must re-raise anything that is not EYOC_ERR_RANGE (a HIP error out of the check would
    otherwise hide behind a silent re-run)."""
    L, lib = _lib()
    m = make_model(cloud["sd"], "auto")
    forward(m, cloud)

    def boom():
        raise L.EyocError("eyoc_model_range_check failed (-2): hipMemcpyAsync failed", L.ERR_HIP)
    monkeypatch.setattr(m, "check_range", boom)
    with pytest.raises(L.EyocError) as ei:
        forward(m, cloud)
    assert ei.value.code == L.ERR_HIP and m.last_spconv_math == "split16"


def test_registration_pipeline_switches_to_fp32_for_good(cloud):
    """``RegistrationPipeline.register`` defers the check to its read-back (harness._checked): on an overflow in automatic
    mode it switches the model to fp32 MFMAs and runs the step again - same poses as a pipeline that ran in fp32 all along."""
    from eyoc_amd import synthetic as syn
    from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
    sd, _ = doctor(cloud["sd"], cloud["base"], "block1.conv1")
    pairs = [syn.make_pair(s, beams=32, azimuths=1000, band=None) for s in (3, 4)]
    dev = torch.device("cuda")
    cfg = RegistrationConfig(ransac_max_iteration=100000, n_points=2000)
    desc = dict(inlier_ratio=0.3)
    res = {}
    for math in ("auto", "fp32"):
        m = make_model(sd, math)
        pipe = RegistrationPipeline(m, cfg)
        batch = DeviceBatch(pairs, [3, 4], dev, n_points=cfg.n_points, descriptor=desc)
        res[math] = pipe.register(batch, seed=7)
        if math == "auto":
            assert m.spconv_math == "fp32" and m.last_spconv_math == "fp32"
            again = pipe.register(batch, seed=7)               # stays there, no second fallback
            for r0, r1 in zip(res[math], again):
                np.testing.assert_array_equal(r0.transformation, r1.transformation)
    for r0, r1 in zip(res["auto"], res["fp32"]):
        assert np.isfinite(r0.transformation).all()
        np.testing.assert_array_equal(r0.transformation, r1.transformation)


def test_enqueued_steps_carry_their_own_verdict_and_results(cloud):
    """``RegistrationPipeline.enqueue`` (what a caller that pipelines steps uses: bench.py): the read-back of the result records
    and of the guard's verdict on THAT forward is enqueued with the step.  Three steps in flight - clean, overflowing, clean: each
    ``wait()`` returns its own step's records (equal to ``register``'s) and only the middle one reports an overflow."""
    from eyoc_amd import registration as reg, synthetic as syn
    from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
    L, lib = _lib()
    pairs = [syn.make_pair(s, beams=32, azimuths=1000, band=None) for s in (3, 4)]
    dev = torch.device("cuda")
    cfg = RegistrationConfig(ransac_max_iteration=100000, n_points=2000)
    batch = DeviceBatch(pairs, [3, 4], dev, n_points=cfg.n_points, descriptor=dict(inlier_ratio=0.3))
    clean = make_model(cloud["sd"], "split16")
    bad_sd, _ = doctor(cloud["sd"], cloud["base"], "block1.conv1")
    bad = make_model(bad_sd, "split16")
    p_clean, p_bad = RegistrationPipeline(clean, cfg), RegistrationPipeline(bad, cfg)
    want = p_clean.register(batch, seed=7)
    pend = [p_clean.enqueue(batch, seed=7, slot=0), p_bad.enqueue(batch, seed=7, slot=0), p_clean.enqueue(batch, seed=7, slot=1)]
    out = [p.wait() for p in pend]
    assert [o[1] for o in out] == [False, True, False]
    for host, _ in (out[0], out[2]):
        got = [reg.decode_ransac_result(host[i], batch.n_points) for i in range(batch.P)]
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g.transformation, w.transformation)
            assert g.inliers == w.inliers and g.survivors == w.survivors
    with pytest.raises(L.EyocError) as ei:                  # the sticky flag of the model that overflowed is still there to be read
        bad.check_range()
    assert ei.value.code == L.ERR_RANGE
    clean.check_range()


def test_two_steps_in_flight_give_the_serial_records(cloud):
    """``enqueue(tail_stream=True)`` (bench.py's round-5 loop): only the forward runs on the caller's stream; gather / NN / RANSAC /
    read-back run on the pipeline's second stream while the NEXT forward is already enqueued on the first.  Four steps back to
    back, the maps of each built on the side stream: every step's records are those of the serial ``register`` - bit for bit -,
    and the overflowing model in the middle is flagged on its own step only."""
    from eyoc_amd import registration as reg, synthetic as syn
    from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
    L, lib = _lib()
    dev = torch.device("cuda")
    cfg = RegistrationConfig(ransac_max_iteration=100000, n_points=2000)
    seeds = [(3, 4), (5, 6)]
    batches = [DeviceBatch([syn.make_pair(s, beams=32, azimuths=1000, band=None) for s in ss], list(ss), dev, n_points=cfg.n_points,
                           descriptor=dict(inlier_ratio=0.3)) for ss in seeds]
    clean = make_model(cloud["sd"], "split16")
    bad_sd, _ = doctor(cloud["sd"], cloud["base"], "block1.conv1")
    bad = make_model(bad_sd, "split16")
    p_clean, p_bad = RegistrationPipeline(clean, cfg), RegistrationPipeline(bad, cfg)
    want = [p_clean.register(b, seed=7) for b in batches]
    plan = [(p_clean, 0), (p_clean, 1), (p_bad, 0), (p_clean, 1), (p_clean, 0)]
    pend = []
    for k, (pipe, b) in enumerate(plan):
        maps = pipe.prepare_maps(batches[b])
        pend.append(pipe.enqueue(batches[b], seed=7, maps=maps, slot=k & 1, tail_stream=True))
        if k >= 1:                      # two steps in flight: step k - 1 is read while step k runs
            host, over = pend[k - 1].wait()
            pipe_prev, b_prev = plan[k - 1]
            assert over == (pipe_prev is p_bad)
            if not over:
                got = [reg.decode_ransac_result(host[i], batches[b_prev].n_points) for i in range(batches[b_prev].P)]
                for g, w in zip(got, want[b_prev]):
                    np.testing.assert_array_equal(g.transformation, w.transformation)
                    assert g.inliers == w.inliers and g.survivors == w.survivors and g.best_hypothesis == w.best_hypothesis
    host, over = pend[-1].wait()
    assert not over
    for g, w in zip([reg.decode_ransac_result(host[i], batches[0].n_points) for i in range(batches[0].P)], want[0]):
        np.testing.assert_array_equal(g.transformation, w.transformation)
    with pytest.raises(L.EyocError):
        bad.check_range()
    clean.check_range()


def test_fp32_forwards_after_an_overflow_report_a_clean_verdict(cloud):
    """word 0 of the guard (this forward's verdict) was cleared by split16 forwards only - after an overflow had switched
    the model to fp32 MFMAs, every later ``enqueue`` kept reporting ``overflowed`` although fp32 forwards cannot overflow."""
    from eyoc_amd import synthetic as syn
    from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
    L, lib = _lib()
    dev = torch.device("cuda")
    cfg = RegistrationConfig(ransac_max_iteration=50000, n_points=2000)
    batch = DeviceBatch([syn.make_pair(3, beams=32, azimuths=1000, band=None)], [3], dev, n_points=cfg.n_points,
                        descriptor=dict(inlier_ratio=0.3))
    bad_sd, _ = doctor(cloud["sd"], cloud["base"], "block1.conv1")
    m = make_model(bad_sd, "split16")
    pipe = RegistrationPipeline(m, cfg)
    _, over = pipe.enqueue(batch, seed=1, slot=0).wait()
    assert over
    with pytest.raises(L.EyocError):
        m.check_range()                              # the sticky word, reported once
    m.spconv_math = "fp32"                           # what harness._checked does in automatic mode
    for tail in (False, True):
        host, over = pipe.enqueue(batch, seed=1, slot=1, tail_stream=tail).wait()
        assert not over and m.last_spconv_math == "fp32"
        assert np.isfinite(np.frombuffer(host.numpy().tobytes(), np.float32)[:16]).all()
    m.check_range()


def test_progress_event_fires_inside_the_forward(cloud):
    """``model.progress_event(layer)``: an event every forward records in front of launch ``layer`` - a side stream that waits for it
    (bench.py --maps-after layer:-3) runs beside the rest of the forward.  It must lie between the forward's first and last event."""
    m = make_model(cloud["sd"], "split16")
    ev = m.progress_event(-3)
    for _ in range(2):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        forward(m, cloud)
        t1.record()
        torch.cuda.synchronize()
        assert ev.query()
    # a timing-enabled twin of the same event kind: where inside the forward does it fire?
    evt = torch.cuda.Event(enable_timing=True)
    evt.record()
    from eyoc_amd import _lib as L
    import ctypes as C
    L.check(L.load().eyoc_model_set_progress_event(m._handle, -3, C.c_void_p(evt.cuda_event)))
    t0.record()
    forward(m, cloud)
    t1.record()
    torch.cuda.synchronize()
    assert 0.3 * t0.elapsed_time(t1) < t0.elapsed_time(evt) < t0.elapsed_time(t1)
    m.progress_event(None)
    forward(m, cloud)
