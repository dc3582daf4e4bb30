"""GPU-box tests of the two-batch software pipeline (round 6): ``separation.DCPipeline`` / ``separate_dc_stream`` on
``onssen_blstm_pipe2_forward_f32`` -- layer 1 of batch n-1 beside layer 0 of batch n in one persistent launch.  The contract is
bit equality with ``separate_dc`` on 16-row recurrence groups without the fused first layer, which is what the same rows get
inside a call of more than 32 rows (a tile column never sees its neighbours; the clustering and the iSTFT are per utterance)."""
import warnings

import numpy as np
import pytest
import torch

from onssen_amd.synthetic import make_state_dict, synth_mixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from onssen_amd.hip import get_lib
    get_lib()
    return torch.device("cuda:0")


def _dc(dev, H, seed=3, F=129, D=20):
    from onssen_amd import nn as onn
    sd = make_state_dict("deep_clustering", F, H, 2, D, 2, seed=seed)
    m = onn.deep_clustering(F, H, 2, D)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to(dev).eval()


def _batches(dev, B, n, count, seed=100):
    return [torch.from_numpy(np.stack([synth_mixture(seed + 97 * k + b, n) for b in range(B)])).to(dev) for k in range(count)]


def _reference(model, wav):
    """What ``separate_dc`` gives these rows on the arithmetic the pipeline promises: up to 16 rows the default call (stacked tiles,
    first layer not fused); above, the rows inside a 2B-row call (> 32 rows: 16-row groups) with the fused first layer switched off."""
    from onssen_amd import options
    from onssen_amd.separation import separate_dc
    B = wav.shape[0]
    old = options.configure(fuse_first_layer="0")
    try:
        if B <= 16:
            return separate_dc(model, wav)
        pad = torch.flip(wav, dims=[0]) * 0.5
        big = torch.cat([wav, pad, pad])[:max(2 * B, 34)]
        return separate_dc(model, big)[:B]
    finally:
        options.configure(**old)


@pytest.mark.parametrize("H,B,n,graph", [(32, 5, 64 * 24, False), (64, 20, 64 * 37, False), (64, 20, 64 * 37, True), (128, 32, 64 * 19, True),
                                         (40, 16, 64 * 21, True), (40, 17, 64 * 21, False)])
def test_pipeline_is_bit_identical_to_separate_dc_on_16_row_groups(dev, H, B, n, graph):
    from onssen_amd.nn import _core
    from onssen_amd.separation import DCPipeline
    m = _dc(dev, H, seed=H + B)
    xs = _batches(dev, B, n, 4, seed=H)
    refs = [_reference(m, x) for x in xs]
    a0 = _core._XcdPolicy.aborts
    pipe = DCPipeline(m, B, n, graph=graph)
    outs = []
    assert pipe.push(xs[0]) is None
    for x in xs[1:]:
        outs.append(pipe.push(x).clone())
    outs.append(pipe.flush().clone())
    _core._XcdStatus.flush()
    assert _core._XcdPolicy.aborts == a0
    for k, (o, r) in enumerate(zip(outs, refs)):
        assert torch.isfinite(o).all()
        assert torch.equal(o, r), (k, float((o - r).abs().max()))
    # a second stream of batches through the same object (after the drain), interleaved parities
    assert pipe.push(xs[2]) is None
    assert torch.equal(pipe.push(xs[0]), refs[2])
    assert torch.equal(pipe.flush(), refs[0])


def test_stream_generator_matches_separate_dc_and_recovers_from_an_abort(dev):
    """``separate_dc_stream``: one result per batch, in order; with the bounded waits at 0 every persistent launch gives up --
    the batches of the aborted steps are separated again by ``separate_dc`` (which itself falls back to the launch-per-step
    recurrence), a RuntimeWarning is raised, nothing is lost or duplicated."""
    from onssen_amd.hip import get_lib
    from onssen_amd.nn import _core
    from onssen_amd.separation import separate_dc, separate_dc_stream
    lib = get_lib()
    m = _dc(dev, 64, seed=11)
    xs = _batches(dev, 6, 64 * 30, 5, seed=5)
    refs = [_reference(m, x) for x in xs]
    got = list(separate_dc_stream(m, xs, graph=True))
    assert len(got) == len(xs) and all(torch.equal(g, r) for g, r in zip(got, refs))
    # a batch of another shape in the middle goes through separate_dc; the stream carries on
    odd = _batches(dev, 3, 64 * 30, 1, seed=77)[0]
    got = list(separate_dc_stream(m, xs[:2] + [odd] + xs[2:4], graph=False))
    assert len(got) == 5 and torch.equal(got[2], separate_dc(m, odd))
    assert all(torch.equal(g, refs[i]) for g, i in zip(got[:2] + got[3:], (0, 1, 2, 3)))
    P = _core._XcdPolicy
    r0 = P.recovered
    old = lib.dll.onssen_xcd_spin_limit(0)
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = list(separate_dc_stream(m, xs, graph=False))
    finally:
        lib.dll.onssen_xcd_spin_limit(old)
    assert P.recovered > r0 and any("separate_dc" in str(x.message) for x in w)
    assert len(got) == len(xs)
    for g, r in zip(got, refs):       # launch-per-step path vs persistent path: a borderline bin may change sides
        np.testing.assert_allclose(g.cpu().numpy(), r.cpu().numpy(), atol=2e-3)
    while not P.persistent_allowed():             # consume the back-off the forced aborts left behind
        separate_dc(m, xs[0])
    separate_dc(m, xs[0])
    _core._XcdStatus.flush()
    assert P.persistent_allowed()


def test_pipeline_refuses_what_it_cannot_run(dev):
    from onssen_amd import nn as onn
    from onssen_amd.separation import DCPipeline
    m3 = onn.deep_clustering(129, 32, 3, 20).to(dev).eval()
    with pytest.raises(RuntimeError, match="num_layers = 2"):
        DCPipeline(m3, 4, 64 * 20)
    m = _dc(dev, 32)
    with pytest.raises(RuntimeError, match="B <= 32"):
        DCPipeline(m, 33, 64 * 20)
    with pytest.raises(RuntimeError, match="eval-mode"):
        DCPipeline(m.train(), 4, 64 * 20)
    m.eval()
    pipe = DCPipeline(m, 4, 64 * 20, graph=False)
    with pytest.raises(ValueError, match="expected a"):
        pipe.push(torch.zeros(4, 64 * 21, device=dev))


def test_pipeline_follows_a_weight_update(dev):
    """The captured graphs are bound to the packed images of their capture; a push after the weights changed (load_state_dict,
    repack) re-captures on fresh images."""
    from onssen_amd.separation import DCPipeline
    m = _dc(dev, 32, seed=1)
    xs = _batches(dev, 4, 64 * 22, 2, seed=9)
    pipe = DCPipeline(m, 4, 64 * 22, graph=True)
    pipe.push(xs[0])
    before = pipe.flush().clone()
    assert torch.equal(before, _reference(m, xs[0]))
    sd2 = make_state_dict("deep_clustering", 129, 32, 2, 20, 2, seed=2)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd2.items()})
    pipe.push(xs[0])
    after = pipe.flush().clone()
    assert torch.equal(after, _reference(m, xs[0])) and not torch.equal(after, before)


def test_headline_shape_pipeline(dev):
    """BASELINE configs[1] (2 x BLSTM-600, 32 chunks of 400 frames): bit equality with the 64-row route, and the step time next
    to the sequential call's (printed; bench.py is where it is measured properly)."""
    from onssen_amd.nn import _core
    from onssen_amd.separation import DCPipeline, separate_dc
    m = _dc(dev, 600, seed=0)
    n = 25536
    xs = _batches(dev, 32, n, 2, seed=1)
    refs = [_reference(m, x) for x in xs]
    pipe = DCPipeline(m, 32, n, graph=True)
    assert pipe.push(xs[0]) is None
    o0 = pipe.push(xs[1]).clone()
    o1 = pipe.flush().clone()
    assert torch.equal(o0, refs[0]) and torch.equal(o1, refs[1])
    # against the default 32-row call (stacked 8-row groups, fused first layer): the same separation up to borderline bins
    d = separate_dc(m, xs[0])
    frac = float(((o0 - d).abs() > 1e-4).float().mean())
    print(f"headline pipeline vs default separate_dc: max |d| {float((o0 - d).abs().max()):.3e}, samples off by > 1e-4: {frac:.2e}")
    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    pipe.push(xs[0], check=False)
    pipe.wav[1].copy_(xs[1])
    t_pipe = timed(pipe.replay, 20)
    t_seq = timed(lambda: separate_dc(m, xs[0]), 10)
    print(f"headline shape: pipelined step {t_pipe:.3f} ms, eager separate_dc {t_seq:.3f} ms")
    _core._XcdStatus.post(pipe.ws)
    _core._XcdStatus.flush()
    assert t_pipe < t_seq


# ---- round 6c: the pipeline over RAGGED batches of whole utterances (onssen_blstm_pipe2_forward_ragged_f32) -----------------------
def _ragged_batches(dev, B, ns, seed=300):
    """One ragged batch per entry of ``ns`` (its padded length): row lengths between half of it and all of it, one row at the full length."""
    rng = np.random.default_rng(seed)
    out = []
    for k, n in enumerate(ns):
        lengths = rng.integers(max(n // 2, 192), n + 1, B)
        lengths[rng.integers(0, B)] = n
        wav = np.zeros((B, n), np.float32)
        for b in range(B):
            wav[b, :lengths[b]] = synth_mixture(seed + 131 * k + b, int(lengths[b]))
        out.append((torch.from_numpy(wav).to(dev), torch.from_numpy(lengths.astype(np.int32))))
    return out


@pytest.mark.parametrize("H,B,ns", [(32, 5, (64 * 24, 64 * 31 + 17, 64 * 12, 64 * 31 + 17)), (64, 16, (64 * 40, 64 * 22 + 5, 64 * 33, 64 * 9)),
                                    (600, 16, (64 * 30, 64 * 45 + 63, 64 * 38, 64 * 45))])
def test_ragged_pipeline_is_bit_identical_to_separate_dc_with_lengths(dev, H, B, ns):
    """Every batch of the stream -- each padded to its own longest row -- against ``separate_dc(model, wav, lengths=)`` (whose rows are
    bit for bit their own batch-1 runs, tests/test_gpu_ragged.py): equal bits, zeros behind every row's own length."""
    from onssen_amd.nn import _core
    from onssen_amd.separation import DCRaggedPipeline, separate_dc
    m = _dc(dev, H, seed=H + B)
    xs = _ragged_batches(dev, B, ns, seed=H)
    refs = [separate_dc(m, w, lengths=l) for w, l in xs]
    a0 = _core._XcdPolicy.aborts
    pipe = DCRaggedPipeline(m, B, max(ns) + 100)
    outs = []
    assert pipe.push(*xs[0]) is None
    for w, l in xs[1:]:
        outs.append(pipe.push(w, l).clone())
    outs.append(pipe.flush().clone())
    _core._XcdStatus.flush()
    assert _core._XcdPolicy.aborts == a0
    for k, (o, r) in enumerate(zip(outs, refs)):
        assert o.shape == r.shape and torch.isfinite(o).all()
        assert torch.equal(o, r), (k, float((o - r).abs().max()))
        for b in range(B):
            assert not o[b, :, int(xs[k][1][b]):].any()
    # a second stream through the same object, other parities
    assert pipe.push(*xs[3]) is None
    assert torch.equal(pipe.push(*xs[1]), refs[3])
    assert torch.equal(pipe.flush(), refs[1])
    # one utterance of the stream alone (the reference's batch-1 loop): the same bits inside its length
    w, l = xs[1]
    b = 2
    one = separate_dc(m, w[b:b + 1, :int(l[b])])
    assert torch.equal(outs[1][b, :, :int(l[b])], one[0])


def test_ragged_stream_generator_grows_falls_back_and_recovers(dev):
    """``separate_dc_ragged_stream``: one result per batch, in order; a longer batch than the buffers drains the pipeline and starts a
    larger one; a batch of another B or of more than 16 rows goes through ``separate_dc``; with the bounded waits at 0 every persistent
    launch gives up and every batch is re-run."""
    from onssen_amd.hip import get_lib
    from onssen_amd.nn import _core
    from onssen_amd.separation import separate_dc, separate_dc_ragged_stream
    m = _dc(dev, 48, seed=9)
    xs = _ragged_batches(dev, 6, (64 * 20, 64 * 18, 64 * 40, 64 * 41, 64 * 12), seed=11)
    odd = _ragged_batches(dev, 3, (64 * 15,), seed=12)[0]
    big = _ragged_batches(dev, 18, (64 * 10,), seed=13)[0]
    items = xs[:2] + [odd] + xs[2:4] + [big] + xs[4:]
    refs = [separate_dc(m, w, lengths=l) for w, l in items]
    got = list(separate_dc_ragged_stream(m, items))
    assert len(got) == len(refs)
    for k, (o, r) in enumerate(zip(got, refs)):
        assert torch.equal(o, r), k
    lib = get_lib()
    P = _core._XcdPolicy
    r0 = P.recovered
    old = lib.dll.onssen_xcd_spin_limit(0)
    try:
        with warnings.catch_warnings(record=True) as wn:
            warnings.simplefilter("always")
            got = list(separate_dc_ragged_stream(m, xs))
    finally:
        lib.dll.onssen_xcd_spin_limit(old)
    assert P.recovered > r0 and any("separate_dc" in str(x.message) for x in wn)
    assert len(got) == len(xs)
    own = [refs[i] for i in (0, 1, 3, 4, 6)]
    for g, r in zip(got, own):        # launch-per-step path vs persistent path: a borderline bin may change sides
        np.testing.assert_allclose(g.cpu().numpy(), r.cpu().numpy(), atol=2e-3)
    while not P.persistent_allowed():             # consume the back-off the forced aborts left behind
        separate_dc(m, xs[0][0], lengths=xs[0][1])
    separate_dc(m, xs[0][0], lengths=xs[0][1])
    _core._XcdStatus.flush()
    assert P.persistent_allowed()


def test_ragged_pipeline_refuses_what_it_cannot_run(dev):
    from onssen_amd.separation import DCRaggedPipeline
    m = _dc(dev, 32)
    with pytest.raises(RuntimeError):
        DCRaggedPipeline(m, 17, 64 * 20)
    pipe = DCRaggedPipeline(m, 4, 64 * 20)
    w = torch.zeros(4, 64 * 21, device=dev)
    with pytest.raises(ValueError):
        pipe.push(w, [64 * 21] * 4)                       # longer than n_cap
    with pytest.raises(ValueError):
        pipe.push(w[:, :64 * 10], [64 * 10 + 1] * 4)      # a length past the padded width
