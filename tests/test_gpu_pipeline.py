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
