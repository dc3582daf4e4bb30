"""The path separates (round 5): train deep clustering for a short, fixed-seed schedule on the synthetic voice-pair corpus
with the package's own training step (1 000 steps, ~5 s), then evaluate on HELD-OUT synthetic mixtures the way the reference does
(egs/wsj0-2mix/deep_clustering/evaluate.py:31-45, RESULT:1) -- once with the device 2-means, once with
sklearn.cluster.KMeans(2, random_state=0) like upstream.  Asserted: the loss falls, the separated signals are far better
than the mixture, and on TRAINED embeddings (not planted clusters) the device back end lands where sklearn lands: same masks
on the active bins up to the arbitrary cluster numbering, same SI-SDR.  tools/trained_probe.py is the long form (2 000
steps; profiles/r05_trained_probe.txt)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_trained_deep_clustering_separates_and_device_2means_matches_sklearn(monkeypatch):
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import trained_probe as P
    from onssen_amd import nn as onn
    from onssen_amd.nn._core import _XcdStatus
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = onn.deep_clustering(129, 600, 2, 20, dropout=0.3).to(dev)
    before, _ = P.evaluate(model, dev, n_utt=4, log=lambda s: None)
    curve, secs = P.train(model, 1000, dev, log_every=100, log=lambda s: None)
    mean, rows = P.evaluate(model, dev, n_utt=8, log=print)
    _XcdStatus.flush()
    print(f"1000 steps in {secs:.1f} s; loss {curve[0][1]:.1f} -> {curve[-1][1]:.1f}; SI-SDR mixture {mean['sdr_mixture']:.2f}, untrained "
          f"{before['sdr_device']:.2f}, device {mean['sdr_device']:.2f}, sklearn {mean['sdr_sklearn']:.2f}, ideal {mean['sdr_ideal_binary']:.2f} dB; "
          f"agreement mean {mean['agree_device_sklearn']:.4f} min {mean['min_agree_device_sklearn']:.4f}")
    assert curve[-1][1] < 0.6 * curve[0][1], curve                       # it learns (3 420 -> 1 450 in the probe)
    assert mean["sdr_device"] > mean["sdr_mixture"] + 6.0, mean           # it separates (the mixture scores ~0 dB, untrained < 2 dB)
    assert mean["sdr_device"] > before["sdr_device"] + 5.0, (before, mean)
    # the device back end against upstream's sklearn on the same trained embeddings
    # (probe, 16 utterances: >= 99.98 % on every utterance at 250 / 500 / 1 000 / 2 000 steps, max gap 0.004 dB; one outlier --
    #  another optimum than sklearn's best of 10 initialisations, seen once on a half-trained network -- is tolerated, not more)
    same = [r for r in rows if r["agree_device_sklearn"] >= 0.99]
    assert len(same) >= len(rows) - 1, [r["agree_device_sklearn"] for r in rows]
    assert max(abs(r["sdr_device"] - r["sdr_sklearn"]) for r in same) <= 0.05, same
    assert abs(mean["sdr_device"] - mean["sdr_sklearn"]) <= 0.6, mean
    assert all(np.isfinite(r["sdr_device"]) for r in rows)


@pytest.mark.parametrize("which", ["clip_adam", "torch_fused"])
@pytest.mark.parametrize("eval_first", [False, True])
def test_eval_after_training_runs_on_the_trained_weights(eval_first, which):
    """Regression (round 5).  ``build_optimizer`` returns ``torch.optim.Adam(fused=True)`` on a GPU, and a fused step moves the
    parameters WITHOUT bumping ``tensor._version`` -- the key the packed weight images were cached on.  Rounds 3-4 therefore ran
    every training forward on the BLSTM images of step 0, and an eval-mode forward after training used whichever images an earlier
    eval-mode forward had left (``tools/micro/train_then_eval_diag.py``: 1.24 max|d| against the oracle on the live state_dict).
    Now: after a few optimizer steps the eval-mode embedding must be the ATen-CPU oracle's on the LIVE state_dict, whether or
    not an eval-mode forward ran before training, and the training forward must see each update (the loss of a fixed batch
    moves from step to step exactly as a from-scratch forward on the updated weights says)."""
    import trained_probe as P
    from onssen_amd import nn as onn
    from onssen_amd.data import SyntheticVoicePairs
    from onssen_amd.dist import train_step
    from onssen_amd.features import stft_logmag
    from onssen_amd.loss import loss_dc
    from onssen_amd.synthetic import synth_mixture
    from onssen_amd.utils import build_optimizer
    from oracle import torch_cpu as TC
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = onn.deep_clustering(129, 600, 2, 20, dropout=0.0).to(dev)
    wav = torch.from_numpy(synth_mixture(900_001, 64 * 120)[None]).to(dev)
    with torch.no_grad():
        logmag, _ = stft_logmag(wav, 256, 64)
        if eval_first:
            model.eval()
            model([logmag])
    model.train()
    # both optimizers that move parameters WITHOUT bumping their versions: the package's own (what build_optimizer returns on a GPU
    # since round 5: raw-pointer kernel) and torch's fused Adam (what it returned in rounds 3-4: the original bug)
    from onssen_amd.utils import ClipAdam
    if which == "clip_adam":
        opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})
        assert isinstance(opt, ClipAdam)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    data = SyntheticVoicePairs(dict(batch_size=4, frame_length=100, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40), device=dev, voices=8)
    w0 = model.rnn.weight_hh_l0.detach().clone()
    batch = next(data)
    losses = [train_step(model, opt, loss_dc, *batch) for _ in range(12)]          # the SAME batch: the loss must fall
    assert float((model.rnn.weight_hh_l0 - w0).abs().max()) > 1e-3
    assert losses[-1] < 0.9 * losses[0], losses
    # a from-scratch training-mode forward on the updated weights (fresh module, state_dict copied) gives the loss the next step reports
    twin = onn.deep_clustering(129, 600, 2, 20, dropout=0.0).to(dev).train()
    twin.load_state_dict(model.state_dict())
    l_twin = float(torch.mean(loss_dc(twin(batch[0]), batch[1])))
    l_next = train_step(model, opt, loss_dc, *batch)
    assert abs(l_next - l_twin) <= 2e-4 * abs(l_twin), (l_next, l_twin)
    model.eval()
    with torch.no_grad():
        emb, = model([logmag])
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ref = TC.deep_clustering_forward(sd, logmag.cpu().numpy()).numpy()
    assert np.abs(emb.cpu().numpy() - ref).max() < 5e-5


def test_clip_adam_is_clip_grad_norm_plus_torch_adam():
    """utils.ClipAdam.step_clipped(5) against ``clip_grad_norm_(params, 5); torch.optim.Adam.step()`` (onssen/utils/train.py:83-84) on
    the parameter shapes of the as-shipped recipe (DC 3xBLSTM-600: 23.9 M parameters in 28 tensors), five steps with gradients
    of very different scales (clipping active on some steps, not on others); also the plain ``step()`` and a state_dict round trip
    into torch.optim.Adam."""
    from onssen_amd import nn as onn
    from onssen_amd.utils import ClipAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ma = onn.deep_clustering(129, 600, 3, 20).to(dev)
    mb = onn.deep_clustering(129, 600, 3, 20).to(dev)
    mb.load_state_dict(ma.state_dict())
    oa = ClipAdam(ma.parameters(), lr=1e-3)
    ob = torch.optim.Adam(mb.parameters(), lr=1e-3)
    g = torch.Generator(device=dev).manual_seed(1)
    for step, scale in enumerate((1e-3, 3e-3, 1e-5, 2e-2, 1e-3)):          # 23.9 M elements: norms ~ 4.9, 14.7, 0.05, 98, 4.9
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            pa.grad = torch.randn(pa.shape, device=dev, generator=g) * scale
            pb.grad = pa.grad.clone()
        if step == 2:
            oa.step()                                                      # the plain update on the same kernel
            ob.step()
        else:
            na = oa.step_clipped(5.0)
            nb = torch.nn.utils.clip_grad_norm_(mb.parameters(), 5.0)
            ob.step()
            assert abs(float(na) - float(nb)) <= 2e-6 * float(nb)
        for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-8, msg=lambda m: f"step {step} {n}: {m}")
    sd = oa.state_dict()
    oc = torch.optim.Adam(ma.parameters(), lr=1e-3)
    oc.load_state_dict(sd)                                                 # same state layout as torch.optim.Adam
    assert float(oc.state[next(iter(ma.parameters()))]["step"]) == 5.0


def test_weight_guard_catches_updates_that_bypass_the_version_counter():
    """a model that STAYS in eval mode (BatchNorm frozen) is fine-tuned with a ``torch.optim.Adam(fused=True)``
    built by hand -- not ``build_optimizer`` -- which moves the parameters without bumping ``tensor._version``: the cached weight images
    of the inference path are then stale and nothing in their key says so.  The weight guard (a device-side sampled checksum compared on
    every reuse, fetched asynchronously) must (a) report it at the next status poll as ``StalePackedWeights`` with the images already
    dropped, so that the repeated forward runs on the live weights, and (b) let the entry points that own their call
    (``separate_dc``) repeat it by themselves.  ``p.data`` arithmetic is caught the same way; a forward without any update raises nothing."""
    import warnings
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    dev = torch.device("cuda:0")
    from onssen_amd import nn as onn
    from onssen_amd.nn import _core
    from onssen_amd.separation import separate_dc
    from onssen_amd.synthetic import make_state_dict, synth_mixture
    from oracle import torch_cpu as TC
    sd = make_state_dict("deep_clustering", 129, 64, 2, 20, 2, seed=11, gain=1.0)
    m = onn.deep_clustering(129, 64, 2, 20, dropout=0.0)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to(dev).eval()
    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.uniform(-6, 1.5, (4, 30, 129)).astype(np.float32)).to(dev)
    live = lambda: TC.deep_clustering_forward({k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, x.cpu().numpy()).numpy()

    def fwd():
        with torch.no_grad():
            return m([x])[0].cpu().numpy()
    y0 = fwd()
    _core._XcdStatus.flush()
    np.testing.assert_allclose(y0, live(), atol=3e-5, rtol=1e-4)
    fwd(); fwd()
    _core._XcdStatus.flush()                                    # reuse without an update: the guard stays quiet
    seen0 = _core._WeightGuard.stale_seen

    opt = torch.optim.Adam(m.parameters(), lr=3e-2, fused=True)          # by hand: no invalidate hook
    v0 = m.rnn.weight_hh_l0._version
    for _ in range(3):
        opt.zero_grad()
        m([x])[0].square().mul(torch.linspace(0, 1, 20, device=dev)).sum().backward()      # eval mode + autograd: the training path
        opt.step()
    bumped = m.rnn.weight_hh_l0._version != v0
    y_stale = fwd()                                              # (runs on the old images unless this torch bumps versions after all)
    if not bumped:
        with pytest.raises(_core.StalePackedWeights):
            _core._XcdStatus.flush()
        assert _core._WeightGuard.stale_seen >= seen0 + 1          # (the stack's guard and the head's guard may both fire)
        assert np.abs(y_stale - live()).max() > 1e-3             # it really was a forward on stale weights
    y1 = fwd()
    _core._XcdStatus.flush()
    np.testing.assert_allclose(y1, live(), atol=3e-5, rtol=1e-4)

    # p.data arithmetic on a head parameter, then an entry point that owns its call: it repeats the call by itself
    wav = torch.from_numpy(np.stack([synth_mixture(300 + b, 64 * 29) for b in range(2)])).to(dev)
    ref0 = separate_dc(m, wav).cpu().numpy()
    with torch.no_grad():
        m.fc_dc.weight.data.mul_(-1.0)                            # flips every embedding: the 2-means partition (and the outputs) survive,
        m.fc_dc.bias.data.mul_(-1.0)                              # so compare the embedding path too
        m.rnn.weight_ih_l1.data.mul_(0.5)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sig = separate_dc(m, wav).cpu().numpy()
    assert any("freshly packed weights" in str(i.message) for i in w)
    m.repack()
    np.testing.assert_array_equal(sig, separate_dc(m, wav).cpu().numpy())          # = what the live weights give
    assert ref0.shape == sig.shape       # (binary masks: the separated signals themselves may coincide with the old ones)
    y2 = fwd()
    _core._XcdStatus.flush()
    np.testing.assert_allclose(y2, live(), atol=3e-5, rtol=1e-4)
