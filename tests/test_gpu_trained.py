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
    assert curve[-1][1] < 0.8 * curve[0][1], curve                       # it learns (3 700 -> ~2 450 in the probe: the norm-form loss flattens early)
    assert mean["sdr_device"] > mean["sdr_mixture"] + 6.0, mean           # it separates (the mixture scores ~0 dB, untrained < 2 dB)
    assert mean["sdr_device"] > before["sdr_device"] + 5.0, (before, mean)
    # the device back end against upstream's sklearn on the same trained embeddings
    # (probe, 16 utterances: 99.98-100 % at 1 000 and 2 000 steps, max gap 0.014 dB; at 500 steps ONE utterance sat in another
    #  optimum than sklearn's best of 10 initialisations -- 66 % --: one such outlier is tolerated here, not more)
    same = [r for r in rows if r["agree_device_sklearn"] >= 0.99]
    assert len(same) >= len(rows) - 1, [r["agree_device_sklearn"] for r in rows]
    assert max(abs(r["sdr_device"] - r["sdr_sklearn"]) for r in same) <= 0.05, same
    assert abs(mean["sdr_device"] - mean["sdr_sklearn"]) <= 0.6, mean
    assert all(np.isfinite(r["sdr_device"]) for r in rows)
