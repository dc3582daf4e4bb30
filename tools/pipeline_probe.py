#!/usr/bin/env python
"""VERDICT r5 item 6, measured (GPU box): does a two-batch software pipeline pay on the headline workload?

The headline step is a dependency chain: STFT -> target map -> BLSTM stack -> fc_dc (active rows) -> Lloyd + masks -> mask-apply +
iSTFT.  Across CONSECUTIVE batches only the two MFMA-free ends are independent of the batch in flight: batch k - 1's mask-apply + iSTFT
and batch k + 1's STFT (0.028 + 0.017 ms) could run on a side stream under batch k's recurrences, which leave 16 of the 256 CUs free.
Two captured graphs over the same resident buffers, same box, identical work per replay (one batch's worth of every kernel):

  serial     the bench's step: everything on one stream
  piped      side stream: mask_istft(previous masks) + stft_logmag(next waveforms), forked at the start of the step; main stream:
             target map -> network -> clustering; joined at the end

(The Lloyd launch -- 256 co-operating workgroups -- cannot run beside a recurrence -- 240 -- at all, whichever comes first: that end
of the back end stays where it is.)  Prints ms per replay of each and the persistent kernels' status."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from onssen_amd.features import mask_istft, stft_logmag
    from onssen_amd.nn._core import _XcdPolicy, _XcdStatus
    from onssen_amd.separation import dc_masks_from_features
    dev = torch.device("cuda", 0)
    wl = bench.build_workload("dc_l2", 32, dev)
    model, wav, hop, n, nfft = wl["model"], wl["wav"], wl["HOP"], wl["N"], wl["NFFT"]
    side = torch.cuda.Stream()
    out = {}
    with torch.no_grad():
        logmag, ri = stft_logmag(wav, nfft, hop)
        masks_prev = dc_masks_from_features(model, logmag).clone()
        torch.cuda.synchronize()

        def serial():
            lm, r = stft_logmag(wav, nfft, hop)
            m = dc_masks_from_features(model, lm)
            return mask_istft(r, m, hop, n)

        def piped():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                sig = mask_istft(ri, masks_prev, hop, n)          # batch k - 1's tail
                lm2, r2 = stft_logmag(wav, nfft, hop)             # batch k + 1's head
            m = dc_masks_from_features(model, logmag)             # batch k: target map -> network -> clustering
            main.wait_stream(side)
            return sig, lm2, r2, m

        for name, fn in (("serial", serial), ("piped", piped)):
            run, g = bench.capture(fn, True)
            ts = [bench.time_replays(run, 40) for _ in range(3)]
            out[name] = {"ms_per_replay": min(ts), "all": ts}
            torch.cuda.synchronize()
            _XcdStatus.poll(wait=True)
        # the legs that move, alone
        for name, fn in (("mask_istft_alone", lambda: mask_istft(ri, masks_prev, hop, n)), ("stft_alone", lambda: stft_logmag(wav, nfft, hop)),
                         ("network_and_clustering_alone", lambda: dc_masks_from_features(model, logmag))):
            run, g = bench.capture(fn, True)
            out[name] = {"ms_per_replay": bench.time_replays(run, 40)}
        torch.cuda.synchronize()
        _XcdStatus.poll(wait=True)
    out["aborts"] = _XcdPolicy.aborts
    out["gain_ms"] = out["serial"]["ms_per_replay"] - out["piped"]["ms_per_replay"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
