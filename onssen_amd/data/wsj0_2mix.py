"""wsj0-2mix loader over real files with the reference's yield contract (SURVEY rows H1 / N4 tail).

``wsj0_2mix_dataloader(model_name, feature_options, partition, device)`` of the reference (onssen/data/wsj0_2mix.py:26-37)
globs ``<data_path>/wav8k/min/<partition>/mix/*.wav`` (``:78-79``), reads mix / s1 / s2 with librosa / torchaudio and computes
three host STFTs, a random ``frame_length`` crop and the label features per sample (``:103-158``) on the training thread
(``num_workers`` 0, ``:28-32``).  Here (round 5) a producer thread keeps ``loader_prefetch`` batches in flight: the library's
batch reader (``onssen_wav_read_batch_f32``, csrc/wav_io.inc: RIFF PCM 8/16/24/32-bit and IEEE float -> float32, channels
averaged -- what ``librosa.load(fn, sr=None)`` returns for them, and bit for bit what ``read_wav`` below returns) fills one
pinned buffer with the batch's 3 x batch_size signals on ``loader_workers`` host threads; the training thread then issues ONE
host-to-device transfer, ONE ragged STFT launch for all of them (``onssen_stft_logmag_ragged_f32``), two gathers for the crops
and ONE label-kernel launch (``onssen_labels_f32``).  Same
``feature_options`` keys, same list layouts per ``model_name`` as the synthetic loader next door (and the reference):

    "dc"        [feature_mix] , [one_hot, mag_mix]
    "chimera"   [feature_mix] , [one_hot, mag_mix, mag_s1, mag_s2]
    "chimera++" [feature_mix] , [one_hot, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2]
    "phase"     [feature_mix, phase_mix] , [one_hot, mag_mix, mag_s1, mag_s2, phase_s1, phase_s2]

Partitions "tr" / "cv": batches of ``batch_size`` in shuffled order (the reference's ``DataLoader(shuffle=True)``), an
utterance of at most ``frame_length`` frames is repeated ``frame_length // T + 1`` times before the crop (``:118-123``).
Partition "tt": whole utterances, batch 1, ``[feature_mix (1,T,F)] , [stft_r (1,T,F), stft_i (1,T,F), sig_ref (1,2,n)]`` with the
signals zero-padded by ``32 - n % 32`` samples (``get_sigs``, ``:216-228``; upstream's STFT-model branch calls a ``get_ref_sig``
that does not exist, ``:237`` -- the padded pair is what its time-domain branch builds).

Not librosa: a file whose rate differs from ``sampling_rate`` is resampled with ``scipy.signal.resample_poly`` (librosa's
``resample`` uses a different kernel: results differ in the last digits; wsj0-2mix ``wav8k`` files never take this branch).
"""
import glob
import os

import numpy as np
import torch

from ..features import stft_logmag, training_labels


def read_wav(fn):
    """(float32 mono signal, sample rate) of a RIFF file -- ``librosa.load(fn, sr=None)`` for the formats wsj0-2mix ships in."""
    from scipy.io import wavfile
    rate, data = wavfile.read(fn)
    if data.dtype == np.uint8:                       # 8-bit PCM is unsigned
        sig = (data.astype(np.float32) - 128.0) / 128.0
    elif data.dtype == np.int16:
        sig = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:                     # 24-bit comes back left-justified in 32
        sig = data.astype(np.float32) / 2147483648.0
    elif data.dtype in (np.float32, np.float64):
        sig = data.astype(np.float32)
    else:
        raise ValueError(f"{fn}: unsupported sample type {data.dtype}")
    if sig.ndim == 2:
        sig = sig.mean(axis=1, dtype=np.float32)
    return np.ascontiguousarray(sig, dtype=np.float32), int(rate)


def write_wav(fn, sig, rate, subtype="PCM_16"):
    """float signal in [-1, 1) -> RIFF file (the separated signals of an evaluation run; ``subtype`` PCM_16 or FLOAT)."""
    from scipy.io import wavfile
    sig = np.asarray(sig, dtype=np.float32)
    if subtype == "PCM_16":
        wavfile.write(fn, rate, np.clip(np.rint(sig * 32768.0), -32768, 32767).astype(np.int16))
    elif subtype == "FLOAT":
        wavfile.write(fn, rate, sig)
    else:
        raise ValueError(f"unknown subtype {subtype!r}")


def _load(fn, sampling_rate):
    sig, rate = read_wav(fn)
    if rate != sampling_rate:                        # feature_utils.get_stft:17-20 resamples instead of failing
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(rate), int(sampling_rate))
        sig = resample_poly(sig, sampling_rate // g, rate // g).astype(np.float32)
    return sig


class _Slot:
    """One (pinned) host buffer of the loader's ring: ``rows`` signals of ``stride`` float32 samples, contiguous (so that the
    batch goes to the device in ONE plain transfer), + the reader's per-file outputs.  ``release()`` (consumer, after it
    queued its H2D copy) records an event; ``wait()`` (producer, before refilling) waits for it."""

    def __init__(self, pin, numel=48 * 8000 * 8):
        self.pin = pin
        self.event = None
        self.buf = torch.empty(numel, dtype=torch.float32, pin_memory=pin)
        self.frames = self.rates = self.status = torch.zeros(0, dtype=torch.int32)

    def shape(self, rows, stride):
        """The (rows, stride) view of the buffer (reallocated, 25 % larger, when it does not fit)."""
        if rows * stride > self.buf.numel():
            self.buf = torch.empty(int(rows * stride * 1.25), dtype=torch.float32, pin_memory=self.pin)
        if rows > self.frames.numel():
            self.frames, self.rates, self.status = (torch.zeros(rows, dtype=torch.int32) for _ in range(3))
        self.wav = self.buf[:rows * stride].view(rows, stride)
        return self.wav

    def release(self):
        if self.pin:
            self.event = torch.cuda.Event()
            self.event.record()

    def wait(self):
        if self.event is not None:
            self.event.synchronize()
            self.event = None


class Wsj02mixFiles:
    """Iterable over ``(input_list, label_list)`` batches of one partition; ``len()`` = number of batches."""

    def __init__(self, model_name, feature_options, partition="tr", device="cuda:0", shuffle=None, seed=None):
        fo = feature_options
        g = (lambda k: fo[k]) if isinstance(fo, dict) else (lambda k: getattr(fo, k))
        if model_name not in ("dc", "chimera", "chimera++", "phase"):
            raise ValueError(f"unknown model_name {model_name!r}")
        self.model_name = model_name
        self.batch_size, self.frame_length = int(g("batch_size")), int(g("frame_length"))
        self.sampling_rate, self.window_size, self.hop_size = int(g("sampling_rate")), int(g("window_size")), int(g("hop_size"))
        self.db_threshold = float(g("db_threshold"))
        self.device = torch.device(device if device is not None else "cuda:0")
        self.partition = partition
        self.file_list = sorted(glob.glob(os.path.join(g("data_path"), "wav8k", "min", partition, "mix", "*.wav")))
        self.shuffle = (partition != "tt") if shuffle is None else shuffle
        self.rng = np.random.default_rng(seed)      # seed=None: fresh entropy, like the reference's unseeded np.random
        self._headers = {}                          # path -> (frames, rate): see _frames_of

    def __len__(self):
        n = len(self.file_list)
        return n if self.partition == "tt" else -(-n // self.batch_size)

    def _sources(self, fn):
        sep = os.sep + "mix" + os.sep
        return fn, fn.replace(sep, os.sep + "s1" + os.sep), fn.replace(sep, os.sep + "s2" + os.sep)

    def _iter_eval(self):
        for fn in self.file_list:
            mix, s1, s2 = (_load(f, self.sampling_rate) for f in self._sources(fn))
            gap = 32 - len(mix) % 32
            pad = lambda a: np.pad(a, (0, gap))
            # the STFT of the file AS IT IS (get_stft(fn), wsj0_2mix.py:231-233); only the reference signals are padded to a
            # multiple of 32 samples (get_sigs, :216-228) -- the estimate is then istft(..., length = padded length)
            wav = torch.from_numpy(np.ascontiguousarray(mix)[None]).to(self.device)
            logmag, ri = stft_logmag(wav, self.window_size, self.hop_size)
            sig_ref = torch.from_numpy(np.stack([pad(s1), pad(s2)])[None]).to(self.device)
            yield [logmag], [ri[..., 0].contiguous(), ri[..., 1].contiguous(), sig_ref]

    # ---- training partitions: a producer thread reads whole batches into pinned buffers, the consumer makes ONE transfer,
    #      ONE ragged STFT launch and two gathers per batch (round 5; the reference does three librosa.load + three host STFTs
    #      per SAMPLE on the training thread: wsj0_2mix.py:103-158 with num_workers 0, :28-32)
    def _frames_of(self, path):
        """(frames, rate) from the file's header, remembered across epochs (the rows of a batch are sized before it is read)."""
        hit = self._headers.get(path)
        if hit is None:
            from ..hip import get_lib
            fr, rate, ch, bits = get_lib().wav_info(path)
            # a header is input: a streamed file states 0xFFFFFFFF data bytes, a truncated one more frames than it holds -- the row (and the
            # pinned buffer behind it) must be sized from what the file can actually contain
            frame_bytes = max(1, ch) * max(1, abs(bits) // 8)
            fr = min(int(fr), os.path.getsize(path) // frame_bytes)
            if fr <= 0 or rate <= 0:
                raise OSError(f"{path}: no audio frames (header: {fr} frames at {rate} Hz)")
            hit = self._headers[path] = (fr, rate)
        return hit

    def _read_batch(self, files, slot, rng=None):
        """Host side of one batch: the 3 x len(files) signals (mix, s1, s2 per utterance) into the rows of ``slot`` by the
        library's batch reader (csrc/wav_io.inc, ``loader_workers`` host threads, no Python per sample), per-utterance sample
        counts, and the crops the seeded generator draws -- in file order, as the per-utterance loop of rounds 1-4 drew them.
        Returns (slot, n_utt (B,) int32, starts (B,) int64)."""
        from .. import options
        from ..hip import get_lib
        lib = get_lib()
        paths = [p for fn in files for p in self._sources(fn)]
        heads = [self._frames_of(p) for p in paths]
        sr = self.sampling_rate
        # a file at another rate is resampled on the host (feature_utils.get_stft:17-20 resamples instead of failing): its row
        # must hold whichever of the two lengths is longer
        need = max(max(fr, -(-fr * sr // rate)) for fr, rate in heads)
        stride = -(-need // 64) * 64
        wav = slot.shape(len(paths), stride)
        rc = lib.wav_read_batch(paths, wav.data_ptr(), stride, slot.frames.data_ptr(), slot.rates.data_ptr(),
                                slot.status.data_ptr(), max(1, int(options.get("loader_workers"))))
        status = slot.status[:len(paths)].numpy()
        if rc != 0 or status.any():                               # (truncation cannot happen: the rows were sized from the headers)
            bad = int(np.argmax(status != 0))
            raise OSError(f"{paths[bad]}: {lib.dll.onssen_error_string(int(status[bad])).decode()} (status {int(status[bad])})")
        frames = slot.frames[:len(paths)].numpy().copy()
        for r in np.nonzero(slot.rates[:len(paths)].numpy() != sr)[0]:
            from math import gcd
            from scipy.signal import resample_poly
            rate = int(slot.rates[r])
            g = gcd(rate, sr)
            sig = resample_poly(wav[r, :frames[r]].numpy(), sr // g, rate // g).astype(np.float32)[:stride]
            wav[r, :len(sig)] = torch.from_numpy(sig)
            frames[r] = len(sig)
        n_utt = frames.reshape(-1, 3).min(axis=1).astype(np.int32)
        L, hop = self.frame_length, self.hop_size
        starts = np.empty(len(files), np.int64)
        for b, n in enumerate(n_utt):
            T = 1 + int(n) // hop
            times = L // T + 1 if T <= L else 1                   # "pad in a double-copy fashion" (wsj0_2mix.py:118-123)
            starts[b] = int((rng or self.rng).integers(0, T * times - L))
        return slot, n_utt, starts

    def _device_batch(self, item):
        """Device side of one batch: (B, 3, L, F) log-magnitude and (B, 3, L, F, 2) spectrum of (mix, s1, s2), cropped."""
        slot, n_utt, starts = item
        B, L, hop = len(n_utt), self.frame_length, self.hop_size
        wav = slot.wav.to(self.device, non_blocking=True)               # ONE contiguous transfer from the pinned buffer
        slot.release()                                                  # the producer may refill it once that copy has run
        lengths = torch.from_numpy(np.repeat(n_utt, 3))
        logmag, ri = stft_logmag(wav, self.window_size, self.hop_size, lengths=lengths)      # ONE ragged launch for the 3B signals
        T, F = logmag.shape[1], logmag.shape[2]
        Tb = 1 + n_utt.astype(np.int64) // hop
        idx = torch.from_numpy((starts[:, None] + np.arange(L)[None, :]) % Tb[:, None]).to(self.device)   # wraps: the repeated utterance
        rows = torch.arange(B, device=self.device)[:, None, None]
        trip = torch.arange(3, device=self.device)[None, :, None]
        return logmag.view(B, 3, T, F)[rows, trip, idx[:, None, :]], ri.view(B, 3, T, F, 2)[rows, trip, idx[:, None, :]]

    def host_batches(self, order=None, ring=None, rng=None):
        """Generator over the HOST side of the epoch's batches (``_read_batch`` items), synchronously on the calling thread.
        ``rng``: the generator the crops are drawn from (default: the loader's own)."""
        if order is None:
            order = self.rng.permutation(len(self.file_list)) if self.shuffle else np.arange(len(self.file_list))
        ring = ring or [_Slot(self.device.type == "cuda")]
        for k, i0 in enumerate(range(0, len(order), self.batch_size)):
            slot = ring[k % len(ring)]
            slot.wait()
            yield self._read_batch([self.file_list[i] for i in order[i0:i0 + self.batch_size]], slot, rng)

    def _prefetched(self, order):
        """``host_batches`` run ahead of the consumer by ``loader_prefetch`` batches on a producer thread (``loader_workers``
        = 0: on the calling thread, nothing in flight)."""
        import queue
        import threading
        from .. import options
        depth, workers = int(options.get("loader_prefetch")), int(options.get("loader_workers"))
        if workers <= 0 or depth <= 0:
            yield from self.host_batches(order)
            return
        ring = [_Slot(self.device.type == "cuda") for _ in range(depth + 2)]    # queue + the one being filled + the one being copied
        q, stop = queue.Queue(maxsize=depth), threading.Event()

        def put(item):                             # never blocks past `stop`: a consumer that abandons the epoch must not strand the producer
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for item in self.host_batches(order, ring, rng):
                    if not put(item):
                        return
                put(None)
            except BaseException as e:             # reported by the consumer, on the training thread
                put(e)
        # the producer draws its crops from a CHILD generator spawned here, on the calling thread: the parent generator is never touched
        # off-thread (numpy Generators are not thread-safe), and an abandoned epoch cannot perturb the next one's sequence
        rng = self.rng.spawn(1)[0]
        th = threading.Thread(target=produce, name="onssen-wsj0-2mix-loader", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            th.join()                              # (bounded: every put gives up within 0.1 s of `stop`, a batch read is finite)

    def __iter__(self):
        if self.partition == "tt":
            yield from self._iter_eval()
            return
        order = self.rng.permutation(len(self.file_list)) if self.shuffle else np.arange(len(self.file_list))
        for item in self._prefetched(order):
            logmag, ri = self._device_batch(item)                   # (B, 3, L, F), (B, 3, L, F, 2)
            feat = logmag[:, 0].contiguous()
            mix, s1, s2 = (ri[:, i].contiguous() for i in range(3))
            out = training_labels(mix, s1, s2, feat, self.db_threshold, with_cos=self.model_name == "chimera++")
            one_hot, mm, m1, m2 = out[:4]
            one_hot = one_hot.double()                             # np.zeros default dtype upstream (feature_utils.py:86)
            if self.model_name == "dc":
                yield [feat], [one_hot, mm]
            elif self.model_name == "chimera":
                yield [feat], [one_hot, mm, m1, m2]
            elif self.model_name == "chimera++":
                yield [feat], [one_hot, mm, m1, m2, out[4], out[5]]
            else:
                yield [feat, mix], [one_hot, mm, m1, m2, s1, s2]
