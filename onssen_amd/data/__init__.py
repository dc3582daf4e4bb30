"""``onssen.data`` counterpart: ``wsj0_2mix_dataloader`` with the reference's call signature (onssen/data/wsj0_2mix.py:26-37).
With a ``data_path`` that holds ``wav8k/min/<partition>/mix/*.wav`` it reads the files (``wsj0_2mix.Wsj02mixFiles``).  The
synthetic corpus (``synthetic_wsj0_2mix.SyntheticWsj02mix``; there is no WSJ0 in the build image) is used only when it is ASKED
for: ``data_path`` absent / ``None`` / ``""`` / ``"synthetic"``, or ``ONSSEN_SYNTHETIC_DATA=1`` in the environment (recipe configs
carry the author's corpus path).  A ``data_path`` that is given but holds no files for the partition raises ``FileNotFoundError``
like the reference's empty dataset would fail -- numbers must never come from synthetic mixtures by accident.  Either way the
features and labels are computed on the GPU and the yield contract is the reference's.

``feature_utils`` carries the reference's helper names (``get_stft``, ``get_log_magnitude``, ``get_phase``, ``get_cos_difference``,
``get_one_hot``: onssen/data/feature_utils.py:5-21,49-95), NumPy in / NumPy out over the same device kernels."""
import glob
import os

from .. import options
from . import feature_utils
from .feature_utils import get_cos_difference, get_log_magnitude, get_one_hot, get_phase, get_stft
from .synthetic_wsj0_2mix import SyntheticVoicePairs, SyntheticWsj02mix
from .wsj0_2mix import Wsj02mixFiles, read_wav, write_wav


def wsj0_2mix_dataloader(model_name, feature_options, partition, device=None):
    fo = feature_options
    path = fo.get("data_path") if isinstance(fo, dict) else getattr(fo, "data_path", None)
    if not path or path == "synthetic":
        return SyntheticWsj02mix(model_name, feature_options, partition, device)
    pattern = os.path.join(path, "wav8k", "min", partition, "mix", "*.wav")
    if glob.glob(pattern):
        return Wsj02mixFiles(model_name, feature_options, partition, device)
    if options.get("synthetic_data") not in ("", "0"):
        return SyntheticWsj02mix(model_name, feature_options, partition, device)
    raise FileNotFoundError(f"wsj0_2mix_dataloader: no files match {pattern!r} (data_path is set but holds nothing for partition "
                            f"{partition!r}); pass data_path='' / 'synthetic' or set ONSSEN_SYNTHETIC_DATA=1 for the synthetic corpus")


__all__ = ["wsj0_2mix_dataloader", "feature_utils", "get_stft", "get_log_magnitude", "get_phase", "get_cos_difference", "get_one_hot",
           "SyntheticVoicePairs", "SyntheticWsj02mix", "Wsj02mixFiles", "read_wav", "write_wav"]
