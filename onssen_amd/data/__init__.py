"""``onssen.data`` counterpart: ``wsj0_2mix_dataloader`` with the reference's call signature (onssen/data/wsj0_2mix.py:26-37).
With a ``data_path`` that holds ``wav8k/min/<partition>/mix/*.wav`` it reads the files (``wsj0_2mix.Wsj02mixFiles``); without
one (there is no corpus in the build image) it synthesises utterances (``synthetic_wsj0_2mix.SyntheticWsj02mix``).  Either way
the features and labels are computed on the GPU and the yield contract is the reference's."""
import glob
import os

from .synthetic_wsj0_2mix import SyntheticWsj02mix
from .wsj0_2mix import Wsj02mixFiles, read_wav, write_wav


def wsj0_2mix_dataloader(model_name, feature_options, partition, device=None):
    fo = feature_options
    path = fo.get("data_path") if isinstance(fo, dict) else getattr(fo, "data_path", None)
    if path and glob.glob(os.path.join(path, "wav8k", "min", partition, "mix", "*.wav")):
        return Wsj02mixFiles(model_name, feature_options, partition, device)
    return SyntheticWsj02mix(model_name, feature_options, partition, device)


__all__ = ["wsj0_2mix_dataloader", "SyntheticWsj02mix", "Wsj02mixFiles", "read_wav", "write_wav"]
