"""MI355X-native counterparts of the onssen.nn models on the STFT-domain hot
path.  As upstream (onssen/nn/__init__.py:1-5), every model takes a list of
tensors named ``input`` and returns a list of tensors."""
from .chimera import chimera
from .deep_clustering import deep_clustering
from .enhancement import enhance
from .phase_network import phase_net

__all__ = ["chimera", "deep_clustering", "enhance", "phase_net"]
