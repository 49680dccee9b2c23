"""radiative3d_amd -- MI355X-native engine for the Phonon::Propagate hot path
of Radiative3D.  Host model builder in C++ (host/), HIP
kernels + C-ABI in csrc/, thin ctypes mirror here."""
from .model import Engine, Model, Node, Result, run_model  # noqa: F401

__all__ = ["Model", "Engine", "Node", "Result", "run_model"]
