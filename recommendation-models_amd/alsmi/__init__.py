"""alsmi — MI355X-native ALS / IMC / DirtyIMC estimators.

The estimator classes keep the scikit-learn style surface of
GrierPhillips/Recommendation-Models (``src/__init__.py``); all per-entry and
per-row arithmetic runs in hand-written HIP kernels (libalsmi.so, gfx950) bound
through ctypes.  Importing the package does not need a GPU; using an estimator
does, and there is no CPU fallback.
"""
from .als import ALS
from .dimc import DirtyIMC
from .imc import IMC
from .utils import DataHolder

__all__ = ['ALS', 'IMC', 'DirtyIMC', 'DataHolder']
