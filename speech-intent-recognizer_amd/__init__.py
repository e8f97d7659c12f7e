"""MI355X-native audio -> log-mel -> CNN+BiGRU intent path.

Import as ``sir_amd`` (alias package at the repository root).  All per-batch
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI declared
in ``include/sir_hip.h``; this Python layer mirrors the reference's surface
(``models/models.py``, ``scripts/{precompute_features,dataset,train,evaluate}.py``)
and only moves pointers.  There is no CPU fallback: ops raise if the HIP
library or a GPU is missing.
"""
__version__ = "0.1.0"
