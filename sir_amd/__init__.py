"""Importable alias for the ``speech-intent-recognizer_amd/`` package directory.

The package directory carries the reference's repository name and therefore a
hyphen, which Python cannot import.  This alias points ``sir_amd.__path__`` at
that directory so that ``import sir_amd.models.models`` resolves to
``speech-intent-recognizer_amd/models/models.py``.  No code lives here.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "speech-intent-recognizer_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
