"""`import pycolmap` for code written against the reference package: every name of the match + verify path
(/root/reference/pycolmap/main.cc:91-118 registers them on the `pycolmap` module) resolves to pycolmap_amd's
MI355X implementation.  Only what SURVEY.md section 8 puts in scope exists; anything else raises AttributeError
naming this package, so that a script reaching for extraction / SfM / MVS fails at the attribute, not later."""
import pycolmap_amd as _impl
from pycolmap_amd import *  # noqa: F401,F403
from pycolmap_amd import __version__  # noqa: F401

_PUBLIC = [n for n in dir(_impl) if not n.startswith("_")]
globals().update({n: getattr(_impl, n) for n in _PUBLIC})


def __getattr__(name):
    raise AttributeError(f"pycolmap.{name} is outside pycolmap_amd's scope (exhaustive / sequential matching + two-view "
                         f"verification behind the pycolmap API); available: {', '.join(sorted(_PUBLIC))}")
