"""voicefixer_amd -- the VoiceFixer restore / vocoder inference path on AMD MI355X (gfx950).

Same public surface as ``voicefixer`` for this path::

    from voicefixer_amd import VoiceFixer, Vocoder
"""
from .api import VoiceFixer, Vocoder  # noqa: F401

__all__ = ["VoiceFixer", "Vocoder"]
