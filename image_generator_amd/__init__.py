"""MI355X-native (gfx950) implementation of the text-conditioned denoiser hot path.

Importing the package loads libmi355_denoiser.so; there is no fallback implementation.
"""
from ._lib import lib, LibraryError  # noqa: F401
