"""`Powerful_sincos` (reference model/ppgn_indnoise_sincos.py:102-262); implementation in ppgn_indnoise.py."""
from .ppgn_indnoise import Powerful_sincos, SinusoidalPosEmb  # noqa: F401
