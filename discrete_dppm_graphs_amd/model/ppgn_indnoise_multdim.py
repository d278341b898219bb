"""`Powerful_Ind_Mult` (reference model/ppgn_indnoise_multdim.py:88-244): Powerful_Ind whose noise MLP ends in a second GELU
(:112-117).  See ppgn_indnoise.py for how the variants map onto the HIP kernels."""
from .ppgn_indnoise import Powerful_Ind


class Powerful_Ind_Mult(Powerful_Ind):
    time_mlp_final_gelu = True

    def __init__(self, *args, **kwargs):
        kwargs.setdefault("name", "ppgn_indnoise_mult")
        super().__init__(*args, **kwargs)
