"""sbgm_danra_amd — MI355X (gfx950) implementation of the SBGM_DANRA score-UNet forward and reverse-SDE sampling
hot path behind the reference's Python surface.  Importing the package does not load the HIP
library; the first kernel call does, and fails loudly if it is missing."""
from . import _native  # noqa: F401
from . import optim  # noqa: F401
from .score_unet import (Decoder, DecoderBlock, Encoder, ImageSelfAttention, ScoreNet, SinusoidalEmbedding,  # noqa: F401
                         diffusion_coeff, diffusion_coeff_fn, loss_fn, marginal_prob_std, marginal_prob_std_fn)
from .score_sampling import (Euler_Maruyama_sampler, edm_sigma_schedule, guided_score_fn, ode_sampler,  # noqa: F401
                             pc_sampler)

__version__ = "0.1.0"
