from .diffusion_steps import EulerDiffusionStep, to_denoised, to_velocity
from .guiders import CFGGuider, CFGStarRescalingGuider, projection_coef
from .noisers import GaussianNoiser
from .patchifiers import AudioPatchifier, VideoLatentPatchifier, get_pixel_coords
from .schedulers import (DISTILLED_SIGMA_VALUES, STAGE_2_DISTILLED_SIGMA_VALUES, LTX2Scheduler,
                         get_sigma_schedule)

__all__ = ["CFGGuider", "CFGStarRescalingGuider", "projection_coef", "EulerDiffusionStep", "to_denoised", "to_velocity", "GaussianNoiser", "AudioPatchifier", "VideoLatentPatchifier",
           "get_pixel_coords", "DISTILLED_SIGMA_VALUES", "STAGE_2_DISTILLED_SIGMA_VALUES", "LTX2Scheduler",
           "get_sigma_schedule"]
