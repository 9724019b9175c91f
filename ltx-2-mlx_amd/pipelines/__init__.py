from .common import (ImageCondition, apply_conditionings, audio_modality_from_state, create_image_conditionings, joint_denoise_loop,
                     load_image_tensor, modality_from_state, post_process_latent, timesteps_from_mask)
from .distilled import DistilledConfig, DistilledPipeline, create_distilled_pipeline
from .one_stage import OneStageCFGConfig, OneStagePipeline, create_one_stage_pipeline

__all__ = ["ImageCondition", "apply_conditionings", "create_image_conditionings", "load_image_tensor", "joint_denoise_loop",
           "audio_modality_from_state", "modality_from_state", "post_process_latent", "timesteps_from_mask", "DistilledConfig", "DistilledPipeline",
           "create_distilled_pipeline", "OneStageCFGConfig", "OneStagePipeline", "create_one_stage_pipeline"]
