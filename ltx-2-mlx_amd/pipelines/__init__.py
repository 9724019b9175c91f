from .common import audio_modality_from_state, modality_from_state, post_process_latent, timesteps_from_mask
from .distilled import DistilledConfig, DistilledPipeline, create_distilled_pipeline

__all__ = ["audio_modality_from_state", "modality_from_state", "post_process_latent", "timesteps_from_mask", "DistilledConfig", "DistilledPipeline",
           "create_distilled_pipeline"]
