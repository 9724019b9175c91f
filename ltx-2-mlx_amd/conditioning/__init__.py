from .latent import ConditioningError, VideoConditionByLatentIndex
from .tools import AudioLatentTools, VideoLatentTools
