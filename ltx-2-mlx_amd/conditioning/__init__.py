from .tools import AudioLatentTools, VideoLatentTools
