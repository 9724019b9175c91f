from .tools import VideoLatentTools
