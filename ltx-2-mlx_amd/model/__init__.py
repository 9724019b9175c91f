from . import transformer, video_vae

__all__ = ["transformer", "video_vae"]
