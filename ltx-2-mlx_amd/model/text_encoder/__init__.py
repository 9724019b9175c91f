"""Text-side per-prompt path: Gemma feature extractors + Embeddings1DConnector on the MI355X kernels.
Mirrors LTX_2_MLX/model/text_encoder/__init__.py for the classes on this path; Gemma-3 itself is out of scope."""
from .connector import Embeddings1DConnector
from .encoder import (AudioVideoGemmaEncoderOutput, AudioVideoGemmaTextEncoderModel, VideoGemmaEncoderOutput, VideoGemmaTextEncoderModel,
                      create_text_encoder, load_text_encoder_weights)
from .feature_extractor import GemmaFeaturesExtractorProjLinear, GemmaFeaturesExtractorV2, norm_and_concat_padded_batch

__all__ = ["Embeddings1DConnector", "GemmaFeaturesExtractorProjLinear", "GemmaFeaturesExtractorV2", "norm_and_concat_padded_batch",
           "VideoGemmaTextEncoderModel", "AudioVideoGemmaTextEncoderModel", "VideoGemmaEncoderOutput", "AudioVideoGemmaEncoderOutput",
           "create_text_encoder", "load_text_encoder_weights"]
