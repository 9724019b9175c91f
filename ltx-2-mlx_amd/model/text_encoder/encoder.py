"""Text-side per-prompt path behind the reference's API (LTX_2_MLX/model/text_encoder/encoder.py:13-32 output records,
:65-253 VideoGemmaTextEncoderModel, :255-370 AudioVideoGemmaTextEncoderModel, :373-413 create_text_encoder,
:415-560 load_text_encoder_weights).  Gemma itself is outside this build: the entry points take its hidden states
(`encode_from_hidden_states`) or already-projected features (`encode_projected`).  The caption projection
3840 -> 4096 stays in the transformer (`ltx2_dit_prepare`), as in the reference."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .connector import Embeddings1DConnector
from .feature_extractor import GemmaFeaturesExtractorProjLinear, GemmaFeaturesExtractorV2

CONNECTOR_PREFIX = "model.diffusion_model.video_embeddings_connector."
AUDIO_CONNECTOR_PREFIX = "model.diffusion_model.audio_embeddings_connector."
FEATURE_EXTRACTOR_PREFIX = "text_embedding_projection."


@dataclass
class VideoGemmaEncoderOutput:
    video_encoding: torch.Tensor       # [B, T, D]
    attention_mask: torch.Tensor       # [B, T]


@dataclass
class AudioVideoGemmaEncoderOutput:
    video_encoding: torch.Tensor
    audio_encoding: torch.Tensor
    attention_mask: torch.Tensor


def _additive_mask(attention_mask: torch.Tensor) -> torch.Tensor:
    """binary [B, T] (1 = attend) -> additive [B,1,1,T] (encoder.py:103-134)."""
    am = attention_mask.float()
    return ((am - 1) * 3.40e38).reshape(am.shape[0], 1, 1, am.shape[-1])


def _binary_mask(output_mask: torch.Tensor) -> torch.Tensor:
    return (output_mask.squeeze(1).squeeze(1) >= -0.5).to(torch.int32)


class VideoGemmaTextEncoderModel:
    def __init__(self, feature_extractor: Optional[GemmaFeaturesExtractorProjLinear] = None,
                 embeddings_connector: Optional[Embeddings1DConnector] = None):
        self.feature_extractor = feature_extractor or GemmaFeaturesExtractorProjLinear()
        self.embeddings_connector = embeddings_connector or Embeddings1DConnector()

    def encode_projected(self, projected_features: torch.Tensor, attention_mask: torch.Tensor) -> VideoGemmaEncoderOutput:
        encoded, output_mask = self.embeddings_connector(projected_features, _additive_mask(attention_mask).to(projected_features.device))
        binary = _binary_mask(output_mask)
        return VideoGemmaEncoderOutput(video_encoding=encoded * binary[:, :, None], attention_mask=binary)

    def encode_from_hidden_states(self, hidden_states: List[torch.Tensor], attention_mask: torch.Tensor,
                                  padding_side: str = "left") -> VideoGemmaEncoderOutput:
        feats = self.feature_extractor.extract_from_hidden_states(hidden_states=hidden_states, attention_mask=attention_mask,
                                                                  padding_side=padding_side)
        return self.encode_projected(feats, attention_mask)

    __call__ = encode_from_hidden_states


class AudioVideoGemmaTextEncoderModel:
    def __init__(self, feature_extractor=None, embeddings_connector: Optional[Embeddings1DConnector] = None,
                 audio_embeddings_connector: Optional[Embeddings1DConnector] = None):
        self.feature_extractor = feature_extractor or GemmaFeaturesExtractorProjLinear()
        self.embeddings_connector = embeddings_connector or Embeddings1DConnector()
        self.audio_embeddings_connector = audio_embeddings_connector or Embeddings1DConnector()

    def encode_from_hidden_states(self, hidden_states: List[torch.Tensor], attention_mask: torch.Tensor,
                                  padding_side: str = "left") -> AudioVideoGemmaEncoderOutput:
        feats = self.feature_extractor.extract_from_hidden_states(hidden_states=hidden_states, attention_mask=attention_mask,
                                                                  padding_side=padding_side)
        video_in, audio_in = feats if isinstance(self.feature_extractor, GemmaFeaturesExtractorV2) else (feats, feats)
        add = _additive_mask(attention_mask).to(video_in.device)
        video, output_mask = self.embeddings_connector(video_in, add)
        binary = _binary_mask(output_mask)
        audio, _ = self.audio_embeddings_connector(audio_in, add)
        return AudioVideoGemmaEncoderOutput(video_encoding=video * binary[:, :, None], audio_encoding=audio, attention_mask=binary)

    __call__ = encode_from_hidden_states


def create_text_encoder(hidden_dim: int = 3840, num_gemma_layers: int = 49, connector_heads: int = 30, connector_head_dim: int = 128,
                        connector_layers: int = 2, num_registers: int = 128, device="cuda") -> VideoGemmaTextEncoderModel:
    return VideoGemmaTextEncoderModel(
        feature_extractor=GemmaFeaturesExtractorProjLinear(hidden_dim=hidden_dim, num_layers=num_gemma_layers, device=device),
        embeddings_connector=Embeddings1DConnector(attention_head_dim=connector_head_dim, num_attention_heads=connector_heads,
                                                   num_layers=connector_layers, num_learnable_registers=num_registers, device=device))


def _strip(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def load_text_encoder_weights(encoder, weights_path: str) -> int:
    """Feature extractor (`text_embedding_projection.*`) and connector(s) (`model.diffusion_model.video_embeddings_connector.*`,
    `…audio_embeddings_connector.*` for the AV encoder) from the LTX-2 safetensors checkpoint (reference
    encoder.py:415-560).  Caption-projection weights belong to the transformer and are not read here.  Returns the number
    of tensors consumed."""
    from safetensors import safe_open
    want = (FEATURE_EXTRACTOR_PREFIX, CONNECTOR_PREFIX, AUDIO_CONNECTOR_PREFIX)
    sd: Dict[str, torch.Tensor] = {}
    with safe_open(weights_path, framework="pt") as f:
        for k in f.keys():
            if k.startswith(want):
                sd[k] = f.get_tensor(k)
    n = 0
    fe = _strip(sd, FEATURE_EXTRACTOR_PREFIX)
    if fe:
        encoder.feature_extractor.load_state_dict(fe)
        n += len(fe)
    conn = _strip(sd, CONNECTOR_PREFIX)
    if conn:
        encoder.embeddings_connector.load_state_dict(conn)
        n += len(conn)
    aconn = _strip(sd, AUDIO_CONNECTOR_PREFIX)
    if aconn and hasattr(encoder, "audio_embeddings_connector"):
        encoder.audio_embeddings_connector.load_state_dict(aconn)
        n += len(aconn)
    if n == 0:
        raise KeyError(f"no text-encoder tensors under {want} in {weights_path}")
    return n
