"""ltx-2-mlx_amd: MI355X (gfx950) native implementation of the LTX-2 denoise + VAE-decode hot path.

Import as ``ltx_2_mlx_amd`` (a hyphen is not importable; the root-level ``ltx_2_mlx_amd`` package
re-points ``__path__`` at this directory).  Public surface mirrors the names of the reference
package ``LTX_2_MLX`` for this path: ``model.transformer`` (Modality, LTXModel, X0Model),
``model.video_vae`` (SimpleVideoDecoder, decode_latent, decode_tiled), ``components``,
``conditioning``, ``pipelines`` (DistilledPipeline, DistilledConfig), ``types``.
"""
__version__ = "0.1.0"
