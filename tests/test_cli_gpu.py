"""CLI plumbing config of BASELINE.json (distilled pipeline, 256x384x17, 8 steps, random-init
2-layer DiT) through scripts/generate.py: hipGraph replay and the eager per-step API agree."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generate_cli_plumbing(dev, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import generate
    kw = dict(height=256, width=384, num_frames=17, num_inference_steps=8, seed=3, num_layers=2, num_heads=2,
              vae_base_channels=64)
    f1 = generate.generate_video("a test prompt", output_path=str(tmp_path / "a.mp4"), use_hip_graph=True, **kw)
    f2 = generate.generate_video("a test prompt", output_path=str(tmp_path / "b.mp4"), use_hip_graph=False, **kw)
    assert f1.shape == (17, 256, 384, 3) and f1.dtype == torch.uint8
    la = np.load(tmp_path / "a_latent.npz")["latent"]
    lb = np.load(tmp_path / "b_latent.npz")["latent"]
    assert la.shape == (1, 128, 3, 8, 12)
    assert np.abs(la - lb).max() < 1e-4 * max(1.0, np.abs(lb).max())
    assert os.path.exists(tmp_path / "a.npz")
    with pytest.raises(ValueError, match="8\\*k \\+ 1"):
        generate.generate_video("x", num_frames=16, **{k: v for k, v in kw.items() if k != "num_frames"})
    with pytest.raises(NotImplementedError):
        generate.generate_video("x", generate_audio=True, **kw)
    with pytest.raises(ValueError, match="--lora needs --weights"):
        generate.generate_video("x", lora_path="style.safetensors", **kw)
    # --image: the conditioned latent frame survives the loop at strength 1.0 (image-to-video through the VAE encoder)
    from PIL import Image
    Image.fromarray((np.random.RandomState(1).rand(256, 384, 3) * 255).astype(np.uint8)).save(tmp_path / "cond.png")
    generate.generate_video("a test prompt", output_path=str(tmp_path / "c.mp4"), image_path=str(tmp_path / "cond.png"),
                            image_strength=1.0, skip_vae=True, **kw)
    lc = np.load(tmp_path / "c_latent.npz")["latent"]
    assert lc.shape == (1, 128, 3, 8, 12) and np.isfinite(lc).all()
    assert np.abs(lc[:, :, 0] - la[:, :, 0]).mean() > 0.05
    # --text-features: Gemma features -> Embeddings1DConnector on the GPU -> 1024 x 3840 context for the DiT
    np.savez(tmp_path / "feats.npz", features=(0.1 * np.random.RandomState(2).randn(24, 3840)).astype(np.float32),
             attention_mask=np.ones(24, dtype=np.float32))
    generate.generate_video("a test prompt", output_path=str(tmp_path / "d.mp4"), text_features_path=str(tmp_path / "feats.npz"),
                            skip_vae=True, **kw)
    ld = np.load(tmp_path / "d_latent.npz")["latent"]
    assert ld.shape == (1, 128, 3, 8, 12) and np.isfinite(ld).all() and np.abs(ld - la).mean() > 1e-3
    # a.mp4 went through save_video: an .mp4 with ffmpeg on the box, PNG frames without
    assert os.path.exists(tmp_path / "a.mp4") or len(os.listdir(tmp_path / "a_frames")) == 17
