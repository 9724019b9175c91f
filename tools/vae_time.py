"""decode_latent 768x512x65 (random-init full-width decoder): 1 warm-up + 3 timed repetitions; prints ms per decode."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, decode_latent
dev = torch.device("cuda:0")
dec = SimpleVideoDecoder(device=dev)
dec.init_random_weights(seed=7)
dec.generator = torch.Generator(device=dev).manual_seed(99)
g = torch.Generator(device=dev).manual_seed(1)
H, W = (int(a) for a in (sys.argv[1:3] if len(sys.argv) >= 3 else (16, 24)))        # latent height / width: 16 24 = 768x512, 32 48 = 1536x1024
z = torch.randn(1, 128, 9, H, W, generator=g, device=dev)
decode_latent(z, dec)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    fr = decode_latent(z, dec)
torch.cuda.synchronize()
print(f"decode_latent {W * 32}x{H * 32}x65: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms, frames {tuple(fr.shape)}")
