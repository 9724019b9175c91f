"""decode_tiled (reference tiling.py defaults) on a 9 x H x W latent (default 32 x 48 = 1536x1024x65): wall time, tiles, time per tile."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, TilingConfig, decode_tiled, generate_tile_specs
dev = torch.device("cuda:0")
dec = SimpleVideoDecoder(device=dev)
dec.init_random_weights(seed=7)
dec.generator = torch.Generator(device=dev).manual_seed(99)
g = torch.Generator(device=dev).manual_seed(1)
H, W = (int(a) for a in (sys.argv[1:3] if len(sys.argv) >= 3 else (32, 48)))
z = torch.randn(1, 128, 9, H, W, generator=g, device=dev)
cfgt = TilingConfig.default()
specs = list(generate_tile_specs(z.shape, cfgt))
print(len(specs), "tiles; first:", specs[0])
next(decode_tiled(z, dec, cfgt, show_progress=False)); torch.cuda.synchronize()
t0 = time.perf_counter()
v = next(decode_tiled(z, dec, cfgt, show_progress=False)); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"decode_tiled {W*32}x{H*32}x65: {dt*1e3:.1f} ms ({dt*1e3/len(specs):.1f} ms per tile), out {tuple(v.shape)}")
