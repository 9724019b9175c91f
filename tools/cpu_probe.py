import sys, time, os, torch
sys.path.insert(0, os.getcwd())
from oracle import dit, loop
print("cpus", os.cpu_count())
cfg = dit.DiTConfig(num_layers=1)
D = cfg.inner_dim
g = torch.Generator().manual_seed(0)
w = {}
for name, shape in dit.dit_weight_shapes(cfg).items():
    if name.startswith("transformer_blocks.0."):
        w[name] = torch.randn(shape, generator=g) * (0.02 if len(shape) == 2 else 1.0)
N, S = 3456, 1024
x = torch.randn(1, N, D, generator=g); ctx = torch.randn(1, S, D, generator=g) * 0.1; emb = torch.randn(1, 1, 6, D, generator=g) * 0.1
pe = dit.rope_split_tables(loop.video_positions(1, 9, 16, 24, 24.0), D, 32, 10000.0, [20, 2048, 2048])
for th in (32, 64, 128, 256):
    torch.set_num_threads(th)
    with torch.no_grad():
        ts = []
        for r in range(2):
            t0 = time.time(); dit.transformer_block(x, ctx, emb, pe, w, 0, cfg); ts.append(time.time() - t0)
    print("threads", th, "block s", [round(t, 2) for t in ts], flush=True)
