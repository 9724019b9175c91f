"""Build container only: record the keyword surface of the reference's `generate_video` (scripts/generate.py:933-997) --
parameter names and defaults, read with `ast` (the module itself imports mlx) -- as tests/golden/generate_video_signature.json.
tests/test_host_cpu.py checks the product's generate_video against it; nothing on a GPU box reads /root/reference."""
import ast, json, os, sys
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/scripts/generate.py"
tree = ast.parse(open(ref).read())
fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "generate_video")
args = fn.args.args
defaults = [None] * (len(args) - len(fn.args.defaults)) + list(fn.args.defaults)
out = []
for a, d in zip(args, defaults):
    out.append({"name": a.arg, "required": d is None, "default": None if d is None else ast.literal_eval(d)})
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "generate_video_signature.json")
json.dump({"source": "scripts/generate.py:933-997 (Acelogic/LTX-2-MLX)", "params": out}, open(dst, "w"), indent=1)
print(f"wrote {dst}: {len(out)} parameters")
