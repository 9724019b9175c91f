"""Build container only: record the command-line surface of the reference's scripts/generate.py (argparse block :2364-2641)
-- flag strings, dest, action, type, default, choices, read with `ast` (the module itself imports mlx) -- and the mapping
main() uses from parsed arguments to generate_video keywords (:2658-2725), as tests/golden/generate_cli_flags.json.
tests/test_host_cpu.py parses every flag with the product's parser against it; nothing on a GPU box reads /root/reference."""
import ast, json, os, sys
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/scripts/generate.py"
tree = ast.parse(open(ref).read())
fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
flags = []
for node in ast.walk(fn):
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
        names = [ast.literal_eval(a) for a in node.args]
        kw = {}
        for k in node.keywords:
            if k.arg == "help":
                continue
            kw[k.arg] = k.value.id if (k.arg == "type" and isinstance(k.value, ast.Name)) else ast.literal_eval(k.value)
        flags.append({"flags": names, **kw})
flags.sort(key=lambda f: f["flags"][0])
# generate_video(...) call in main(): keyword -> expression text
call = next(n for n in ast.walk(fn) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "generate_video")
kwmap = {k.arg: ast.unparse(k.value) for k in call.keywords}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "generate_cli_flags.json")
json.dump({"source": "scripts/generate.py:2364-2725 (Acelogic/LTX-2-MLX)", "flags": flags, "generate_video_kwargs": kwmap}, open(dst, "w"), indent=1)
print(f"wrote {dst}: {len(flags)} arguments, {len(kwmap)} generate_video keywords")
