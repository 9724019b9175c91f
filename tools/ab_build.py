#!/usr/bin/env python3
"""Build a VARIANT of libltx2hip.so for same-box A/B timing: `python tools/ab_build.py NAME file.hip='-DX=1 -fno-slp-vectorize' ...`
recompiles the named translation units with extra flags, links them with the standard objects (make first) and writes
ltx-2-mlx_amd/lib/ab/NAME.so (git-ignored; it travels to the GPU box).  Use with LTX2HIP_LIB=ltx-2-mlx_amd/lib/ab/NAME.so.
`--f16` as the first argument: the float16 build (-DLTX2_F16, objects of build/f16) -> lib/ab/NAME.so for LTX2HIP_LIB_F16."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ltx-2-mlx_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-inline-asm".split()


def main():
    argv = sys.argv[1:]
    f16 = bool(argv) and argv[0] == "--f16"
    if f16:
        argv = argv[1:]
    name, specs = argv[0], dict(a.split("=", 1) for a in argv[1:])
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
    bdir = os.path.join(CSRC, "build", "ab", name)
    os.makedirs(bdir, exist_ok=True)
    srcs = [l.split(":=")[1].split() for l in open(os.path.join(CSRC, "Makefile")) if l.startswith("SRCS")][0]
    objs, procs = [], []
    for s in srcs:
        if s in specs:
            o = os.path.join(bdir, s.replace(".hip", ".o"))
            std = ["-fno-slp-vectorize"] if s == "attention.hip" else []       # (the Makefile's per-file flag)
            procs.append(subprocess.Popen(["hipcc", *FLAGS, *(["-DLTX2_F16"] if f16 else []), *std, *specs[s].split(), "-c", os.path.join(CSRC, s), "-o", o], cwd=CSRC))
        else:
            o = os.path.join(CSRC, "build", *(["f16"] if f16 else []), s.replace(".hip", ".o"))
        objs.append(o)
    if any(p.wait() for p in procs):
        raise SystemExit("compile failed")
    out = os.path.join(ROOT, "ltx-2-mlx_amd", "lib", "ab", name + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    print(out)


if __name__ == "__main__":
    main()
