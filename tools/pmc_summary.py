"""Aggregate rocprofv3 --pmc passes (counter_collection.csv) of one command into a per-kernel table:
launches, avg duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / wall), MFMA-pipe utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 pipes)), fabric-side read/write bytes and GB/s
(FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction, KB * 1024).
usage: python tools/pmc_summary.py out.md passA.csv [passB.csv ...]"""
import collections, csv, sys

out_path, files = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> [values]
dur = collections.defaultdict(list)                                       # kernel -> [ns] (from every pass)
for f in files:
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("gemm", "attn", "norm_mod", "qknorm", "vt_transpose", "pixnorm", "conv", "gemv")):
            continue
        k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (r["Dispatch_Id"], f)
        if key not in seen:
            seen.add(key)
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))

def avg(v):
    return sum(v) / len(v) if v else float("nan")

rows = []
for k, c in agg.items():
    n = max(len(v) for v in c.values())
    t_ns = avg(dur[k])
    gui = avg(c.get("GRBM_GUI_ACTIVE", [])) / 8.0        # rocprofv3 sums the counter over the 8 XCDs
    mfma = avg(c.get("SQ_VALU_MFMA_BUSY_CYCLES", []))
    fetch = avg(c.get("FETCH_SIZE", [])) * 2 * 1024
    write = avg(c.get("WRITE_SIZE", [])) * 1024
    rows.append((t_ns * n, k, n, t_ns / 1e3, gui / t_ns if gui == gui else float("nan"), mfma / (gui * 1024) if gui == gui else float("nan"),
                 fetch / 1e6, write / 1e6, (fetch + write) / t_ns))
rows.sort(reverse=True)
with open(out_path, "w") as f:
    f.write("Per-kernel PMC summary of `bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline [--no-graph]` (tools/pmc_bench.sh: three\n"
            "separate rocprofv3 --pmc passes with --kernel-trace only).  eff. clock = GRBM_GUI_ACTIVE / 8 XCDs / wall (short kernels\n"
            "over-read: wall excludes ramp); MFMA-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 1024 matrix pipes), i.e. the\n"
            "fraction of the peak AT THE CLOCK THE CHIP ACTUALLY RAN; read/write = FETCH_SIZE x2 (gfx950 correction) / WRITE_SIZE, KB x 1024,\n"
            "fabric side (Infinity-Cache hits included).\n\n")
    f.write("| kernel | launches | avg us | eff. clock GHz | MFMA-pipe busy | read MB/launch | write MB/launch | fabric GB/s |\n|---|---|---|---|---|---|---|---|\n")
    for _, k, n, us, ghz, mf, rd, wr, gbs in rows:
        f.write(f"| `{k}` | {n} | {us:.1f} | {ghz:.2f} | {mf:.3f} | {rd:.1f} | {wr:.1f} | {gbs:.0f} |\n")
print(open(out_path).read())
