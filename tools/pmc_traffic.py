"""roofline.traffic of the dominant GEMM from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench.sh
(counter_collection CSVs) -> profiles/<tag>_pmc_traffic.json, the file bench.py reads.
usage: python tools/pmc_traffic.py <commit> out.json pmcB.csv pmcC.csv"""
import csv, hashlib, json, os, sys
commit, out, fb, fc = sys.argv[1:5]


def kernel_source_sha():
    """sha256 over the sources of the measured kernel (bench.py recomputes it: `traffic_stale` when the kernel changed after the counters were collected)"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ltx-2-mlx_amd", "csrc")
    h = hashlib.sha256()
    for f in ("gemm_v4.hip", "gemm_v4_loop.inc", "gemm_epilogue.h", "gemm.h", "common.h"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]
KEY = "gemm_v4_kernel<4, 3, 224, false, "       # both instantiations: plain (attn2.to_out, ff.net.2) and VAR 30 (attn1.to_out with the folded pre-norm's shadow, round 6)
def mean(path, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if KEY in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)
f, nf = mean(fb, "FETCH_SIZE")
w, nw = mean(fc, "WRITE_SIZE")
fetch, write = f * 2 * 1024, w * 1024        # KB -> bytes; gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md)
M, D = 3456, 4096
alg = ((M * D + D * D) * 2 * 2 + (M * 4 * D + D * 4 * D) * 2) / 3 + 2 * M * D * 4 + M * D * 2 / 3      # operands (2 x K=4096, 1 x K=16384) + fp32 x read and write + the bf16 shadow one launch in three writes
json.dump({"kernel": "gemm_v4_kernel<EPI_RESID_GATE_F32, 3, 224>", "commit": commit, "kernel_source_sha16": kernel_source_sha(),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; tools/pmc_bench.sh) over bench.py --steps 2 --warmup 1 "
                     "--no-extra --no-cpu-baseline --no-graph; mean over every launch of the kernel IN the model (2 x K=4096 to_out + 1 x K=16384 ff.net.2 per layer); "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); KB * 1024",
           "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write), "launches_sampled": [nf, nw],
           "per_launch_avg_bytes": round(fetch + write), "algorithmic_bytes_per_launch": round(alg),
           "note": "fabric-side (L2-miss) traffic incl. Infinity-Cache hits: each of the 8 XCD L2s re-fetches the activation / weight panels its tiles need "
                   "(structural with private L2s)"}, open(out, "w"), indent=1)
print(open(out).read())
