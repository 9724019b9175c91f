#!/bin/bash
# Per-dispatch kernel trace of a few denoise steps -> the in-step duration of every kernel of one DiT layer, in launch order
# (gpurun_out/<tag>/layer_trace.txt).  rocprofv3 --kernel-trace only.
set -u
TAG=${1:-layer_trace}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
# LAYER_TRACE_FP8=1: the opt-in fp8 compute mode (tools/fp8_step.py) instead of the bf16 bench loop
if [ "${LAYER_TRACE_FP8:-0}" = 1 ]; then
  CMD="python $REPO/tools/fp8_step.py --steps 3"
else
  CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-loader --no-vae --no-graph --no-kernel-pass $*"
fi
rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_$TAG -o bench -- $CMD > "$OUT/bench_stdout.log" 2>&1
f=$(find /tmp/rp_$TAG -name "*kernel_trace.csv" | head -1)
python - "$f" > "$OUT/layer_trace.txt" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]
names = [short(r["Kernel_Name"]) for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gap = [0.0] + [(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 for i in range(1, len(rows))]
# a layer = the span between consecutive norm_mod_shared ... find the periodic pattern: use the LAST full step
idx = [i for i, n in enumerate(names) if n.startswith("gemm_v4_kernel<1,")]          # FFN-up: once per layer
per = collections.defaultdict(list); gaps = collections.defaultdict(list)
L = idx[-40:-8]                                                                       # 32 layers inside the last step
for a, b in zip(L[:-1], L[1:]):
    seq = list(range(a + 1, b + 1))
    for k, i in enumerate(seq):
        per[(k, names[i])].append(dur[i]); gaps[(k, names[i])].append(gap[i])
tot = 0.0; tg = 0.0
print(f"{'#':>2} {'kernel':60s} {'avg us':>8s} {'min':>8s} {'max':>8s} {'gap before':>10s}")
for (k, n), v in sorted(per.items()):
    g = gaps[(k, n)]
    print(f"{k:2d} {n:60s} {sum(v)/len(v):8.1f} {min(v):8.1f} {max(v):8.1f} {sum(g)/len(g):10.2f}")
    tot += sum(v) / len(v); tg += sum(g) / len(g)
print(f"layer: kernels {tot:.1f} us + gaps {tg:.1f} us = {tot + tg:.1f} us")
PY
cat "$OUT/layer_trace.txt"
