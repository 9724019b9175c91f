#!/bin/bash
# Timeline (start offset, duration, queue) of every kernel of ONE AudioVideo DiT layer inside a step: which of the audio stream's
# kernels sit on the video stream's critical path.  -> gpurun_out/<tag>/av_layer_timeline.txt
set -u
TAG=${1:-av_trace}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o av -- python "$REPO/tools/bench_av.py" --steps 4 "$@" > "$OUT/stdout.log" 2>&1
tail -1 "$OUT/stdout.log"
cp $(find /tmp/rp_$TAG -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
python "$REPO/tools/av_layer_timeline.py" $(find /tmp/rp_$TAG -name "*kernel_trace.csv" | head -1) > "$OUT/av_layer_timeline.txt"
cat "$OUT/av_layer_timeline.txt"
