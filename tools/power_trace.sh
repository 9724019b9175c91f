#!/bin/bash
# Socket power / clocks sampled while bench.py runs (rocm-smi, ~10 Hz): direct evidence of the power-limited clock.
# usage: tools/power_trace.sh <tag> [bench args]   -> gpurun_out/<tag>/{power.csv,bench.json}
tag=${1:-power}; shift
out=gpurun_out/$tag; mkdir -p $out
python bench.py --steps 240 --warmup 4 --no-extra --no-cpu-baseline --no-vae --no-kernel-pass "$@" > $out/bench.log 2>&1 &
bpid=$!
rm -f $out/power_raw.csv
t0=$(date +%s%3N)
while kill -0 $bpid 2>/dev/null; do
    line=$(rocm-smi -d 0 --showpower --showclocks --showtemp --csv 2>/dev/null | grep card0)
    echo "$(( $(date +%s%3N) - t0 )),$line" >> $out/power_raw.csv
    sleep 0.05
done
wait $bpid
tail -1 $out/bench.log > $out/bench.json
rocm-smi -d 0 --showpower --showclocks --showtemp --csv 2>/dev/null | head -2 > $out/header.txt
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|power" | head -5 >> $out/header.txt
