set -x
mkdir -p gpurun_out/r04_t1
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or adaln" 2>&1 | tail -5
python -m pytest tests/test_parity.py tests/test_parity_fullsize.py -m gpu -x -q -k "dit_48 or denoise_loop or per_token" 2>&1 | tail -5
bash tools/ab_run.sh 2 "head base" python tools/gemm_layer_time.py 2>&1 | grep -v "^+" | tee gpurun_out/r04_t1/gemm_layer_ab.txt
bash tools/ab_run.sh 2 "head base" python tools/rowops_time.py 2>&1 | grep -v "^+" | tee gpurun_out/r04_t1/rowops_ab.txt
bash tools/ab_run.sh 2 "head base" python bench.py --steps 16 --no-extra --no-cpu-baseline --no-loader --no-vae --no-power --no-graph 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('==='): print(l)
    elif l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['roofline']['avg_launch_us'])
" | tee gpurun_out/r04_t1/bench_ab.txt
