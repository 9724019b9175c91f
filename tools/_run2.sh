set -x
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "rope_compact or qk_fold or qknorm" 2>&1 | tail -15
python -m pytest tests/test_parity.py -m gpu -x -q -k "qk_fold or dit_full_size_denoise" 2>&1 | tail -15
AB_VALUES=0,1,2 python tools/qk_fold_ab.py qk_fold 3 2>&1 | grep -v "^+" | tail -11
bash tools/layer_trace.sh r04_fold_layer > /dev/null 2>&1; cat gpurun_out/r04_fold_layer/layer_trace.txt
