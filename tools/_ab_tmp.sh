timeout 900 python -m pytest tests/test_gpu_conv_packed.py tests/test_gpu_conv.py tests/test_gpu_parity_fullsize.py tests/test_gpu_model.py -q -x 2>&1 | tail -6
for i in 1 2; do
EMP_CONV256_WIDE=0 timeout 300 python bench.py --fp32-mode 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=0', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
timeout 300 python bench.py --fp32-mode 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=1', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
