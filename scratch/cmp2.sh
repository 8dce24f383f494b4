python -m pytest tests/test_gpu_parity_golden.py tests/test_gpu_fastpath.py tests/test_gpu_properties.py -m gpu -q 2>&1 | tail -3
for i in 1 2 3; do
for ov in 1 0; do
  PYAPES_HIP_OVERLAP=$ov python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap $ov', round(d['ms_per_step'],4), round(d['roofline']['phase_a_ms'],4), round(d['roofline']['phase_b_ms'],4))"
done; done
