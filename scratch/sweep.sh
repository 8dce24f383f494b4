PYAPES_HIP_DEBUG=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "pyapes_hip|ms/iter" | head -4
for a in 1 2 3 4; do for b in 1 2; do
  PYAPES_HIP_BPC_A=$a PYAPES_HIP_BPC_B=$b python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bpc A $a B $b', round(d['ms_per_step'],4), round(d['roofline']['phase_a_ms'],4), round(d['roofline']['phase_b_ms'],4))"
done; done
