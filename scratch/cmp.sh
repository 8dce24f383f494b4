for i in 1 2; do
for mode in fused unfused; do
  if [ $mode = unfused ]; then export PYAPES_HIP_BC_UNFUSED=1; else unset PYAPES_HIP_BC_UNFUSED; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', round(d['ms_per_step'],4), round(d['roofline']['phase_a_ms'],4), round(d['roofline']['phase_b_ms'],4))"
done; done
