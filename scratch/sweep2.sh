for a in 1 2 3 4; do for b in 1 2 3 4; do
  PYAPES_HIP_BPC_A=$a PYAPES_HIP_BPC_B=$b python bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 bpc A $a B $b', round(d['ms_per_step'],4), round(d['roofline']['phase_a_ms'],4), round(d['roofline']['phase_b_ms'],4))"
done; done
