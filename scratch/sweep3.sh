for rj in 4 2 1; do
  export PYAPES_HIP_RJ=$rj
  python bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 RJ $rj', round(d['ms_per_step'],4), round(d['roofline']['phase_a_ms'],4), round(d['roofline']['phase_b_ms'],4))"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 RJ $rj', round(d['ms_per_step'],4), round(d['roofline']['phase_a_ms'],4), round(d['roofline']['phase_b_ms'],4))"
  python bench_ops.py 2>/dev/null | grep -E "euler_march|jacobi 3-D|laplacian" | python scratch/fmt.py
done
