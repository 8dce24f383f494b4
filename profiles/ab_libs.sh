for W in c3 c2 c5; do echo "== $W"; for i in 1 2; do for v in $VARIANTS; do
  PYAPES_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libs/lib_$v.so python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['phase_a_ms'],4), round(d['roofline']['phase_b_ms'],4))"
done; done; done
