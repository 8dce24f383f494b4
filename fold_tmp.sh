set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for f in "" "PYAPES_HIP_BC_TWO_PASS=1"; do
  for w in "c5 --size 128,128,128" "c5 --size 64,64,64" "c5 --size 32,32,32"; do
    echo "$f $w"
    env $f python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline --no-roofline-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/iter', round(d['ms_per_step'],5))"
  done
done
