python -m pytest tests/test_gpu_bc_fused.py -m gpu -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-probe > /dev/null 2>&1
cat /tmp/kt/*/*kernel_stats.csv | head -8 | cut -c1-150
