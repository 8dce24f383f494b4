cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
PYAPES_HIP_DEBUG=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/$1 2>&1 | grep pyapes_hip | head -3
cat /tmp/kt/*/*kernel_stats.csv | head -8 | cut -c1-200
