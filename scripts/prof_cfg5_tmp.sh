cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CFG=${1:-cfg5}
rm -rf /tmp/p5; ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/p5 -o kt -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1 )
f=$(find /tmp/p5 -name 'kt_kernel_trace.csv' | head -1)
python scripts/trace_groups.py $f conv_bf16_fwd 6 | head -40
