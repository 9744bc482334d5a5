# 2 ranks sharing the one GPU of a gpurun box over gloo (FI_BENCH_SHARE_GPU=1), traced by rocprofv3:
# kernel trace + memory-copy trace per process -> profiles/r02_dp2_overlap.txt (scripts/overlap_report.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/dp2
FI_BENCH_SHARE_GPU=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/dp2 -- \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/r02_dp2_bench.json 2> gpurun_out/r02_dp2_bench.err
find /tmp/dp2 -name '*.csv' | head -20
python scripts/overlap_report.py /tmp/dp2 > gpurun_out/r02_dp2_overlap.txt 2>&1
head -60 gpurun_out/r02_dp2_overlap.txt; cut -c1-300 gpurun_out/r02_dp2_bench.json; tail -3 gpurun_out/r02_dp2_bench.err
f=$(find /tmp/dp2 -name '*memory_copy_trace.csv' | head -1); head -4 $f > gpurun_out/r02_memcpy_head.txt; cut -d, -f1-3 $f | sort | uniq -c | sort -rn | head >> gpurun_out/r02_memcpy_head.txt
