# Regenerates the round's evidence under gpurun_out/ (copy what is to be judged into profiles/).
#   bash scripts/refresh_profiles.sh r02
set -x
R=${1:-r02}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
( cd /tmp && timeout 600 python $GRAFT_REPO_ROOT/bench.py > $O/${R}_bench_n1.json 2> $O/${R}_bench_n1.err )
# rocprofv3 per-kernel summary of the same command (headline flags, fewer steps; no nested PMC passes)
# (FI_WGRAD_SIDE_PIXELS=0: weight gradients on the main stream, as in bench.py's profiled pass -- kernel durations are
# exclusive; with the second stream two kernels share the chip and each row's average is longer than the kernel alone)
rm -rf /tmp/prof7; ( cd /tmp && FI_WGRAD_SIDE_PIXELS=0 FI_DEAD_SIDE=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof7 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --profile-steps 1 --no-cpu-baseline --no-pmc --no-dense-reference --no-issue-profile > $O/${R}_bench_n1_under_rocprof.json 2>/dev/null )
f=$(find /tmp/prof7 -name 'b_kernel_stats.csv' | head -1)
( head -81 $f; grep -E "crop_|nms_|sinkhorn|roi_pool|class_mean|weight_transpose" $f ) | awk '!seen[$0]++' > $O/${R}_bench_n1_kernel_stats.csv
# operator micro-benchmarks, and the rocprofv3 summary of the RoI operators in that run
timeout 300 python scripts/op_bench.py > $O/${R}_op_bench.txt 2>&1
rm -rf /tmp/prof8; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -o c -- python $GRAFT_REPO_ROOT/scripts/op_bench.py --ops crop,nhwc,roipool,nms,sinkhorn > /dev/null 2>&1 )
f=$(find /tmp/prof8 -name 'c_kernel_stats.csv' | head -1); grep -E "Name|crop_|roi_pool|nms_|sinkhorn" $f > $O/${R}_op_bench_kernel_stats.csv
# MFMA utilisation of the dominant convolution (PMC passes, one counter group per run)
bash scripts/conv_prof.sh "FPN P2" > $O/${R}_pmc_conv_mfma.txt 2>&1
timeout 200 python scripts/roipool_probe.py > $O/${R}_roipool_probe.txt 2>&1
# BASELINE configs[4] single-GPU slice (bf16 MFMA convs, 2 x 1344^2, 1000 RoIs/img) and the headline workload on the bf16 kernels
( cd /tmp && timeout 500 python $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline > $O/${R}_bench_cfg5_bf16.json 2>$O/${R}_bench_cfg5_bf16.err )      # with the PMC passes: roofline.traffic
( cd /tmp && timeout 400 python $GRAFT_REPO_ROOT/bench.py --conv-precision bf16 --no-cpu-baseline --no-pmc > $O/${R}_bench_cfg3_bf16.json 2>/dev/null )
# per-layer view of the step's kernels (grid size = layer shape) from a kernel trace of the headline command
rm -rf /tmp/prof9; ( cd /tmp && FI_WGRAD_SIDE_PIXELS=0 FI_DEAD_SIDE=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof9 -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-dense-reference --no-issue-profile > /dev/null 2>&1 )
f=$(find /tmp/prof9 -name 'kt_kernel_trace.csv' | head -1)
python scripts/trace_groups.py $f conv 7 | head -70 > $O/${R}_conv_layers.txt
python scripts/trace_groups.py $f bn_act 7 > $O/${R}_bn_bwd_layers.txt
timeout 200 python scripts/conv_bench.py > $O/${R}_conv_bench.txt 2>&1
timeout 200 python scripts/conv_bench.py --bf16 > $O/${R}_conv_bench_bf16.txt 2>&1
# round 3 additions
timeout 200 python scripts/step_attrib.py > $O/${R}_step_attrib.txt 2>&1
timeout 200 python scripts/conv_shapes.py > $O/${R}_conv_shapes_fp32.txt 2>&1
timeout 200 python scripts/conv_shapes.py --bf16 --cfg5 > $O/${R}_conv_shapes_cfg5_bf16.txt 2>&1
( cd /tmp && timeout 400 python $GRAFT_REPO_ROOT/bench.py --config cfg5 --conv-precision fp16 --no-cpu-baseline --no-pmc > $O/${R}_bench_cfg5_fp16.json 2>/dev/null )
# the data-parallel engine on RCCL with ONE rank (all a 1-GPU box can host): bucket sizes, in-place reduction, overlap object
( cd /tmp && FI_DP_FORCE=1 timeout 400 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc > $O/${R}_bench_dp_force_1rank_rccl.json 2>/dev/null )
# `python bench.py --gpus 2` WITHOUT a launcher (self-launch), two ranks sharing the GPU over gloo (test mode)
( cd /tmp && FI_BENCH_SHARE_GPU=1 timeout 600 python $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 6 --no-cpu-baseline --no-pmc > $O/${R}_bench_gpus2_selflaunch_shared_gpu.json 2>/dev/null )
timeout 200 python scripts/host_time.py 2>&1 | grep -v amdgpu | tail -8 > $O/${R}_host_time.txt
# round 4: the short-K 1x1 probe needs a probe build (FI_EXTRA_HIPCC_FLAGS=-DFI_PROBE_1X1) and is run separately (scripts/c4_probe.sh);
# what a batched weight-gradient launch reaches
timeout 200 python scripts/wg_batch_probe.py 2>&1 | grep "^{" > $O/${R}_wgrad_batch_probe.txt
# GPU idle time between kernels (union over streams) from a kernel trace of the default command; the whole step as one hipGraph
rm -rf /tmp/px; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/px -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-dense-reference --no-issue-profile > /dev/null 2>&1 )
f=$(find /tmp/px -name 'kt_kernel_trace.csv' | head -1); python scripts/gpu_idle.py $f > $O/${R}_gpu_idle.txt 2>&1
timeout 300 python scripts/graph_probe.py 2>&1 | grep -v amdgpu > $O/${R}_graph_probe.txt
FI_STATIC_DEV=1 timeout 300 python scripts/graph_probe.py --cfg5 2>&1 | grep -v amdgpu > $O/${R}_graph_probe_cfg5.txt
# which streams share a hardware queue (plain, and with a 1-rank RCCL group in the process); the engine's own cost with the
# streams picked / not picked; run-to-run repeatability of the default backward form
(python scripts/stream_queues.py; python scripts/stream_queues.py --pg) 2>&1 | grep -v "amdgpu\|socket\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|destroy_process" > $O/${R}_stream_queues.txt
bash scripts/ab_dp.sh 2 "X=1" "FI_PICK_STREAMS=0" "FI_DP_BUCKET_MB=25" 2>&1 | grep -v amdgpu > $O/${R}_ab_dp_final.txt
REP=12 bash scripts/bfp_sweep.sh ${R}
ls -la $O | tail -40
# round 5: BASELINE configs[3]'s per-GPU shape (2 images per GPU) on the one GPU -- the step against its exclusive conv
# time (are the one-workgroup latency kernels of the side streams still hidden at half the conv time?) -- and its idle gaps;
# the data-parallel preflight; the target generation alone on the chip; NCHW pyramid crop backward; LDS atomic rates
( cd /tmp && timeout 400 python $GRAFT_REPO_ROOT/bench.py --batch-per-gpu 2 --no-cpu-baseline --no-pmc --no-dense-reference > $O/${R}_bench_2img_per_gpu.json 2>/dev/null )
rm -rf /tmp/py; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/py -o kt -- python $GRAFT_REPO_ROOT/bench.py --batch-per-gpu 2 --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-dense-reference --no-issue-profile > /dev/null 2>&1 )
f=$(find /tmp/py -name 'kt_kernel_trace.csv' | head -1); python scripts/gpu_idle.py $f > $O/${R}_gpu_idle_2img_per_gpu.txt 2>&1
( cd /tmp && FI_DP_FORCE=1 timeout 200 python $GRAFT_REPO_ROOT/bench.py --preflight > $O/${R}_preflight_1rank_rccl.json 2>/dev/null )
timeout 100 python scripts/rpn_target_probe.py 2>&1 | grep prepare > $O/${R}_rpn_target_probe.txt
(timeout 100 python scripts/pyr_bwd_probe.py; FI_CROP_BWD_SCATTER=1 timeout 100 python scripts/pyr_bwd_probe.py) 2>&1 | grep NCHW > $O/${R}_pyr_bwd_probe.txt
scripts/micro/bin/lds_atomic_rate > $O/${R}_lds_atomic_rate.txt 2>&1
ls -la $O | tail -50
# round 6: the 1x1 ring kernel's development harness (library kernel vs ring variants, decomposition experiments), the patch
# kernel alone on the chip by tile count and by reduction length (fixed cost per launch), what sits on the main stream besides
# the MFMA kernels, the ReLU-boundary events of the backward-form comparison with their evidence
( export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/feature_intertwiner_amd; V2_ONLY=1 timeout 300 scripts/micro/bin/conv1x1_ring > $O/${R}_ring_harness.txt 2>&1 )
timeout 200 python scripts/patch_probe.py 2>&1 | grep "^{" > $O/${R}_patch_probe.txt
timeout 200 python scripts/patch_k_probe.py 2>&1 | grep "^{" >> $O/${R}_patch_probe.txt
timeout 300 python scripts/main_stream_report.py 2>&1 | grep -v "Warning\|amdgpu\|_warn_once" > $O/${R}_main_stream.txt
timeout 300 python scripts/main_stream_report.py --batch 2 2>&1 | grep -v "Warning\|amdgpu\|_warn_once" > $O/${R}_main_stream_2img_per_gpu.txt
timeout 400 python scripts/relu_boundary_probe.py 8 --mask 2>&1 | grep "^{" > $O/${R}_relu_boundary_probe.txt
ls -la $O | tail -20
# GPU idle at 2 images per GPU with the whole step replayed as ONE hipGraph (no host in the loop): kernel trace of replays only
rm -rf /tmp/pz; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pz -o kt -- python $GRAFT_REPO_ROOT/scripts/graph_probe.py --batch 2 --replay 8 > $O/${R}_graph_probe_2img_per_gpu.txt 2>&1 )
f=$(find /tmp/pz -name 'kt_kernel_trace.csv' | head -1); python scripts/gpu_idle.py $f > $O/${R}_gpu_idle_2img_per_gpu_graph.txt 2>&1
ls -la $O | tail -8
# the mask-target crop and the NMS scan alone on the chip (new kernel vs A/B switch), the scan's take-apart builds, the
# shader clock a one-workgroup kernel sees
(timeout 100 python scripts/mask_crop_probe.py; FI_CROP_NO_C1=1 timeout 100 python scripts/mask_crop_probe.py) 2>&1 | grep "^{" > $O/${R}_mask_crop_probe.txt
(timeout 100 python scripts/nms_scan_probe.py; FI_NMS_SCAN_NARROW=1 timeout 100 python scripts/nms_scan_probe.py) 2>&1 | grep "^{" > $O/${R}_nms_scan_probe.txt
timeout 400 bash scripts/nms_exp.sh 2>&1 | grep "^EXP\|^{" >> $O/${R}_nms_scan_probe.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/sclk_probe scripts/micro/sclk_probe.hip && /tmp/sclk_probe > $O/${R}_sclk_probe.txt 2>&1
# proposal selection alone on the chip (default / all-LDS sort / single kernel) with its rocprofv3 rows; Sinkhorn; the 7x7
# channels-last crop backward in both forms; the round's kernels switched off one group at a time at step level
(python scripts/proposal_probe.py; FI_PROPOSAL_LDS_SORT=1 python scripts/proposal_probe.py; FI_PROPOSAL_MULTI_WG=0 python scripts/proposal_probe.py) 2>&1 | grep "^{" > $O/${R}_proposal_probe.txt
rm -rf /tmp/pp; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $GRAFT_REPO_ROOT/scripts/proposal_probe.py > /dev/null 2>&1 )
f=$(find /tmp/pp -name 'p_kernel_stats.csv' | head -1); grep -E "Name|proposal_|fill" $f | cut -c1-130 > $O/${R}_proposal_kernel_stats.csv
timeout 200 python scripts/op_bench.py --ops sinkhorn 2>&1 | grep "^{" > $O/${R}_sinkhorn_probe.txt
( echo "FI_CROP_BWD_GATHER_SMALL=0 python scripts/op_bench.py --ops nhwc"; FI_CROP_BWD_GATHER_SMALL=0 timeout 200 python scripts/op_bench.py --ops nhwc 2>&1 | grep "bwd"; echo "FI_CROP_BWD_GATHER_SMALL=1 python scripts/op_bench.py --ops nhwc"; FI_CROP_BWD_GATHER_SMALL=1 timeout 200 python scripts/op_bench.py --ops nhwc 2>&1 | grep "bwd" ) > $O/${R}_crop_bwd_gather_small.txt
( cd $GRAFT_REPO_ROOT && bash scripts/ab_multi.sh 3 "X=1" "FI_NO_RING1X1=1" "FI_NMS_SCAN_NARROW=1 FI_PROPOSAL_MULTI_WG=0 FI_CROP_NO_C1=1" "FI_CROP_BWD_GATHER_SMALL=0" 2>&1 | grep -v amdgpu > $O/${R}_ab_round_kernels.txt )
