#!/bin/bash
# Round-3 profile collection on the GPU box (run from the repo root through gpurun); summaries land in gpurun_out/r03/.
# Kernel timings and PMC counters are separate rocprofv3 runs (counters never together with trace domains other than --kernel-trace).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp
# 1. the bench command under --kernel-trace --stats (CPU baseline leg skipped: it launches no kernels)
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o b --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
cp /tmp/p1/b_kernel_stats.csv $O/r03_bench_n1_kernel_stats.csv
python3 $R/tools/step_breakdown.py /tmp/p1/b_kernel_trace.csv 20 > $O/r03_bench_n1_step_breakdown.txt
python3 $R/tools/kstats.py /tmp/p1/b_kernel_trace.csv rowkey quantile dense_ enqueue feat_ pool_ corr_iou compose strided gather_rows ema_ sgd_ step_scalars > $O/r03_bench_n1_loss_kernels.txt
# 2. quantile statistics: the step's shapes through the cooperative one-launch form and through round 2's row kernel, config 4's shapes
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p6 -o q --output-format csv -- python3 $R/tools/bench_quantiles.py > $O/r03_quantiles.log 2>&1 || exit 1
python3 $R/tools/kstats.py /tmp/p6/q_kernel_trace.csv quantile > $O/r03_quantiles_by_shape.txt
# 3. the on-device input pipeline (photometric kernels)
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p8 -o a --output-format csv -- python3 $R/tools/bench_augment.py > $O/r03_augment.log 2>&1 || exit 1
python3 $R/tools/kstats.py /tmp/p8/a_kernel_trace.csv crop_resize pil_resize resize_coeffs color_kernel blur_tensor erase_rect > $O/r03_augment_kernels.txt
cat $O/r03_augment.log | grep photometric >> $O/r03_augment_kernels.txt
# 4. HBM traffic of the two streaming kernels the bench line's roofline entries name (FETCH_SIZE / WRITE_SIZE: separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/p9_$c -o s --output-format csv -- python3 $R/tools/sgd_only.py > $O/r03_sgd_$c.log 2>&1 || exit 1
  cp /tmp/p9_$c/s_counter_collection.csv $O/r03_sgd_pmc_$c.csv
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/p10_$c -o e --output-format csv -- python3 $R/tools/ema_only.py > $O/r03_ema_$c.log 2>&1 || exit 1
  cp /tmp/p10_$c/e_counter_collection.csv $O/r03_ema_pmc_$c.csv
done
# 5. the multi-GPU plumbing with one rank over RCCL (bench.py --rehearse-collectives): launches and GPU-busy time per step with
#    FlatDDP (default) and with torch DDP, idle gaps, and the host-side cost of enqueuing a step
for GS in flat ddp; do
  timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/p11_$GS -o p --output-format csv -- python3 $R/bench.py --steps 20 --warmup 8 --no-cpu-baseline --rehearse-collectives --grad-sync $GS --nosync-steps 0 > $O/rehearsal_$GS.json 2> $O/rehearsal_$GS.err || exit 1
  python3 $R/tools/step_breakdown.py /tmp/p11_$GS/p_kernel_trace.csv 10 60 > $O/r03_rehearsal_${GS}_step_breakdown.txt
  python3 $R/tools/step_gaps.py /tmp/p11_$GS/p_kernel_trace.csv 10 30 > $O/r03_rehearsal_${GS}_gaps.txt
done
timeout -k 10 300 python3 $R/tools/host_profile.py 30 50 > $O/r03_host_profile.txt 2> $O/host_profile.err || exit 1
# 6. the bench lines themselves, outside the profiler: N = 1 and the single-rank rehearsal of the N > 1 path
cd $R
python3 bench.py --gpus 1 --steps 50 --warmup 10 > $O/r03_bench_n1.json 2> $O/bench_n1.err || exit 1
python3 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --rehearse-collectives > $O/r03_bench_rehearsal_flatddp.json 2> $O/bench_reh.err || exit 1
python3 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --rehearse-collectives --grad-sync ddp > $O/r03_bench_rehearsal_torchddp.json 2>> $O/bench_reh.err || exit 1
ls -la $O
