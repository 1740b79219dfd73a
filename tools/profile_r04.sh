#!/bin/bash
# Round-4 profile collection on the GPU box (run from the repo root through gpurun); summaries land in gpurun_out/r04/ and are
# copied into profiles/ by hand.  Kernel timings and PMC counters are separate rocprofv3 runs (counters never together with
# trace domains other than --kernel-trace).  Sections can be selected: bash tools/profile_r04.sh [cfg2] [cfg5] [cfg4] [micro] [lines]
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04; mkdir -p $O
WHAT=${*:-cfg2 cfg5 cfg4 micro lines}
LOSSK="rowkey quantile dense_ step_tail step_post loss_post feat_ pool_ corr_iou compose strided gather_rows ema_ sgd_ densecl_match keys_split"
cd /tmp
for W in cfg2 cfg5 cfg4; do
  case " $WHAT " in *" $W "*) ;; *) continue;; esac
  S=20; [ $W = cfg4 ] && S=12
  # 1. the bench command of the workload under --kernel-trace --stats (CPU baseline leg skipped: it launches no kernels)
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/pk_$W -o b --output-format csv -- python3 $R/bench.py --workload $W --steps $S --warmup 6 --no-cpu-baseline > $O/r04_bench_${W}_under_rocprof.json 2> $O/bench_${W}_rocprof.err || exit 1
  cp /tmp/pk_$W/b_kernel_stats.csv $O/r04_bench_${W}_kernel_stats.csv
  python3 $R/tools/step_breakdown.py /tmp/pk_$W/b_kernel_trace.csv $((S - 2)) > $O/r04_bench_${W}_step_breakdown.txt
  python3 $R/tools/kstats.py /tmp/pk_$W/b_kernel_trace.csv $LOSSK > $O/r04_bench_${W}_loss_kernels.txt
  # 2. matrix-pipe utilisation of the workload's MFMA kernels (cfg5: rows-vs-queue + positive selection; cfg4: the dense kernels)
  if [ $W != cfg2 ]; then
    # (config 4 with fewer steps: the first attempt, 8 + 6 steps of ~1800 launches, died at dispatch 35216 with
    # HSA_STATUS_ERROR_INVALID_PACKET_FORMAT under counter collection; the 14-step config-5 run, 19400 dispatches, is fine)
    PS=8; PW=6; [ $W = cfg4 ] && PS=2 && PW=2
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm_$W -o m --output-format csv -- python3 $R/bench.py --workload $W --steps $PS --warmup $PW --no-cpu-baseline > $O/bench_${W}_pmc.json 2> $O/bench_${W}_pmc.err || exit 1
    # keep the rows of the MFMA kernels only (the whole collection is ~17 MB of BN / convolution launches)
    python3 - /tmp/pm_$W/m_counter_collection.csv $O/r04_${W}_mfma_pmc_counters.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
k = rows[0].index("Kernel_Name")
out = csv.writer(open(sys.argv[2], "w", newline=""))
out.writerow(rows[0])
out.writerows(r for r in rows[1:] if any(s in r[k] for s in ("rowkey", "dense_fwd", "dense_bwd", "densecl_match")))
PY
    python3 $R/tools/mfma_summarize.py /tmp/pm_$W/m_counter_collection.csv rowkey dense_fwd dense_bwd densecl_match > $O/r04_${W}_mfma_util.json
  fi
done
case " $WHAT " in *" micro "*)
  # 3. quantile statistics on the three distributions (stand-alone table + per-kernel durations), the DenseCL positive selection
  timeout -k 10 200 python3 $R/tools/bench_quantiles.py 20 > $O/r04_quantiles_by_distribution.txt 2> $O/quant.err || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/pq -o q --output-format csv -- python3 $R/tools/bench_quantiles.py 8 > $O/quant_prof.log 2>&1 || exit 1
  python3 $R/tools/kstats.py /tmp/pq/q_kernel_trace.csv quantile > $O/r04_quantiles_kernels.txt
  timeout -k 10 200 python3 $R/tools/bench_densecl_match.py 30 > $O/r04_densecl_match.txt 2> $O/match.err || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/pd -o d --output-format csv -- python3 $R/tools/bench_densecl_match.py 10 > $O/match_prof.log 2>&1 || exit 1
  python3 $R/tools/kstats.py /tmp/pd/d_kernel_trace.csv densecl_match >> $O/r04_densecl_match.txt
  # 4. HBM traffic of the two streaming kernels the cfg2 line's roofline entries name (FETCH_SIZE / WRITE_SIZE: separate passes)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/p9_$c -o s --output-format csv -- python3 $R/tools/sgd_only.py > $O/sgd_$c.log 2>&1 || exit 1
    cp /tmp/p9_$c/s_counter_collection.csv $O/r04_sgd_pmc_$c.csv
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/p10_$c -o e --output-format csv -- python3 $R/tools/ema_only.py > $O/ema_$c.log 2>&1 || exit 1
    cp /tmp/p10_$c/e_counter_collection.csv $O/r04_ema_pmc_$c.csv
  done
;; esac
case " $WHAT " in *" lines "*)
  # 5. the bench lines themselves, outside the profiler (with the CPU baseline of the same workload), and the single-rank
  #    rehearsal of the N > 1 path
  cd $R
  python3 bench.py --gpus 1 --steps 50 --warmup 10 > $O/r04_bench_cfg2.json 2> $O/bench_cfg2.err || exit 1
  python3 bench.py --workload cfg5 --steps 40 --warmup 10 > $O/r04_bench_cfg5.json 2> $O/bench_cfg5.err || exit 1
  python3 bench.py --workload cfg4 --steps 20 --warmup 6 > $O/r04_bench_cfg4.json 2> $O/bench_cfg4.err || exit 1
  python3 bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --rehearse-collectives > $O/r04_bench_rehearsal_flatddp.json 2> $O/bench_reh.err || exit 1
;; esac
ls -la $O
