#!/bin/bash
# Round-2 profile collection on the GPU box (run from the repo root through gpurun); summaries land in gpurun_out/r02/.
# Kernel timings and PMC counters are separate rocprofv3 runs (counters never together with trace domains other than --kernel-trace).
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
# 1. the bench command under --kernel-trace --stats (CPU baseline leg skipped: it launches no kernels)
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o b --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
cp /tmp/p1/b_kernel_stats.csv $O/r02_bench_n1_kernel_stats.csv
python3 tools/step_breakdown.py /tmp/p1/b_kernel_trace.csv 20 > $O/r02_bench_n1_step_breakdown.txt
python3 tools/kstats.py /tmp/p1/b_kernel_trace.csv rowkey quantile dense_ enqueue feat_ pool_ corr_iou compose strided gather_rows ema_ sgd_ > $O/r02_bench_n1_loss_kernels.txt
python3 tools/kstats.py /tmp/p1/b_kernel_trace.csv wgrad maxpool reduce_kernel > $O/r02_bench_n1_encoder_extras.txt
# 2. the MFMA-bound sizes (config 4 / 5) and the instance kernel alone
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o k --output-format csv -- python3 tools/bench_kernels.py > $O/r02_bench_kernels.log 2>&1 || exit 1
cp /tmp/p2/k_kernel_stats.csv $O/r02_kernels_kernel_stats.csv
python3 tools/kstats.py /tmp/p2/k_kernel_trace.csv rowkey dense_ keys_split quantile > $O/r02_kernels_by_shape.txt
# 3. MFMA utilisation counters
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d /tmp/p3 -o m --output-format csv -- python3 tools/mfma_prof.py > $O/r02_mfma_prof.log 2>&1 || exit 1
cp /tmp/p3/m_counter_collection.csv $O/r02_mfma_pmc_counters.csv
python3 tools/mfma_summarize.py /tmp/p3/m_counter_collection.csv rowkey dense_fwd dense_bwd > $O/r02_mfma_util.json
# 4. HBM traffic of the instance kernel (FETCH_SIZE and WRITE_SIZE need separate passes)
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p4 -o f --output-format csv -- python3 tools/bench_instance.py > $O/r02_inst_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p5 -o w --output-format csv -- python3 tools/bench_instance.py > $O/r02_inst_write.log 2>&1 || exit 1
python3 - <<'PY' > $O/r02_rowkey_small_traffic.json
import csv, json
def avg(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if "rowkey_small_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    # first 105 launches: R=32, K=65536; next 105: R=8, K=131072 (tools/bench_instance.py)
    a = [float(r["Counter_Value"]) for r in rows[5:105]]; b = [float(r["Counter_Value"]) for r in rows[110:210]]
    return sum(a) / len(a), sum(b) / len(b)
f, w = avg("/tmp/p4/f_counter_collection.csv", "FETCH_SIZE"), avg("/tmp/p5/w_counter_collection.csv", "WRITE_SIZE")
out = {}
for i, (R, K) in enumerate(((32, 65536), (8, 131072))):
    out[f"R={R},K={K}"] = {"FETCH_SIZE_KiB_raw": f[i], "WRITE_SIZE_KiB_raw": w[i],
                           "hbm_bytes_per_launch": f[i] * 1024 * 2 + w[i] * 1024,
                           "algorithmic_bytes_per_launch": 4 * 128 * K, "partials_written_bytes": 256 * (128 + 3) * R * 4}
out["correction"] = "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as reported"
out["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/bench_instance.py, launches 6-105 / 111-210"
print(json.dumps(out, indent=1))
PY
# 5. quantile statistics: the step's shapes (one-launch row kernel) and BASELINE config 4's shapes (chunked three-launch path)
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p6 -o q --output-format csv -- python3 tools/bench_quantiles.py > $O/r02_quantiles.log 2>&1 || exit 1
python3 tools/kstats.py /tmp/p6/q_kernel_trace.csv quantile > $O/r02_quantiles_by_shape.txt
# 6. supervised CutPaste / mirror path: composition and loss kernels at 10 x 512 x 512
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o m --output-format csv -- python3 tools/bench_mirror.py > $O/r02_mirror.log 2>&1 || exit 1
python3 tools/kstats.py /tmp/p7/m_kernel_trace.csv cutpaste mirror_loss > $O/r02_mirror_kernels.txt
cat $O/r02_mirror.log | grep classes >> $O/r02_mirror_kernels.txt
cp /tmp/p4/f_counter_collection.csv $O/r02_rowkey_small_pmc_fetch_size.csv
cp /tmp/p5/w_counter_collection.csv $O/r02_rowkey_small_pmc_write_size.csv
ls -la $O
