"""bench.py as the driver runs it.  CPU part: `python bench.py --gpus N` without a launcher starts its own N ranks
(reference main.py:732 spawns its ranks itself) before anything touches a GPU, hands them the launcher environment, and
reports a failing rank through its exit code.  GPU part: the N = 1 line and a two-rank rehearsal on one device (gloo)
carry the fields the contract names, including the per-exchange `comm` breakdown."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--config", os.path.join(ROOT, "configs", "config_pretrain_r18.py"), "--img", "64", "--queue", "1024",
         "--batch-per-gpu", "4", "--steps", "3", "--warmup", "4"]


def _run(argv, env=None, timeout=500):
    e = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=timeout)


def test_self_launch_starts_every_rank_and_reports_a_failing_one():
    """On a box without a GPU every rank stops at bench.py's own "needs a GPU" assertion: that message must come from
    ranks the parent started itself (two of them, each with its RANK), and the parent must exit non-zero."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check (on a GPU box the GPU tests below run the real thing)")
    res = _run(["--gpus", "2", "--no-cpu-baseline"] + SMALL, env={"CP2_BENCH_ECHO_RANK": "1"}, timeout=240)
    assert res.returncode != 0
    assert res.stderr.count("bench.py needs a GPU") >= 1
    assert "exited with code" in res.stderr and "stopping the other ranks" in res.stderr
    assert res.stdout.strip() == ""                       # no partial JSON line


def test_world_size_mismatch_is_refused():
    res = _run(["--gpus", "2"], env={"RANK": "0", "WORLD_SIZE": "4", "LOCAL_RANK": "0"}, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in res.stderr


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_n1_line_has_the_contract_fields():
    res = _run(["--gpus", "1", "--cpu-steps", "1", "--cpu-warmup", "0", "--cpu-batch", "2"] + SMALL)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                 # ONE JSON line on stdout
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and "comm" not in out
    # the labels say what ran: ResNet-18 at 64x64 with a 1024-key queue is no BASELINE configuration
    assert out["metric"] == "pretrain images/sec (whole node), ResNet-18 CP2 64^2, queue=1024"
    assert out["config"]["workload"].startswith("custom sizes (not a BASELINE configuration): ResNet-18 + FCN(contrast) head OS16, 64x64")
    r = out["roofline"]
    assert r["bound"] == "hbm" and "sgd_flat_kernel" in r["kernel"] and r["bytes_per_slot_algorithmic"] == 20
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["bytes_fused"] >= r["bytes_algorithmic"]
    ema = out["roofline_kernels"][0]
    assert "ema_flat_kernel" in ema["kernel"] and ema["bytes_per_slot_algorithmic"] == 12
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["threads"] >= 1 and c["value"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange", ["all_to_all", "all_gather"])
def test_bench_two_ranks_one_device_without_a_launcher(exchange):
    """`python bench.py --gpus 2 --one-device --backend gloo`: no launcher, two ranks on cuda:0 (RCCL refuses that, so
    gloo), whole-job value, comm breakdown with every exchange step timed."""
    res = _run(["--gpus", "2", "--one-device", "--backend", "gloo", "--shuffle-exchange", exchange, "--nosync-steps", "2"] + SMALL)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert "cpu_baseline" not in out                       # rank 0 at N = 1 only
    assert abs(out["value"] - 8 / (out["ms_per_step"] * 1e-3)) < 0.02 * out["value"]     # whole-job images/s
    c = out["comm"]
    assert c["backend"] == "gloo" and c["rccl_ranks"] == 2 and c["launcher"] == "self" and c["shuffle_exchange"] == exchange
    ms = c["ms_per_step"]
    for k in ("c1_image_exchange", "c3_key_unshuffle", "c4_key_gather_enqueue", "step_without_grad_allreduce", "ddp_allreduce_exposed"):
        assert ms[k] is not None, k
    # --overlap auto with peers: both forms were timed after the warm-up, the faster one (max over ranks) ran the timed region
    cal = c["overlap_calibration"]
    assert cal["steps_each"] >= 5 and set(cal["ms_per_step"]) == {"off", "gather"} and all(v > 0 for v in cal["ms_per_step"].values())
    assert cal["chosen"] == min(cal["ms_per_step"], key=cal["ms_per_step"].get) == c["overlap_used"]
    assert (ms["key_branch_wait_exposed"] is None) == (c["overlap_used"] == "off")    # one stream: nothing to wait for
    pr = c["per_rank"]
    assert len(pr["ms_per_step"]["by_rank"]) == 2 and pr["ms_per_step"]["max"] <= out["ms_per_step"] * 1.05
    assert len(pr["host_issue_ms_per_step"]["by_rank"]) == 2 and pr["host_issue_ms_per_step"]["min"] > 0
    assert c["grad_sync"] == "flat" and c["grad_buckets"] >= 1 and c["timeout_s"] == 120.0
    assert ms["c1_image_exchange"] > 0 and ms["c3_key_unshuffle"] > 0 and ms["c4_key_gather_enqueue"] > 0
    by = c["bytes_received_per_rank_per_step"]
    row = 3 * 64 * 64 * 2                                  # composed images travel in bf16 under bf16 autocast
    assert by["c1_image_exchange"] == (2 * row if exchange == "all_to_all" else 4 * row)


def test_workload_labels_are_built_from_what_runs():
    """--workload cfg2 | cfg4 | cfg5 pick the BASELINE configurations; explicit sizes drop the BASELINE label (round 3's cfg4
    line carried cfg2's).  Parsed here without a GPU."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    old = sys.argv
    try:
        got = {}
        for wl in ("cfg2", "cfg4", "cfg5"):
            sys.argv = ["bench.py", "--workload", wl]
            a = bench.parse()
            got[wl] = (os.path.basename(a.config), a.img, a.queue, a.batch_per_gpu, a.densecl, a.as_baseline)
        assert got["cfg2"] == ("config_pretrain_r50_fcn.py", 224, 65536, 32, False, True)
        assert got["cfg4"] == ("config_pretrain_r101_d8.py", 512, 131072, 8, False, True)
        assert got["cfg5"] == ("config_pretrain.py", 224, 65536, 32, True, True)       # scripts/10-11-densecl.sh:13
        sys.argv = ["bench.py", "--workload", "cfg4", "--img", "256"]
        assert bench.parse().as_baseline is False
    finally:
        sys.argv = old


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("wl,metric,kernel", [("cfg5", "ResNet-18 DenseCL 64^2, queue=1024", "rowkey"),
                                              ("cfg4", "ResNet-18 CP2 64^2, queue=1024", "dense_bwd_kernel")])
def test_bench_other_workloads_run_and_label_themselves(wl, metric, kernel):
    """The cfg4 / cfg5 code paths of bench.py at toy sizes: the line names the method and sizes that ran and carries that
    workload's roofline kernel and a CPU baseline of the same kind of step."""
    argv = ["--gpus", "1", "--workload", wl, "--cpu-steps", "1", "--cpu-warmup", "0", "--cpu-batch", "2"] + SMALL
    if wl == "cfg5":                # DenseCL needs a 2048-channel backbone (reference builder.py:408): ResNet-50, stride 32
        argv = [a if a != SMALL[1] else os.path.join(ROOT, "configs", "config_moco.py") for a in argv]
        metric = metric.replace("ResNet-18", "ResNet-50")
        argv[argv.index("--batch-per-gpu") + 1] = "24"          # 24 x 4 = 96 rows: the many-rows kernel (R > 32)
    res = _run(argv, timeout=850)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert out["metric"].endswith(metric), out["metric"]
    assert kernel in out["roofline"]["kernel"] and out["roofline"]["bound"] == "mfma" and out["roofline"]["frac"] > 0
    assert out["cpu_baseline"]["value"] > 0 and ("DenseCL" in out["cpu_baseline"]["sample"]) == (wl == "cfg5")
    assert ("DenseCL" in out["config"]["workload"]) == (wl == "cfg5")


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("grad_sync", ["flat", "ddp"])
def test_bench_single_rank_rccl_rehearsal(grad_sync):
    """`--rehearse-collectives`: the N > 1 code path (exchange steps, gradient buckets, RCCL calls) with ONE rank on a one-GPU
    box; the line is an N = 1 line plus the `comm` breakdown, and says so."""
    res = _run(["--gpus", "1", "--no-cpu-baseline", "--rehearse-collectives", "--grad-sync", grad_sync, "--nosync-steps", "2"] + SMALL)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    c = out["comm"]
    assert out["n_gpus"] == 1 and c["single_rank_rehearsal"] is True and c["backend"] == "nccl" and c["rccl_ranks"] == 1
    assert c["grad_sync"] == grad_sync and (c["grad_buckets"] >= 1 if grad_sync == "flat" else c["grad_buckets"] is None)
    ms = c["ms_per_step"]
    assert ms["c1_image_exchange"] > 0 and ms["c3_key_unshuffle"] > 0 and ms["c4_key_gather_enqueue"] > 0
    assert ms["step_without_grad_allreduce"] > 0 and out["host_issue_ms_per_step"] > 0
