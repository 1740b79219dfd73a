"""GPU: the N > 1 code path end to end on ONE device -- two ranks share cuda:0 and talk over gloo (RCCL refuses two
ranks on one GPU; the 8-GPU run itself belongs to the driver).  Checks what must hold on any backend: every rank
ends with the same queue / pointer, gradients are synchronised by DDP, key branch on the side stream, loss finite."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, mode="torch-sgd"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    from cp2_amd import builder, synthetic
    from cp2_amd.config import Config
    from cp2_amd.pretrain_types import PretrainType
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        densecl = mode.startswith("densecl")
        cfg = Config.fromfile(os.path.join(ROOT, "configs", "config_moco.py" if densecl else "config_pretrain_r18.py"))
        extra = dict(pretrain_type=PretrainType.CP2)
        if densecl:      # BASELINE config 5; "-v2": the symmetric variant with the predictor heads in use
            extra = dict(pretrain_type=PretrainType.DENSECL, lmbd_cp2_dense_loss=0.5, dense_logits_temp=0.2)
            if mode == "densecl-v2":
                extra.update(pretrain_type=PretrainType.PROPOSED_V2, use_symmetrical_loss=True, use_avgpool_global=True)
        model = builder.MODEL(cfg, rank=rank, K=256, pretrain_from_scratch=True, device=dev,
                              amp_dtype=torch.bfloat16, channels_last=True, **extra).to(dev).train()
        model.encoder_q.to(memory_format=torch.channels_last)
        model.encoder_k.to(memory_format=torch.channels_last)
        if mode == "exchange-equality":
            # shuffle-BN by all-to-all (only the rows a rank keeps travel) == the reference's all-gather form, bit for bit,
            # for the images going in and the keys coming back (gather_rows kernels + the collectives, two ranks)
            from cp2_amd import dist as cdist
            for it in range(4):
                x = torch.randn(6, 3, 32, 32, device=dev, generator=torch.Generator(dev).manual_seed(10 * rank + it))
                perm = cdist.make_shuffle_index(12, dev)                       # rank 0's permutation on every rank
                model.shuffle_exchange = "all_to_all"
                a, plan = model._batch_shuffle_ddp(x, perm)
                model.shuffle_exchange = "all_gather"
                g, idx_un = model._batch_shuffle_ddp(x, perm)
                assert isinstance(plan, cdist.ShufflePlan) and torch.equal(a, g)
                k = a * 3.0 - 1.0
                assert torch.equal(model._batch_unshuffle_ddp(k, plan), model._batch_unshuffle_ddp(k, idx_un))
                assert torch.equal(model._batch_unshuffle_ddp(k, plan), x * 3.0 - 1.0)
            torch.save({"ok": True}, os.path.join(out_dir, f"r{rank}.pt"))
            dist.barrier()
            return
        if mode == "sync-bn":
            # the mirror path with more than one rank (reference mirror_pretrain.py:229-231, sync_batchnorm=True): every
            # BatchNorm of the encoder takes its batch statistics over ALL ranks -- outputs and running statistics equal
            # one process running plain BatchNorm on the concatenated batch
            from cp2_amd.encoder import SyncFusedBatchNorm2d, build_segmentor, convert_sync_batchnorm
            torch.manual_seed(3)
            ref = build_segmentor(cfg.model).to(dev).train()
            net = build_segmentor(cfg.model).to(dev).train()
            net.load_state_dict(ref.state_dict())
            keys = list(net.state_dict())
            net = convert_sync_batchnorm(net)
            assert list(net.state_dict()) == keys                               # same state-dict keys
            n_sync = sum(isinstance(m, SyncFusedBatchNorm2d) for m in net.modules())
            assert n_sync > 15 and not any(type(m).__name__ == "FusedBatchNorm2d" for m in net.modules())
            xs = [torch.randn(3, 3, 64, 64, device=dev, generator=torch.Generator(dev).manual_seed(50 + r)) for r in range(world)]
            y = net(xs[rank])
            y_ref = ref(torch.cat(xs))[3 * rank:3 * rank + 3]
            assert (y - y_ref).abs().max().item() <= 2e-4 * y_ref.abs().max().item() + 1e-6
            a, b2 = net.backbone.layer2[0].bn1, ref.backbone.layer2[0].bn1
            assert torch.allclose(a.running_mean, b2.running_mean, atol=1e-5) and torch.allclose(a.running_var, b2.running_var, rtol=1e-4, atol=1e-6)
            torch.save({"ok": True}, os.path.join(out_dir, f"r{rank}.pt"))
            dist.barrier()
            return
        if mode == "score-stats":
            # the DenseCL score statistics shared out over the ranks (builder.row_score_stats_over_ranks) == the same
            # statistics taken by one rank alone over rank 0's rows: per-row results bit for bit, so the means agree to 1e-6
            from cp2_amd import ops
            g = torch.Generator(dev).manual_seed(1000 + rank)                  # every rank holds different rows
            rows = torch.nn.functional.normalize(torch.randn(5, 128, 49, device=dev, generator=g), dim=1)
            queue = torch.nn.functional.normalize(torch.randn(128, 2048, device=dev, generator=torch.Generator(dev).manual_seed(7)), dim=0)
            got = builder.row_score_stats_over_ranks(rows, queue)
            src = rows.clone()
            dist.broadcast(src, 0)
            res = ops.rowkey_infonce(src, (49, 128 * 49, 1, 49), 5 * 49, queue, torch.zeros(5 * 49, 1, device=dev), 1.0,
                                     grad_scale=None, want_lneg=True, lneg_row_major=True, precision="f32")
            q = ops.masked_quantiles(res.lneg, 2048, 1, 5 * 49, 2048)
            want = torch.cat([res.lneg.mean(1).mean().reshape(1), q.mean(1)])
            assert (got - want).abs().max().item() <= 1e-6, (got, want)
            torch.save({"ok": True, "stats": got.cpu()}, os.path.join(out_dir, f"r{rank}.pt"))
            dist.barrier()
            return
        if mode == "flat-gather-allgather":
            model.shuffle_exchange = "all_gather"
        if mode in ("flat-inline", "densecl-v2", "flatddp-vs-ddp"):     # cp2_amd.ddp.FlatDDP: what main.py / bench.py use by default
            from cp2_amd.ddp import FlatDDP
            ddp = FlatDDP(model, bucket_mb=4)
            assert len(ddp.reducer.buckets) >= 3
        else:
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, broadcast_buffers=False,
                                                            gradient_as_bucket_view=True)
        if mode == "flatddp-vs-ddp":
            # what DistributedDataParallel would leave in p.grad, from the SAME backward pass (two passes differ by the decode
            # head's dropout mask and the atomics of the weight-gradient kernels): the local gradients are recorded as the pack
            # launch sees them, scaled by 1/2 and summed over the two ranks by the test -- equal bit for bit (two ranks: one
            # summation order), every p.grad a view of the flat gradient buffer; no_sync() leaves the local gradient alone
            from cp2_amd import ops
            red, seen, pack = ddp.reducer, {}, ops.pack_grads

            def spy(plan, flat, ptrs, lo, hi, scale):
                assert scale == 0.5
                for t in range(lo, hi):
                    if red.params[t].requires_grad:
                        seen[t] = red.params[t].grad.detach().clone()
                return pack(plan, flat, ptrs, lo, hi, scale)
            ops.pack_grads = spy
            for step in range(2):
                batch = synthetic.make_batch(4, 64, 64, dev, seed=100 * rank + step)
                model.zero_grad(set_to_none=True)
                seen.clear()
                ddp(visualize=False, step=step, new_epoch=False, **batch).backward()
                assert sorted(seen) == [t for t, p in enumerate(red.params) if p.requires_grad] and len(seen) > 20
                for t, local in seen.items():
                    want = local * 0.5
                    dist.all_reduce(want)
                    p = red.params[t]
                    assert torch.equal(p.grad, want) and float(want.abs().max()) > 0, t
                    assert p.grad.data_ptr() == red.views[t].data_ptr() and p.grad.stride() == p.stride()
            ops.pack_grads = pack
            assert ddp.reducer.layout_copies == 0
            model.zero_grad(set_to_none=True)
            with ddp.no_sync():
                ddp(visualize=False, step=9, new_epoch=False, **batch).backward()
            g = model.encoder_q.backbone.conv1.weight.grad
            assert g.data_ptr() != ddp.reducer.views[0].data_ptr()
            gathered = [torch.empty_like(g) for _ in range(world)]
            dist.all_gather(gathered, g.contiguous())
            assert not torch.equal(gathered[0], gathered[1])                  # local gradients: different data on each rank
            torch.save({"ok": True}, os.path.join(out_dir, f"r{rank}.pt"))
            dist.barrier()
            return
        steps = 3
        if mode in ("torch-sgd", "densecl", "densecl-v2"):
            opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4)
        else:        # the bench / main.py configuration: FlatSGD, enough steps for the key-forward hipGraph to be replayed
            from cp2_amd.optim import FlatSGD
            opt = FlatSGD(ddp, 0.01, momentum=0.9, weight_decay=1e-4)
            # "flat-inline": the default (everything in order on one stream); the side-stream forms stay covered
            model.overlap_key_branch = {"flat-inline": None, "flat-gather": "gather", "flat-gather-allgather": "gather"}[mode]
            steps = 6
        b = 6
        for step in range(steps):
            batch = synthetic.make_batch(b, 64, 64, dev, seed=100 * rank + step)     # different data on each rank
            loss = ddp(visualize=False, step=step, new_epoch=False, **batch)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            assert torch.isfinite(loss)
        assert (model._side_stream is not None) == (mode in ("flat-gather", "flat-gather-allgather"))
        if mode.startswith("flat"):
            assert model._key_graph is not None and any(e["graph"] is not None for e in model._key_graph.entries.values())
        torch.cuda.synchronize()
        g = model.encoder_q.backbone.conv1.weight.grad.detach().float().cpu()
        torch.save({"queue": model.queue.cpu(), "ptr": int(model.queue_ptr), "grad": g, "steps": steps,
                    "w": model.encoder_q.backbone.conv1.weight.detach().cpu(),
                    "k": model.encoder_k.backbone.conv1.weight.detach().cpu()}, os.path.join(out_dir, f"r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["torch-sgd", "flat-inline", "flat-gather", "flat-gather-allgather", "densecl", "densecl-v2"])
def test_two_ranks_one_device_gloo(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["ptr"] == r1["ptr"] == (r0["steps"] * 2 * 6) % 256
    assert torch.equal(r0["queue"], r1["queue"])                  # identical enqueue on every rank, in rank order
    assert torch.equal(r0["grad"], r1["grad"]) and float(r0["grad"].abs().max()) > 0   # DDP-averaged gradients
    assert torch.equal(r0["w"], r1["w"]) and torch.equal(r0["k"], r1["k"])             # replicas stay in lock-step


@pytest.mark.timeout(300)
def test_flat_ddp_gradients_are_the_rank_average_of_this_backward_pass(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "flatddp-vs-ddp"), nprocs=2, join=True)
    assert torch.load(tmp_path / "r0.pt")["ok"] and torch.load(tmp_path / "r1.pt")["ok"]


@pytest.mark.timeout(300)
def test_sync_batchnorm_conversion_equals_one_process_on_the_whole_batch(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "sync-bn"), nprocs=2, join=True)
    assert torch.load(tmp_path / "r0.pt")["ok"] and torch.load(tmp_path / "r1.pt")["ok"]


@pytest.mark.timeout(300)
def test_score_statistics_shared_over_ranks_equal_one_rank(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "score-stats"), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["ok"] and r1["ok"] and torch.equal(r0["stats"], r1["stats"])     # the all-reduce leaves them on every rank


@pytest.mark.timeout(300)
def test_shuffle_by_all_to_all_equals_all_gather_form(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "exchange-equality"), nprocs=2, join=True)
    assert torch.load(tmp_path / "r0.pt")["ok"] and torch.load(tmp_path / "r1.pt")["ok"]
