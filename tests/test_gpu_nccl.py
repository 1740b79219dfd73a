"""GPU: the RCCL ("nccl") code path with the one rank a 1-GPU box allows -- the very calls the 8-GPU run makes
(dist.concat_all_gather -> all_gather_into_tensor, the shuffle-BN all_to_all_single, index / seed broadcast, DDP buckets with gradient_as_bucket_view feeding
FlatSGD), so the driver's multi-GPU bench does not meet them for the first time.  World size 1 makes every collective
an identity; what is checked is that the calls are accepted by this torch / RCCL build on device tensors, run on the
side stream, and leave the same state as the single-process path."""
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


def _worker(rank, port, out_dir, grad_sync):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)       # cdist.init_process_group must put it there itself
    import sys
    sys.path.insert(0, ROOT)
    from cp2_amd import builder, dist as cdist, synthetic
    from cp2_amd.config import Config
    from cp2_amd.optim import FlatSGD
    from cp2_amd.pretrain_types import PretrainType
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cdist.init_process_group("nccl", 0, 1, timeout_s=100)    # a one-rank group: every exchange step of the N > 1 path is issued
    try:
        assert dist.get_backend() == "nccl" and cdist.multi() and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        x = torch.arange(24, dtype=torch.float32, device=dev).reshape(6, 4)
        out = torch.empty_like(x)
        dist.all_gather_into_tensor(out, x)                  # the branch dist.concat_all_gather takes on RCCL
        assert torch.equal(out, x)
        idx = cdist.make_shuffle_index(6, dev)               # randperm + broadcast from rank 0
        assert sorted(idx.tolist()) == list(range(6))
        # the shuffle-BN exchange as the multi-GPU run makes it: seed broadcast once, all_to_all_single with row counts
        from cp2_amd import ops
        perm = cdist.shared_permutation(6, dev)
        plan = cdist.ShufflePlan(perm, 0, 1)
        y = cdist.exchange_rows(x, plan, take=ops.gather_rows)
        assert torch.equal(y, x[perm.to(dev)])
        assert torch.equal(cdist.exchange_rows(y, plan, backward=True, take=ops.gather_rows), x)
        torch.manual_seed(0)
        cfg = Config.fromfile(os.path.join(ROOT, "configs", "config_pretrain_r18.py"))
        model = builder.MODEL(cfg, rank=0, K=256, pretrain_from_scratch=True, pretrain_type=PretrainType.CP2, device=dev,
                              amp_dtype=torch.bfloat16, channels_last=True).to(dev).train()
        model.encoder_q.to(memory_format=torch.channels_last)
        model.encoder_k.to(memory_format=torch.channels_last)
        if grad_sync == "flat":                               # the default of main.py / bench.py
            from cp2_amd.ddp import FlatDDP
            ddp = FlatDDP(model, bucket_mb=4)
        else:
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, broadcast_buffers=False,
                                                            gradient_as_bucket_view=True)
        opt = FlatSGD(ddp, 0.01, momentum=0.9, weight_decay=1e-4)
        assert model.overlap_key_branch is None               # -> one stream, what every world size selects by default
        losses = []
        cdist.barrier("test: asynchronous barrier over RCCL")     # the tracked forms bench.py uses around its timed region
        cdist.assert_same_on_all_ranks("test value", 12345, dev)
        for step in range(4):
            cdist.progress(step)
            if step == 2:
                cdist.steady()                                    # start-up limit -> steady limit (_set_pg_timeout on the RCCL group)
                assert cdist._LIMITS[3] and cdist.COLLECTIVES.first_incomplete() is None
            batch = synthetic.make_batch(6, 64, 64, dev, seed=step)
            loss = ddp(visualize=False, step=step, new_epoch=False, **batch)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert all(l == l for l in losses) and int(model.queue_ptr) == 24
        assert model._side_stream is None                     # default: no side stream
        if grad_sync == "flat":                               # every gradient went through pack + RCCL all_reduce into the flat buffer
            red = ddp.reducer
            assert len(red.buckets) >= 3 and red.layout_copies == 0
            for i, p in enumerate(red.params):
                assert not p.requires_grad or p.grad.data_ptr() == red.views[i].data_ptr()
            assert float(red.flat.abs().sum()) > 0
        model.overlap_key_branch = "gather"                   # the side-stream form stays available: two more steps through it
        for step in range(4, 6):
            batch = synthetic.make_batch(6, 64, 64, dev, seed=step)
            loss = ddp(visualize=False, step=step, new_epoch=False, **batch)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert all(l == l for l in losses) and int(model.queue_ptr) == 36 and model._side_stream is not None
        # every exchange step went through the collective log (what the hang watchdog reads), all of them completed
        names = " | ".join(it[1] for it in cdist.COLLECTIVES.items)
        assert "all_to_all" in names and "all_gather" in names and (grad_sync != "flat" or "c5 gradient all-reduce, bucket" in names)
        assert cdist.COLLECTIVES.first_incomplete() is None
        torch.save({"losses": losses}, os.path.join(out_dir, "nccl.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("grad_sync", ["flat", "ddp"])
def test_single_rank_rccl_path(tmp_path, grad_sync):
    mp.spawn(_worker, args=(_free_port(), str(tmp_path), grad_sync), nprocs=1, join=True)
    assert len(torch.load(tmp_path / "nccl.pt")["losses"]) == 6
