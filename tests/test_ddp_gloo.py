"""CPU, gloo: the bucket bookkeeping of cp2_amd.ddp.GradReducer (the replacement for DistributedDataParallel's reducer,
reference main.py:456-460) against torch's DistributedDataParallel on the same small network: averaged gradients equal
bit for bit with two ranks (to rounding of the summation order with four), no_sync() accumulates locally, a parameter without gradient is reported.  The
local copy is played by tensor operations here (pack_on_host below); the HIP launch is covered by tests/test_gpu_ddp.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cp2_amd import ddp as cddp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pack_on_host(lo, hi, grads, red):
    """The local copy of a bucket written with tensor operations, for the CPU tensors of this test (the product's is the HIP
    launch cp2_pack_grads, checked bit for bit in tests/test_gpu_optim.py)."""
    for t, g in zip(range(lo, hi), grads):
        if g is None:
            red.views[t].zero_()
        else:
            torch.mul(g, red.scale, out=red.views[t])


def _net():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 16, 1),
                               torch.nn.ReLU(), torch.nn.Flatten(), torch.nn.Linear(16 * 6 * 6, 10, bias=True))


def _flatten(net):
    """Re-home the parameters into one flat buffer with 64-element slots, as builder.MODEL.flatten_parameters does."""
    params = list(net.parameters())
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 63) // 64 * 64
    flat = torch.zeros(total)
    for p, o in zip(params, offs):
        v = torch.as_strided(flat, p.shape, p.stride(), o)
        v.copy_(p.data)
        p.data = v
    return params, offs, total


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ref_net, net = _net(), _net()
        net[2].weight.data = net[2].weight.data.contiguous(memory_format=torch.channels_last)   # a [co,ci,1,1] weight: same memory
        ref = torch.nn.parallel.DistributedDataParallel(ref_net)
        params, offs, total = _flatten(net)
        red = cddp.GradReducer(params, offs, total, bucket_mb=100 * 4 / (1 << 20), pack=pack_on_host)
        assert len(red.buckets) >= 3 and red.buckets[0][1] == len(params) and red.buckets[-1][0] == 0
        assert sorted(t for lo, hi in red.buckets for t in range(lo, hi)) == list(range(len(params)))
        assert all(hi_f - lo_f >= 1 for lo_f, hi_f in red.ranges) and red.ranges[0][1] == total and red.ranges[-1][0] == 0
        for step in range(3):
            x = torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(10 * step + rank))
            for p in list(ref_net.parameters()) + params:
                p.grad = None
            ref(x).square().mean().backward()
            net(x).square().mean().backward()
            for i, (a, b) in enumerate(zip(ref_net.parameters(), params)):
                # two ranks: one possible summation order -> equal bit for bit; more: the ring's order depends on where an
                # element sits in its bucket, and DDP's buckets are not these ranges
                assert torch.equal(a.grad, b.grad) if world == 2 else torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-8)
                assert b.grad.data_ptr() == red.views[i].data_ptr()                        # a view of the flat buffer
            with torch.no_grad():
                for a, b in zip(ref_net.parameters(), params):
                    a -= 0.1 * a.grad
                    b -= 0.1 * b.grad
        # gradient accumulation: one local pass under no_sync, then a synchronised one (DDP.no_sync semantics)
        for p in list(ref_net.parameters()) + params:
            p.grad = None
        xs = [torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(500 + 10 * j + rank)) for j in range(2)]
        with ref.no_sync():
            ref(xs[0]).square().mean().backward()
        ref(xs[1]).square().mean().backward()
        red.enabled = False
        net(xs[0]).square().mean().backward()
        local = [p.grad.clone() for p in params]
        red.enabled = True
        net(xs[1]).square().mean().backward()
        for a, b, l in zip(ref_net.parameters(), params, local):
            assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-8) and not torch.equal(b.grad, l)
        # a trainable parameter that gets no gradient is reported (DDP's find_unused_parameters=False behaviour)
        for p in params:
            p.grad = None
        failed = False
        try:
            net[:5](xs[0]).square().mean().backward()          # the Linear layer is not part of this graph
        except RuntimeError as e:
            failed = "received no gradient" in str(e)
        assert failed
        # a backward pass that raises half way (here: inside the first layers' backward, after the last layers' gradients
        # have armed the reducer and started their all-reduce) leaves the reducer armed -- the engine skips its final
        # callbacks: without reset(), which FlatDDP.forward calls, the next pass would not all-reduce at all and the replicas
        # would drift apart silently
        class Boom(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return t.clone()

            @staticmethod
            def backward(ctx, g):
                raise RuntimeError("boom")
        for p in params:
            p.grad = None
        try:
            net[2:](Boom.apply(net[:2](xs[0]))).square().mean().backward()
            boomed = False
        except RuntimeError as e:
            boomed = "boom" in str(e)
        assert boomed and red._armed
        red.reset()
        assert not red._armed
        for p in list(ref_net.parameters()) + params:
            p.grad = None
        x = torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(900 + rank))
        ref(x).square().mean().backward()
        net(x).square().mean().backward()
        for a, b in zip(ref_net.parameters(), params):
            assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-8)
        # every rank planned the same bucket table; a rank with another table is named before the first backward pass
        from cp2_amd import dist as cdist
        cdist.assert_same_on_all_ranks("bucket table", red.table_hash())
        try:
            cdist.assert_same_on_all_ranks("bucket table", red.table_hash() + (1 if rank == world - 1 else 0))
            mismatch = False
        except RuntimeError as e:
            mismatch = f"ranks [{world - 1}]" in str(e)
        assert mismatch
        red.remove_hooks()
        torch.save({"ok": True, "w": params[0].detach().clone()}, os.path.join(out_dir, f"r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_grad_reducer_equals_ddp(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert all(o["ok"] for o in outs) and all(torch.equal(outs[0]["w"], o["w"]) for o in outs)


def test_plan_buckets_covers_every_tensor_once_in_backward_order():
    numels = [9408, 64, 64, 4096, 64, 36864, 16384, 1000, 10]
    b = cddp.plan_buckets(numels, 20000)
    assert b[0][1] == len(numels) and b[-1][0] == 0
    assert all(b[i][0] == b[i + 1][1] for i in range(len(b) - 1))
    assert all(sum(numels[lo:hi]) >= 20000 for lo, hi in b[:-1])
    assert cddp.plan_buckets([5], 100) == [(0, 1)]
    # the range that completes last is cut down to about tail_elems: [0, 3) holds 9536 elements -> (1, 3) then (0, 1)
    numels = [9408, 64, 64, 4096, 64, 36864, 16384, 1000, 10]
    c = cddp.plan_buckets(numels, 50000, 2000)
    assert c == [(5, 9), (1, 5), (0, 1)]
    assert cddp.plan_buckets(numels, 50000, 20000) == [(5, 9), (0, 5)]             # 13696 elements: not worth a cut
    d = cddp.plan_buckets([100] * 40, 1000, 250)
    assert d[-1] == (0, 3) and d[-2][0] == 3 and sorted(t for lo, hi in d for t in range(lo, hi)) == list(range(40))
