"""CPU, world size 2 over gloo: the collective plumbing of the hot path (reference builder.py:609-649,
569-587, 1710-1722).  The device kernels are not involved (they have no CPU path); the row gathers and the
enqueue are played by the oracle so the multi-rank index logic is checked end to end."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cp2_amd import dist as cdist
from oracle import cp2_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = 3
        torch.manual_seed(100 + rank)                       # ranks deliberately hold different RNG states
        x = torch.arange(b * 4, dtype=torch.float32).reshape(b, 4) + 1000 * rank
        # concat_all_gather: rank order on dim 0, no grad
        g = cdist.concat_all_gather(x.clone().requires_grad_(True))
        assert g.shape == (world * b, 4) and not g.requires_grad
        for r in range(world):
            assert torch.equal(g[r * b:(r + 1) * b], torch.arange(b * 4, dtype=torch.float32).reshape(b, 4) + 1000 * r)
        # shuffle index: every rank ends up with rank 0's permutation
        idx = cdist.make_shuffle_index(world * b, "cpu")
        ref = idx.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(idx, ref) and sorted(idx.tolist()) == list(range(world * b))
        # shuffle -> per-sample "encoder" -> unshuffle restores this rank's own samples, in order
        taken = O.shuffle_take(g, idx, rank, world)
        assert torch.equal(taken, g[cdist.shuffle_rows_for_rank(idx, rank, world)])
        enc = taken * 2.0 + 1.0
        enc_all = cdist.concat_all_gather(enc)
        idx_un, rows = cdist.unshuffle_rows_for_rank(idx, rank, world)
        restored = enc_all[rows]
        assert torch.equal(restored, x * 2.0 + 1.0)
        assert torch.equal(restored, O.unshuffle_take(enc_all, idx, rank, world))
        # the same shuffle / un-shuffle by all-to-all (only the rows a rank keeps travel): equal to the all-gather form
        # for the broadcast permutation and for permutations drawn from the shared host generator
        for it in range(6):
            perm = idx if it == 0 else cdist.shared_permutation(world * b)
            chk = perm.clone()
            dist.broadcast(chk, src=0)
            assert torch.equal(perm, chk)                   # every rank drew the same permutation, no per-step broadcast
            plan = cdist.ShufflePlan(perm, rank, world)
            assert sum(plan.send_counts) == b == sum(plan.recv_counts)
            got = cdist.exchange_rows(x, plan)
            assert torch.equal(got, O.shuffle_take(g, perm, rank, world))
            back = cdist.exchange_rows(got * 2.0 + 1.0, plan, backward=True)
            assert torch.equal(back, x * 2.0 + 1.0)
            # the forms the training step uses: the composition kernel already wrote the rows in send order (presorted),
            # and the returned keys stay in arrival order -- the loss kernel reads sample i at row back_place[i] (keep_order)
            assert torch.equal(cdist.exchange_rows(x[plan.send_rows], plan, presorted=True), got)
            arrived = cdist.exchange_rows(got * 2.0 + 1.0, plan, backward=True, keep_order=True)
            assert torch.equal(arrived[plan.back_place], x * 2.0 + 1.0)
            assert plan.bytes_received(16) == 16 * (b - plan.recv_counts[rank])
        # enqueue: every rank appends ALL ranks' keys in rank order -> replicas of the queue stay identical
        K, C = 32, 4
        queue = torch.zeros(C, K)
        ptr = 29                                           # crosses the wrap boundary
        queue, ptr = O.dequeue_and_enqueue(queue, ptr, cdist.concat_all_gather(x))
        assert ptr == (29 + world * b) % K
        torch.save(queue, os.path.join(out_dir, f"q{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_world_n_gloo_collectives(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    q0, q1 = torch.load(tmp_path / "q0.pt"), torch.load(tmp_path / "q1.pt")
    assert torch.equal(q0, q1)
    want = torch.cat([torch.arange(12, dtype=torch.float32).reshape(3, 4) + 1000 * r for r in range(world)])
    cols = [(29 + i) % 32 for i in range(world * 3)]
    assert torch.equal(q0[:, cols], want.t())


def _hang_worker(rank, world, port, out_dir):
    """Rank 1 never joins the second collective: rank 0's watchdog must name it and end the process with HANG_EXIT_CODE."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cdist.init_process_group("gloo", rank, world, timeout_s=30, watchdog=False)
    msgs = []
    cdist.start_watchdog(1.5, poll_s=0.2, on_hang=msgs.append)
    cdist.progress(0)
    x = torch.ones(4) * rank
    w = dist.all_reduce(x, async_op=True)
    cdist.COLLECTIVES.note("c1_image_exchange", w)
    w.wait()
    cdist.progress(1)
    if rank == 0:
        w2 = dist.all_reduce(x, async_op=True)               # rank 1 never issues this one
        cdist.COLLECTIVES.note("c3_key_unshuffle", w2)
        import time
        t0 = time.time()
        while not msgs and time.time() - t0 < 20:
            time.sleep(0.1)
        with open(os.path.join(out_dir, "hang.txt"), "w") as f:
            f.write(msgs[0] if msgs else "no message")
        os._exit(0)                                          # the pending collective can never complete
    else:
        import time
        time.sleep(6)
        os._exit(0)


@pytest.mark.timeout(120)
def test_watchdog_names_the_collective_that_did_not_complete(tmp_path):
    ctx = mp.spawn(_hang_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=False)
    for p in ctx.processes:
        p.join(60)
    text = open(tmp_path / "hang.txt").read()
    assert "rank 0 of 2 made no progress" in text and "at step 1" in text
    assert "first collective not completed: c3_key_unshuffle (step 1)" in text


def test_watchdog_leaves_a_busy_host_alone_and_knows_start_up_from_steady_state():
    """What the watchdog judges is a PENDING collective, not the time since the last step: a rank that spends minutes in
    MIOpen's solver search, a checkpoint or validation has nothing pending and must not be shot (a 4-ranks-on-one-GPU
    rehearsal of bench.py died in its first step under the old "no progress for 96 s" rule); and a collective may stay
    pending longer during start-up (peers still searching) than in steady state."""
    import time

    class Pending:
        def is_completed(self):
            return False

    msgs = []
    cdist.COLLECTIVES.clear()
    cdist._WATCHDOG = None
    cdist._LIMITS[3] = False
    try:
        cdist.start_watchdog(0.6, poll_s=0.1, on_hang=msgs.append, startup_s=30.0)
        cdist.progress(0)
        time.sleep(1.5)                                    # "no progress" for 2.5 x the steady limit, nothing pending
        assert not msgs
        cdist.COLLECTIVES.note("c1 all-to-all of the shuffled rows", Pending())
        time.sleep(1.5)                                    # pending beyond the steady limit, but the run is still starting up
        assert not msgs
        cdist._LIMITS[3] = True                            # what steady() sets (it also lowers the group's timeout: none here)
        t0 = time.time()
        while not msgs and time.time() - t0 < 5:
            time.sleep(0.05)
        assert msgs and "first collective not completed: c1 all-to-all of the shuffled rows (step 0), pending for" in msgs[0]
    finally:
        cdist.COLLECTIVES.clear()
        cdist._WATCHDOG = None
        cdist._LIMITS[3] = False


def test_shuffle_plan_refuses_a_table_that_is_no_permutation():
    with pytest.raises(ValueError, match="not a permutation"):
        cdist.ShufflePlan(torch.tensor([0, 1, 1, 3]), 0, 2)
    with pytest.raises(ValueError, match="not a permutation"):
        cdist.ShufflePlan(torch.tensor([0, 1, 2, 7]), 0, 2)


def test_single_process_helpers_without_process_group():
    x = torch.randn(4, 3)
    assert cdist.world_size() == 1 and cdist.rank() == 0
    assert cdist.concat_all_gather(x) is x
    idx = cdist.make_shuffle_index(8, "cpu", generator=torch.Generator().manual_seed(1))
    assert sorted(idx.tolist()) == list(range(8))
    # a one-rank plan is the identity exchange: everything stays local
    plan = cdist.ShufflePlan(idx, 0, 1)
    assert plan.send_counts == [8] and plan.recv_counts == [8] and plan.bytes_received(4) == 0
    assert torch.equal(plan.send_rows[plan.place], idx)
