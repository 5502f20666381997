"""world_size-2 tests of the batch-shard scatter/gather logic on the gloo backend (CPU tensors).

The product forward has no CPU path, so the stand-in forward here is the oracle (tests may use it);
what is under test is fullycnnspeechenhancement_amd/dist.py: slice bounds, chunked pipelining, ragged
and empty shards, and that the gathered result equals the single-process result bit for bit."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from oracle import rced_c, rced_np


def test_shard_bounds_cover_and_balance():
    from fullycnnspeechenhancement_amd.dist import chunk_bounds, shard_bounds
    for n in (0, 1, 2, 5, 8, 256, 2048, 2049):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(2048, 8) == [(256 * i, 256 * (i + 1)) for i in range(8)]   # BASELINE config 4
    assert chunk_bounds(10, 10, 4) == []
    assert chunk_bounds(0, 3, 8) == [(0, 1), (1, 2), (2, 3)]
    assert chunk_bounds(4, 12, 2) == [(4, 8), (8, 12)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, t, chunks, root, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fullycnnspeechenhancement_amd.dist import BatchShardedForward
        w = rced_np.make_weights("FullyCNNV3", seed=42)
        calls = []

        def forward(x):   # stand-in for the GPU model: the oracle's fp32 port
            calls.append(int(x.shape[0]))
            return torch.from_numpy(rced_c.forward("FullyCNNV3", w, x.numpy(), np.float32))

        eng = BatchShardedForward(forward, device="cpu")
        x = torch.from_numpy(rced_np.make_input(n, t, seed=77)) if rank == root else None
        y = eng.forward_from_root(x, root=root, chunks=chunks)
        if rank == root:
            np.save(out_path, y.numpy())
        else:
            assert y is None
        # the resident path is a plain local call
        xl = torch.from_numpy(rced_np.make_input(1, t, seed=5 + rank))
        assert torch.equal(eng.forward_resident(xl), forward(xl))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,t,chunks,root", [(4, 9, 1, 0), (5, 8, 2, 0), (1, 8, 1, 0), (3, 12, 4, 1)])
def test_scatter_forward_gather_world2(tmp_path, built, n, t, chunks, root):
    out = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(2, _free_port(), n, t, chunks, root, out), nprocs=2, join=True)
    y = np.load(out)
    w = rced_np.make_weights("FullyCNNV3", seed=42)
    x = rced_np.make_input(n, t, seed=77)
    ref = rced_c.forward("FullyCNNV3", w, x, np.float32)
    assert y.shape == ref.shape
    assert np.array_equal(y, ref)     # utterances are independent: sharding changes nothing


def _pipeline_worker(rank, world, port, out_path):
    """4 utterances per peer in 4 chunks; the stand-in forward sleeps, so the pipeline order is observable."""
    import json
    import sys
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fullycnnspeechenhancement_amd import dist as D
        batches = []                         # every batch_isend_irecv this rank issues: [(op, first utterance row)]
        real = dist.batch_isend_irecv

        def spy(ops):
            batches.append([(o.op.__name__, int(o.tensor.shape[0]), id(o.group)) for o in ops])
            return real(ops)

        D.dist.batch_isend_irecv = spy
        trace, stamps = [], []

        def forward(x):
            time.sleep(0.15)
            stamps.append(time.perf_counter())
            return x * 2.0

        eng = D.BatchShardedForward(forward, device="cpu", trace=trace)
        n, t, chunks = 8, 3, 4
        x = torch.arange(n * t * 129, dtype=torch.float32).reshape(n, t, 129, 1) if rank == 0 else None
        t0 = time.perf_counter()
        y = eng.forward_from_root(x, root=0, chunks=chunks)
        if rank == 0:
            assert torch.equal(y, x * 2.0)
        json.dump({"trace": trace, "batches": batches, "elapsed": time.perf_counter() - t0,
                   "groups": [id(eng.scatter_group), id(eng.gather_group)]}, open("%s.%d" % (out_path, rank), "w"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_from_root_issues_one_batch_per_chunk_and_pipelines(tmp_path):
    """dist.py: ONE batch_isend_irecv per chunk and per direction (on RCCL a batch completes as a whole, so chunks in
    one batch cannot overlap compute), scatter and gather on different groups (communicators; one communicator runs
    its operations in issue order), every receive posted before the first compute, results sent back chunk by chunk,
    and the root's own slice computed while the peers compute theirs."""
    import json
    out = str(tmp_path / "t")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    root, peer = (json.load(open("%s.%d" % (out, r))) for r in (0, 1))
    # peer: 4 receive batches of one op each (posted up front), then one send batch per chunk
    assert [b[0][0] for b in peer["batches"]] == ["irecv"] * 4 + ["isend"] * 4
    assert all(len(b) == 1 and b[0][1] == 1 for b in peer["batches"])
    assert {b[0][2] for b in peer["batches"][:4]} == {peer["groups"][0]}        # scatter group
    assert {b[0][2] for b in peer["batches"][4:]} == {peer["groups"][1]}        # gather group
    assert peer["groups"][0] != peer["groups"][1]
    # root: per chunk one send batch and one receive batch, alternating
    assert [b[0][0] for b in root["batches"]] == ["isend", "irecv"] * 4
    # pipeline order on the peer: wait for chunk c only, compute it, send it back at once
    kinds = [k for k, _ in peer["trace"]]
    assert kinds == ["recv", "fwd", "send"] * 4
    assert [c for k, c in peer["trace"] if k == "send"] == [0, 1, 2, 3]
    assert [c for k, c in root["trace"] if k == "result"] == [0, 1, 2, 3]
    # the root computed its own 4 chunks while the peer computed its 4: ~4 x 0.15 s, not 8 x
    assert root["elapsed"] < 0.15 * 4 + 0.45, root["elapsed"]


def _world4_worker(rank, world, port, n, t, chunks, root, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fullycnnspeechenhancement_amd.dist import BatchShardedForward, shard_bounds
        sizes = []

        def forward(x):      # cheap stand-in with a per-element, per-utterance answer: any misrouted row shows
            sizes.append(int(x.shape[0]))
            return x * 3.0 + 1.0

        eng = BatchShardedForward(forward, device="cpu", timeout_s=60)
        x = torch.arange(n * t * 129, dtype=torch.float32).reshape(n, t, 129, 1) if rank == root else None
        for _ in range(2):   # the object is reusable: the second call runs on the already opened communicators
            y = eng.forward_from_root(x, root=root, chunks=chunks)
        lo, hi = shard_bounds(n, world)[rank]
        assert sum(sizes) == 2 * (hi - lo), (rank, sizes)          # every rank computed exactly its own slice, twice
        assert len(sizes) == 2 * min(chunks, hi - lo)              # at most `chunks` pieces, never an empty one
        if rank == root:
            assert torch.equal(y, x * 3.0 + 1.0)
            open(out_path, "w").write("ok")
        else:
            assert y is None
        # compute-free probes (bench.py's transfer-only figures) leave the engine usable
        eng.forward_from_root(x, root=root, chunks=chunks, direction="scatter")
        eng.forward_from_root(x, root=root, chunks=chunks, direction="gather")
        y = eng.forward_from_root(x, root=root, chunks=chunks)
        if rank == root:
            assert torch.equal(y, x * 3.0 + 1.0)
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,chunks,root", [(7, 8, 0), (3, 2, 2), (10, 3, 1)])
def test_from_root_world4_uneven_shards_and_more_chunks_than_utterances(tmp_path, n, chunks, root):
    """Config 4's shape in small: 4 ranks, shards of unequal size (7 -> 2,2,2,1; 3 -> 1,1,1,0: one rank gets nothing),
    more chunks asked for than a peer has utterances, a root that is not rank 0."""
    out = str(tmp_path / "ok")
    mp.spawn(_world4_worker, args=(4, _free_port(), n, 5, chunks, root, out), nprocs=4, join=True)
    assert open(out).read() == "ok"


def _failing_worker(rank, world, port, bad_rank, out_path):
    import sys
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fullycnnspeechenhancement_amd.dist import BatchShardedForward, PeerForwardError
        calls = []

        def forward(x):
            calls.append(1)
            if rank == bad_rank and len(calls) == 2:      # the second chunk of that rank fails
                raise FloatingPointError("boom on rank %d" % rank)
            return x + 1.0

        eng = BatchShardedForward(forward, device="cpu", timeout_s=60)
        x = torch.ones(12, 4, 129, 1) if rank == 0 else None
        t0 = time.perf_counter()
        what = "returned"
        try:
            eng.forward_from_root(x, root=0, chunks=3)
        except FloatingPointError:
            what = "own"
        except PeerForwardError as e:
            what = "peer:%s" % e
        open("%s.%d" % (out_path, rank), "w").write("%s|%.1f" % (what, time.perf_counter() - t0))
        # ... and the communicators are still in step: the next call works
        y = eng.forward_from_root(x, root=0, chunks=1)
        if rank == 0:
            assert torch.equal(y, x + 1.0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank", [2, 0])
def test_from_root_fails_on_every_rank_when_one_forward_raises(tmp_path, bad_rank):
    """A forward that raises on a peer (or on the root) must not leave anybody waiting: the failing rank re-raises its
    exception, every other rank raises PeerForwardError naming it, within seconds, and the engine stays usable."""
    out = str(tmp_path / "r")
    mp.spawn(_failing_worker, args=(3, _free_port(), bad_rank, out), nprocs=3, join=True)
    for r in range(3):
        what, el = open("%s.%d" % (out, r)).read().split("|")
        assert float(el) < 30.0
        if r == bad_rank:
            assert what == "own"
        else:
            assert what.startswith("peer:") and "rank %d" % bad_rank in what, (r, what)


def _bad_input_worker(rank, world, port, out_path):
    import sys
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fullycnnspeechenhancement_amd.dist import BatchShardedForward, PeerForwardError
        eng = BatchShardedForward(lambda a: a + 1.0, device="cpu")
        t0 = time.perf_counter()
        what = "none"
        try:
            eng.forward_from_root(torch.zeros(4, 9, 128, 1) if rank == 0 else None, root=0, chunks=2)   # 128 bins: refused
        except ValueError as e:
            what = "own:%s" % e
        except PeerForwardError as e:
            what = "peer:%s" % e
        open("%s.%d" % (out_path, rank), "w").write("%s|%.1f" % (what, time.perf_counter() - t0))
        x = torch.ones(4, 9, 129, 1)
        y = eng.forward_from_root(x if rank == 0 else None, root=0, chunks=2)    # the engine is still usable
        if rank == 0:
            assert torch.equal(y, x + 1.0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_from_root_input_refused_by_the_root_fails_every_rank(tmp_path):
    """The root validates its input while the peers already wait in the meta broadcast: what is broadcast is the shape OR the
    refusal, so that the peers raise too (PeerForwardError) instead of waiting for a shape that never comes."""
    out = str(tmp_path / "r")
    mp.spawn(_bad_input_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    for r in range(3):
        what, el = open("%s.%d" % (out, r)).read().split("|")
        assert float(el) < 30.0
        assert what.startswith("own:" if r == 0 else "peer:") and "129" in what, (r, what)


def _subgroup_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fullycnnspeechenhancement_amd.dist import BatchShardedForward
        members = [1, 2]
        sub = dist.new_group(ranks=members)
        # every process of the default group constructs the object; the non-member names the members' ranks
        eng = BatchShardedForward(lambda a: a * 2.0, group=sub, device="cpu", group_ranks=members)
        if rank in members:
            x = torch.arange(5 * 9 * 129, dtype=torch.float32).reshape(5, 9, 129, 1) if rank == 1 else None
            y = eng.forward_from_root(x, root=0, chunks=2)           # root = rank 0 OF THE SUBGROUP = global rank 1
            if rank == 1:
                assert torch.equal(y, x * 2.0)
        # the group-name counters of all three processes are still in step: a later collective new_group pairs up
        late = dist.new_group(ranks=[0, 1, 2])
        t = torch.tensor([float(rank)])
        dist.all_reduce(t, group=late)
        assert float(t) == 3.0
        open("%s.%d" % (out_path, rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_from_root_on_a_subgroup_with_a_non_member_process(tmp_path):
    """A real subgroup (ranks 1, 2 of 3): the non-member constructs the object too (group_ranks), entering both new_group
    calls, so a LATER collective new_group over all three still pairs up."""
    out = str(tmp_path / "r")
    mp.spawn(_subgroup_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert all(open("%s.%d" % (out, r)).read() == "ok" for r in range(3))


def _copy_worker(rank, world, port, n, t, chunks, root, out_path):
    """transport="copy" (peers pull / push through handles to the root's tensors) against transport="rccl" on the same input."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fullycnnspeechenhancement_amd.dist import BatchShardedForward, PeerForwardError, shard_bounds
        sizes, trace = [], []

        def forward(x):
            sizes.append(int(x.shape[0]))
            return torch.sin(x) * 3.0 + 1.0

        ref_eng = BatchShardedForward(forward, device="cpu", timeout_s=60)
        eng = BatchShardedForward(forward, device="cpu", transport="copy", trace=trace)
        x = (torch.arange(n * t * 129, dtype=torch.float32).reshape(n, t, 129, 1) * 1e-3) if rank == root else None
        y_ref = ref_eng.forward_from_root(x, root=root, chunks=chunks)
        del sizes[:]
        ys = []
        for _ in range(3):        # reusable; two output buffers per shape, used alternately: call k's result lives until call k + 2
            y = eng.forward_from_root(x, root=root, chunks=chunks)
            ys.append(y)
        if rank == root:
            assert ys[0] is not ys[1] and ys[2].data_ptr() == ys[0].data_ptr() and torch.equal(ys[1], ys[2])
        lo, hi = shard_bounds(n, world)[rank]
        assert sum(sizes) == 3 * (hi - lo), (rank, sizes)
        if rank == root:
            assert torch.equal(y, y_ref) and torch.equal(y, torch.sin(x) * 3.0 + 1.0)     # bit-equal with the send / recv transport
            open(out_path, "w").write("ok")
        else:
            assert y is None
            if hi > lo:           # pull, compute, push -- chunk by chunk
                k = min(chunks, hi - lo)
                assert [w for w, _ in trace[:3 * k]] == ["recv", "fwd", "send"] * k
        # the compute-free probes and a root that refuses its input leave the engine usable
        eng.forward_from_root(x, root=root, chunks=chunks, direction="scatter")
        eng.forward_from_root(x, root=root, chunks=chunks, direction="gather")
        with pytest.raises((ValueError, PeerForwardError)):
            eng.forward_from_root(torch.zeros(2, 3) if rank == root else None, root=root)
        y = eng.forward_from_root(x, root=root, chunks=chunks)
        if rank == root:
            assert torch.equal(y, y_ref)
        dist.barrier()
        eng.close()
        ref_eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,chunks,root", [(2, 5, 2, 0), (4, 7, 8, 0), (3, 3, 2, 2), (4, 10, 3, 1)])
def test_copy_transport_equals_send_recv(tmp_path, world, n, chunks, root):
    """dist.py's plan B -- no communicator kernels on the data path: the root shares handles to its tensors, the peers copy their
    slices out of its input and their results into its output -- gives the same bits as the send / recv form; uneven and empty
    shards, more chunks than utterances, a root that is not rank 0.  (gloo + CPU tensors: the handles are shared-memory files; on a
    GPU they are CUDA IPC memory handles and the copies hipMemcpyAsync between peers -- tests/test_forward_gpu.py has that leg.)"""
    out = str(tmp_path / "ok")
    mp.spawn(_copy_worker, args=(world, _free_port(), n, 4, chunks, root, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def _poison_worker(rank, world, port, out_path):
    """A transfer that times out ends the call on every rank with an error AND poisons the object ON EVERY RANK -- also on a
    peer whose own transfers all completed (world 3: rank 2): its sends / receives may still be pending on the direction groups,
    so later calls are refused everywhere (a rank that went on would wait in the next call's broadcast for a root that refuses
    to enter it) until the object is closed and rebuilt."""
    import sys
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fullycnnspeechenhancement_amd.dist import BatchShardedForward, TransferTimeout

        def forward(x):
            if rank == 1:
                time.sleep(3.0)       # the root's wait for this peer's result (timeout 1 s) gives up first
            return x + 1.0

        eng = BatchShardedForward(forward, device="cpu", timeout_s=1.0)
        x = torch.ones(6, 2, 129, 1) if rank == 0 else None
        failed = False
        try:
            eng.forward_from_root(x, root=0, chunks=1)
        except Exception:
            failed = True
        assert failed                                         # every rank leaves the call with an exception
        assert eng._poisoned is not None, rank                # ... and knows of the time-out, whoever saw it
        if rank == 0:
            assert eng._pending
        with pytest.raises(TransferTimeout):                  # EVERY rank makes the second call: refused without touching the groups
            eng.forward_from_root(x, root=0, chunks=1)
        open(out_path + str(rank), "w").write("ok")
        time.sleep(3.5)                                       # let the late message drain before the groups go away
        eng.close()
        fresh = BatchShardedForward(lambda v: v + 1.0, device="cpu", timeout_s=30)
        y = fresh.forward_from_root(x, root=0, chunks=1)      # a new object (new direction groups) works
        if rank == 0:
            assert torch.equal(y, x + 1.0)
        fresh.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_transfer_timeout_poisons_the_engine(tmp_path, world):
    out = str(tmp_path / "ok")
    mp.spawn(_poison_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(open(out + str(r)).read() == "ok" for r in range(world))
