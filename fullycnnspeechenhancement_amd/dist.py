"""Batch-axis sharding of the forward pass over the GPUs of one node (one process per GPU).

The reference has no parallelism of any kind (single tf.Session on one device: SURVEY F9); what
shards naturally here is the utterance (batch) axis of `sess.run(pred, {input_x: x})`
(model_utils/tester.py:85-90; layout data_utils/data_loader.py:198-209): inference BatchNorm uses
moving statistics, so utterances are independent and NO collective is needed on the data path when
each rank already owns its shard (`forward_resident`, what bench.py's `value` times).  When one rank
(the reference's single host process) holds the whole batch, `forward_from_root` scatters contiguous
batch slices and gathers the masks back with point-to-point transfers (ncclSend/ncclRecv under
torch.distributed's "nccl" backend = RCCL; one peer per xGMI link, so the root's 7 links work in
parallel).

Overlap on RCCL.  Every peer's slice is cut into `chunks` pieces and the pipeline is
    scatter chunk c+1   ||   compute chunk c   ||   gather chunk c-1.
Two things make that real on RCCL and not only on gloo:
  * ONE `batch_isend_irecv` PER CHUNK.  Under NCCL/RCCL a batch is one coalesced group with one
    work handle that completes when its LAST transfer does, so a single batch holding every chunk
    (round 1) could not release chunk 0 before chunk k had arrived.
  * The two directions run on TWO process groups (= two communicators, each with its own stream):
    operations of one communicator execute in issue order, so with a single one the send of chunk
    c+1 would queue behind the receive of result c, which waits for the peer's compute -- a
    serial scatter -> compute -> gather per chunk.
The same code runs on the "gloo" backend with CPU tensors, which is how tests/ cover the
world_size > 1 logic without GPUs.

CUs for the transfers.  The fused forward kernels are persistent grids of one workgroup per CU that take almost all of a
CU's LDS, and RCCL's send / recv run as kernels that need CU slots of their own: with every CU taken they can only start
when a persistent workgroup exits, and the "scatter c+1 || compute c || gather c-1" pipeline serialises.  `reserved_cus(model,
r)` runs the forwards inside it on `num_cus - r` workgroups (the library's per-handle option "fused_grid"; the kernels'
tile ranges are balanced over whatever grid they are launched with, and results do not depend on it), leaving r CUs to the
communicators; bench.py's from_root leg sweeps r.  Fallback if no r hides the transfers: copies over IPC handles on the
SDMA engines instead of RCCL kernels (not built: no multi-GPU box in the build pool to measure it on).

Failures.  A forward that raises on one rank must not leave the others waiting for transfers that never come: the
failing rank finishes the protocol with zero-filled results, every call ends with one status all-reduce on `group`, and
then EVERY rank raises (the failing one its own exception, the others a RuntimeError naming the rank).  `timeout_s`
bounds each wait on a transfer where the backend supports it (gloo; on RCCL a wait only orders the stream, and the
process group's own timeout / watchdog applies).
"""

import datetime

import torch
import torch.distributed as dist


class reserved_cus(object):
    """with reserved_cus(model, r): ...  -- forwards of `model` (a fullycnnspeechenhancement_amd model) inside the block
    run on num_cus - r workgroups, so that r CUs stay free for the communicators' kernels; r = 0 restores the default."""

    def __init__(self, model, r):
        self.model, self.r = model, int(r)

    def __enter__(self):
        self.prev = self.model.get_option("fused_grid")
        cus = self.model.get_option("num_cus")
        self.model.set_option("fused_grid", max(1, cus - self.r) if self.r > 0 else 0)
        return self

    def __exit__(self, *exc):
        self.model.set_option("fused_grid", self.prev)
        return False


def shard_bounds(n, world):
    """Contiguous, balanced slices of range(n): the first n % world ranks get one extra utterance."""
    base, extra = divmod(int(n), int(world))
    bounds, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def chunk_bounds(lo, hi, chunks):
    """Split [lo, hi) into at most `chunks` contiguous non-empty pieces."""
    n = hi - lo
    if n <= 0:
        return []
    c = max(1, min(int(chunks), n))
    return [(lo + a, lo + b) for a, b in shard_bounds(n, c) if b > a]


class PeerForwardError(RuntimeError):
    """forward_from_root: the forward of another rank raised; this rank's result is incomplete."""


class BatchShardedForward(object):
    """forward: callable mapping a [n, T, 129, 1] tensor on this rank's device to the same shape
    (a fullycnnspeechenhancement_amd model on GPU; any stand-in under gloo in tests).

    Construction is COLLECTIVE OVER THE DEFAULT PROCESS GROUP unless `scatter_group` / `gather_group` are handed in:
    it creates the two direction groups with torch.distributed.new_group, which every process of the default group
    must enter -- also those that are not members of `group`.  With a real subgroup either construct the object on all
    processes (non-members pass `group_ranks=[...]`, the members' global ranks -- a non-member cannot ask the group for
    them -- and never call the forward methods), or create the two groups yourself (collectively) and pass them.  `close()` destroys the groups this object created (their communicators hold device
    memory on RCCL).  `trace`, if given, is a list that receives ("recv"|"fwd"|"send"|"result", chunk) events in the
    order this rank passed them (tests use it to check the pipeline order)."""

    def __init__(self, forward, group=None, device=None, trace=None, forward_into=None, scatter_group=None,
                 gather_group=None, timeout_s=None, group_ranks=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (one process per GPU)")
        if (scatter_group is None) != (gather_group is None):
            raise ValueError("pass both scatter_group and gather_group, or neither")
        self.forward = forward
        self.forward_into = forward_into     # optional (x, out) -> None: writes the result in place (no copy on the root)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.trace = trace
        self.timeout = datetime.timedelta(seconds=timeout_s) if timeout_s else None
        member = self.rank >= 0
        if group_ranks is not None:
            ranks = [int(r) for r in group_ranks]
        elif group is None:
            ranks = list(range(dist.get_world_size()))
        elif member:
            ranks = dist.get_process_group_ranks(group)
        else:
            raise ValueError("this process is not a member of `group`: pass group_ranks (the members' global ranks) so that "
                             "it can enter the two new_group calls, or create scatter_group / gather_group yourself")
        self._ranks = ranks
        self._own_groups = []
        # scatter (root -> peers) and gather (peers -> root) each get a communicator of their own.  new_group is collective
        # over the DEFAULT group: members and non-members alike make both calls (a process that skipped them would be two
        # group-name counters behind, and its next collective new_group would pair with the wrong one).
        if scatter_group is not None:
            self.scatter_group, self.gather_group = scatter_group, gather_group
        elif dist.get_world_size() > 1 and len(ranks) > 1:
            self.scatter_group = dist.new_group(ranks=ranks)
            self.gather_group = dist.new_group(ranks=ranks)
            self._own_groups = [self.scatter_group, self.gather_group]
        else:
            self.scatter_group = self.gather_group = group
        if self.world > 1 and member:
            # The FIRST call on a group's communicator must involve all its ranks (torch.distributed.batch_isend_irecv:
            # otherwise "the behavior is undefined" on NCCL/RCCL), and forward_from_root's transfers only ever pair the
            # root with one peer: open both communicators with a collective here.
            for g in (self.scatter_group, self.gather_group):
                dist.all_reduce(torch.zeros(1, device=self.device), group=g)

    def close(self):
        """Destroy the two direction groups this object created (collective over their ranks on some backends: call it
        on every rank).  Groups handed in by the caller are the caller's."""
        groups, self._own_groups = self._own_groups, []
        for g in groups:
            try:
                dist.destroy_process_group(g)
            except Exception:
                pass

    def _note(self, what, c):
        if self.trace is not None:
            self.trace.append((what, c))

    def _wait(self, works):
        for w in works:
            if self.timeout is not None:
                w.wait(timeout=self.timeout)
            else:
                w.wait()

    def _finish(self, err):
        """End of a forward_from_root call on every rank: one status all-reduce (MAX over 1 + failing rank), then raise
        where something failed."""
        if self.world > 1:
            flag = torch.tensor([0 if err is None else self.rank + 1], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            bad = int(flag.item())
        else:
            bad = 0 if err is None else self.rank + 1
        if err is not None:
            raise err
        if bad:
            raise PeerForwardError("forward_from_root: the forward of rank %d raised; the gathered result is incomplete"
                                   % (bad - 1))

    # -- every rank already holds its own utterances: no data-path collective -----------------
    def forward_resident(self, x_local):
        return self.forward(x_local)

    # -- one rank holds the whole batch (the reference's calling convention) -------------------
    def forward_from_root(self, x_root, root=0, chunks=1, direction="both"):
        """x_root: [N, T, 129, 1] on `root` (ignored elsewhere).  Returns [N, T, 129, 1] on root, None
        on the other ranks.  `root` is a rank of `group`.  chunks > 1 pipelines each peer's slice.
        direction (measurement only): "scatter" = the peers receive their slices and nothing else happens, "gather" =
        the peers send zero-filled results of the right shape and nothing else happens -- the compute-free transfer
        times bench.py reports next to the pipelined figure; the returned tensor is then meaningless."""
        if direction not in ("both", "scatter", "gather"):
            raise ValueError("direction must be 'both', 'scatter' or 'gather'")
        # what the root broadcasts is (shape, dtype) or the text of its own validation error: the peers are already waiting
        # in the broadcast when the root looks at its input, and must fail with it rather than hang
        meta, bad_input = [None], None
        if self.rank == root:
            if not torch.is_tensor(x_root) or x_root.dim() != 4 or x_root.shape[2] != 129 or x_root.shape[3] != 1:
                bad_input = "input must be a tensor [N, T, 129, 1], got %s" % (
                    tuple(x_root.shape) if torch.is_tensor(x_root) else type(x_root).__name__,)
                meta = [("error", bad_input)]
            else:
                meta = [(tuple(x_root.shape), str(x_root.dtype))]
        if self.world > 1:
            dist.broadcast_object_list(meta, src=self._ranks[root], group=self.group)
        if meta[0][0] == "error":
            if self.rank == root:
                raise ValueError(bad_input)
            raise PeerForwardError("forward_from_root: the root refused its input (%s)" % meta[0][1])
        shape, dtype_name = meta[0]
        dtype = getattr(torch, dtype_name.split(".")[-1])
        n, t = shape[0], shape[1]
        bounds = shard_bounds(n, self.world)
        groot = self._ranks[root]            # P2POp peers are global ranks
        do_scatter, do_gather, do_compute = direction != "gather", direction != "scatter", direction == "both"
        err = None

        if self.rank == root:
            x_root = x_root.contiguous()
            y = torch.empty_like(x_root)
            pieces = {r: chunk_bounds(*bounds[r], chunks) for r in range(self.world) if r != root}
            depth = max([len(p) for p in pieces.values()] + [0])
            send_works, recv_works = [], []
            for c in range(depth):       # chunk c of every peer: one grouped batch per direction
                sends = [dist.P2POp(dist.isend, x_root[p[c][0]:p[c][1]], self._ranks[r], self.scatter_group)
                         for r, p in pieces.items() if c < len(p)]
                recvs = [dist.P2POp(dist.irecv, y[p[c][0]:p[c][1]], self._ranks[r], self.gather_group)
                         for r, p in pieces.items() if c < len(p)]
                send_works.append(dist.batch_isend_irecv(sends) if do_scatter else [])
                recv_works.append(dist.batch_isend_irecv(recvs) if do_gather else [])
            lo, hi = bounds[root]
            if hi > lo and do_compute:   # the root's own slice computes while its links carry the others'
                try:
                    for i, (a, b) in enumerate(chunk_bounds(lo, hi, chunks)):
                        if self.forward_into is not None:
                            self.forward_into(x_root[a:b], y[a:b])
                        else:
                            y[a:b] = self.forward(x_root[a:b])
                        self._note("fwd", i)
                except Exception as e:   # the transfers in flight are still drained: the peers must not be left waiting
                    err = e
            try:
                for c in range(depth):   # results stream back chunk by chunk
                    self._wait(send_works[c] + recv_works[c])
                    self._note("result", c)
            except Exception as e:       # a transfer that timed out (gloo): the status all-reduce below still runs, so that
                err = err or e           # the other ranks leave the call with an error instead of waiting in it
            self._finish(err)
            return y

        lo, hi = bounds[self.rank]
        pieces = chunk_bounds(lo, hi, chunks)
        if not pieces:
            self._finish(None)
            return None
        alloc = torch.empty if do_scatter else torch.zeros
        bufs = [alloc((b - a, t) + tuple(shape[2:]), dtype=dtype, device=self.device) for a, b in pieces]
        # every receive is posted up front, each as a batch of its own: chunk c+1 lands while chunk c computes
        recv_works = [dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, groot, self.scatter_group)]) if do_scatter else []
                      for buf in bufs]
        send_works, outs = [], []
        for c, buf in enumerate(bufs):
            try:
                self._wait(recv_works[c])    # nccl: the current stream waits for THIS chunk only
            except Exception as e:       # a receive that timed out: go on with zeros so that the protocol still ends
                err = err or e
                buf.zero_()
            self._note("recv", c)
            out = buf
            if do_compute:
                if err is None:
                    try:
                        out = self.forward(buf).contiguous()
                        if out.shape != buf.shape or out.dtype != buf.dtype:
                            raise ValueError("forward returned %s %s for an input of %s %s"
                                             % (tuple(out.shape), out.dtype, tuple(buf.shape), buf.dtype))
                    except Exception as e:
                        err = e
                if err is not None:      # finish the protocol with zeros: the root is waiting for this chunk
                    out = torch.zeros_like(buf)
                self._note("fwd", c)
            if do_gather:
                outs.append(out)         # keep alive until sent
                send_works += dist.batch_isend_irecv([dist.P2POp(dist.isend, out, groot, self.gather_group)])
                self._note("send", c)
        try:
            self._wait(send_works)
        except Exception as e:
            err = err or e
        self._finish(err)
        return None
