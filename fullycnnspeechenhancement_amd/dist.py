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
communicators; bench.py's from_root leg sweeps r.

Plan B: transport="copy".  No communicator kernels at all on the data path: the root hands the peers handles to its input
and output tensors (CUDA IPC memory handles, pickled by torch.multiprocessing's reductions and broadcast over the control
group; under gloo with CPU tensors: shared-memory files, the same code) and every peer PULLS its chunks out of the root's
input and PUSHES its results into the root's output with plain device-to-device copies -- hipMemcpyAsync between peers, which
the runtime puts on the SDMA engines -- on two side streams, pipelined pull c+1 || compute c || push c-1 like the RCCL form.
The copies need no CU, so the persistent forward grids keep every CU (no `reserved_cus`).  A handle is made once per storage on the
root and opened once per peer (both sides cache it until close()): a caller that reuses its input tensor pays neither again; the
root's output is one of two buffers kept per shape, used alternately.
bench.py's from_root leg times both transports; tests hold them bit-equal (gloo world 2..4 on CPU; two processes sharing one GPU
through IPC handles on the GPU box).

Failures.  A forward that raises on one rank must not leave the others waiting for transfers that never come: the
failing rank finishes the protocol with zero-filled results, every call ends with one status all-reduce on `group`, and
then EVERY rank raises (the failing one its own exception, the others a RuntimeError naming the rank), and the object stays
usable.  `timeout_s` bounds each wait on a transfer where the backend supports it (gloo; on RCCL a wait only orders the
stream, and the process group's own timeout / watchdog applies).  A transfer TIME-OUT is different: the call still ends on
every rank with an exception, but transfers may be left pending on the direction groups (gloo closes the pair), so the object
is poisoned -- later forward_from_root calls raise TransferTimeout until it is closed and built again.
"""

import datetime
import io
import pickle

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


class TransferTimeout(RuntimeError):
    """A transfer of forward_from_root timed out (gloo).  The call ended on every rank, but its sends / receives may still be
    pending on the two direction groups: a late message could land in a LATER call's buffers, so the object refuses further
    forward_from_root calls until it is closed and built again (new direction groups)."""


def _share(t):
    """A DEVICE tensor -> bytes another process of this node can turn back into a tensor over THE SAME memory
    (torch.multiprocessing.reductions: a CUDA IPC memory handle)."""
    from multiprocessing.reduction import ForkingPickler
    buf = io.BytesIO()
    ForkingPickler(buf, pickle.HIGHEST_PROTOCOL).dump(t)
    return buf.getvalue()


def _open_shared(blob):
    return pickle.loads(blob)


class _ShmTensor(object):
    """CPU stand-in for a device buffer other processes can map (the gloo tests of transport="copy"): a tensor over a named
    POSIX shared-memory block; `handle` is what a peer needs to map it."""

    def __init__(self, shape=None, dtype=None, handle=None):
        from multiprocessing import shared_memory
        if handle is None:
            nbytes = max(int(torch.empty((), dtype=dtype).element_size()) * int(torch.Size(shape).numel()), 1)
            self.shm, self.owner = shared_memory.SharedMemory(create=True, size=nbytes), True
            self.handle = ("shm", self.shm.name, tuple(shape), str(dtype))
        else:
            _, name, shape, dtype_name = handle
            dtype = getattr(torch, dtype_name.split(".")[-1])
            self.shm, self.owner, self.handle = shared_memory.SharedMemory(name=name), False, handle
        n = int(torch.Size(shape).numel())
        self.tensor = torch.frombuffer(self.shm.buf, dtype=dtype, count=n).reshape(shape) if n else torch.empty(shape, dtype=dtype)

    def close(self):
        self.tensor = None
        try:
            self.shm.close()         # raises BufferError while a caller still holds a view of the block ...
        except Exception:
            pass
        finally:
            if self.owner:           # ... the name goes away all the same: the memory is freed with its last mapping
                try:
                    self.shm.unlink()
                except Exception:
                    pass


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
                 gather_group=None, timeout_s=None, group_ranks=None, transport="rccl"):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (one process per GPU)")
        if transport not in ("rccl", "copy"):
            raise ValueError("transport must be 'rccl' (send / recv on the two direction groups) or 'copy' (peers pull / push "
                             "through handles to the root's tensors)")
        self.transport = transport
        self._poisoned = None        # set by a transfer time-out: see TransferTimeout
        self._pending = []           # works / buffers of a timed-out call, kept alive until close()
        self._out_cache = {}         # transport "copy": the root's output buffer (CPU: also its shared input copy) per (shape, dtype)
        self._opened = {}            # transport "copy": what this peer has mapped of the root's memory -- CPU: shared-memory blocks by
                                     # name; CUDA: tensors over opened IPC handles, by the handle's bytes -- kept until close()
        self._shared = {}            # transport "copy" on CUDA, root: IPC handle blobs already made, per storage (_shared_blob)
        self._streams = None         # transport "copy" on CUDA: (pull stream, push stream)
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
        # control-plane tensors (the status all-reduce) live where the control group's backend can move them
        try:
            self._ctl_device = self.device if (self.device.type != "cuda" or "nccl" in str(dist.get_backend(group if member else None))) \
                else torch.device("cpu")
        except Exception:
            self._ctl_device = self.device
        if scatter_group is not None:
            self.scatter_group, self.gather_group = scatter_group, gather_group
        elif transport == "copy":
            self.scatter_group = self.gather_group = group     # no data-path communicators at all (and no collective new_group)
        elif dist.get_world_size() > 1 and len(ranks) > 1:
            self.scatter_group = dist.new_group(ranks=ranks)
            self.gather_group = dist.new_group(ranks=ranks)
            self._own_groups = [self.scatter_group, self.gather_group]
        else:
            self.scatter_group = self.gather_group = group
        if self.world > 1 and member and transport != "copy":
            # The FIRST call on a group's communicator must involve all its ranks (torch.distributed.batch_isend_irecv:
            # otherwise "the behavior is undefined" on NCCL/RCCL), and forward_from_root's transfers only ever pair the
            # root with one peer: open both communicators with a collective here.
            for g in (self.scatter_group, self.gather_group):
                dist.all_reduce(torch.zeros(1, device=self.device), group=g)

    def close(self):
        """Destroy the two direction groups this object created (collective over their ranks on some backends: call it
        on every rank).  Groups handed in by the caller are the caller's."""
        groups, self._own_groups = self._own_groups, []
        for v in list(self._out_cache.values()) + list(self._opened.values()):
            for o in (v[0] if isinstance(v, list) else (v,)):
                if isinstance(o, _ShmTensor):
                    o.close()
        self._pending, self._out_cache, self._opened, self._shared = [], {}, {}, {}
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
        """End of a forward_from_root call on every rank: one status all-reduce (MAX over [1 + failing rank, 1 + rank whose
        transfer timed out]), then raise where something failed.  A time-out ANYWHERE poisons the object on EVERY rank: the
        direction groups may hold pending transfers on ranks whose own waits all completed, and a rank that went on would
        sit in the next call's broadcast while the poisoned root refuses to enter it."""
        mine = self._poisoned is not None
        if self.world > 1:
            flag = torch.tensor([0 if err is None else self.rank + 1, self.rank + 1 if mine else 0], dtype=torch.int32,
                                device=self._ctl_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            bad, timed = int(flag[0].item()), int(flag[1].item())
        else:
            bad, timed = (0 if err is None else self.rank + 1), (self.rank + 1 if mine else 0)
        if timed and self._poisoned is None:
            self._poisoned = "a transfer of rank %d timed out" % (timed - 1)
        if err is not None:
            raise err
        if timed:
            raise TransferTimeout("forward_from_root: a transfer of rank %d timed out; the gathered result is incomplete and this "
                                  "object refuses further calls -- close() it and build a new one" % (timed - 1))
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
        transport="rccl" returns a fresh tensor per call.  transport="copy" returns one of TWO output buffers this object keeps
        per shape, alternately (the peers write into memory they have mapped): the result of call k is overwritten by call
        k + 2 -- copy it if it must live longer.  `timeout_s` has no effect with transport="copy" (its waits are stream
        waits on plain copies; a peer that dies is seen by the status all-reduce's own process-group time-out).
        direction (measurement only): "scatter" = the peers receive their slices and nothing else happens, "gather" =
        the peers send zero-filled results of the right shape and nothing else happens -- the compute-free transfer
        times bench.py reports next to the pipelined figure; the returned tensor is then meaningless."""
        if direction not in ("both", "scatter", "gather"):
            raise ValueError("direction must be 'both', 'scatter' or 'gather'")
        if self._poisoned is not None:
            raise TransferTimeout("forward_from_root: an earlier call on this object ended with a transfer time-out (%s); its transfers "
                                  "may still be pending -- close() it and build a new one" % self._poisoned)
        if self.transport == "copy":
            return self._from_root_copy(x_root, root, chunks, direction)
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
                try:
                    send_works.append(dist.batch_isend_irecv(sends) if do_scatter else [])
                    recv_works.append(dist.batch_isend_irecv(recvs) if do_gather else [])
                except Exception as e:   # a direction group that a time-out broke: finish the call (status all-reduce), refuse later ones
                    err = err or e
                    self._poison(e, send_works, recv_works, x_root, y)
                    while len(send_works) <= c:
                        send_works.append([])
                    while len(recv_works) <= c:
                        recv_works.append([])
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
                self._poison(e, send_works, recv_works, x_root, y)
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
                self._poison(e, recv_works, bufs)
                buf = bufs[c] = torch.zeros_like(buf)   # (not the buffer the late message may still land in)
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
                try:
                    send_works += dist.batch_isend_irecv([dist.P2POp(dist.isend, out, groot, self.gather_group)])
                except Exception as e:   # gloo: the root's timed-out wait has closed the pair -- the send fails when it is issued
                    err = err or e
                    self._poison(e, outs)
                self._note("send", c)
        try:
            self._wait(send_works)
        except Exception as e:
            err = err or e
            self._poison(e, send_works, outs)
        self._finish(err)
        return None

    def _poison(self, exc, *keep):
        """A wait on a transfer raised (time-out): remember why, and keep the works and buffers of this call alive -- the
        transfer may still complete into them."""
        if self._poisoned is None:
            self._poisoned = "%s: %s" % (type(exc).__name__, exc)
        self._pending.extend(keep)

    # -- transport "copy": the peers pull / push through handles to the root's tensors ---------------------------
    def _shared_blob(self, t):
        """The IPC handle of a device tensor, made ONCE per storage: every `_share_cuda_()` allocates a reference-counter slot
        and an interprocess event that live as long as the storage does, so a serving loop that shared its tensors per call
        would grow without bound.  The cache holds the tensor too (a handle must not outlive its memory)."""
        st = t.untyped_storage()
        key = (st.data_ptr(), st.nbytes(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype)
        hit = self._shared.get(key)
        if hit is None:
            if len(self._shared) >= 8:      # a caller that hands a fresh input tensor every call: forget the oldest
                self._shared.pop(next(iter(self._shared)))
            hit = self._shared[key] = (_share(t), t)
        return hit[0]

    def _from_root_copy(self, x_root, root, chunks, direction):
        cuda = self.device.type == "cuda"
        meta, bad_input = [None], None
        y = None
        if self.rank == root:
            if not torch.is_tensor(x_root) or x_root.dim() != 4 or x_root.shape[2] != 129 or x_root.shape[3] != 1:
                bad_input = "input must be a tensor [N, T, 129, 1], got %s" % (
                    tuple(x_root.shape) if torch.is_tensor(x_root) else type(x_root).__name__,)
                meta = [("error", bad_input)]
            else:
                try:
                    x_root = x_root.contiguous()
                    key = (tuple(x_root.shape), x_root.dtype)
                    bounds_r = shard_bounds(x_root.shape[0], self.world)
                    peers = any(b > a for r, (a, b) in enumerate(bounds_r) if r != root)   # does any peer hold a shard?
                    if cuda:
                        slot = self._out_cache.get(key)
                        if slot is None:        # two output buffers per shape, used alternately (forward_from_root's docstring)
                            slot = [[torch.empty_like(x_root), torch.empty_like(x_root)], 0]
                            self._out_cache = {key: slot}
                        y = slot[0][slot[1]]
                        slot[1] ^= 1
                        torch.cuda.current_stream().synchronize()    # the input is complete before a peer reads it
                        meta = [(tuple(x_root.shape), str(x_root.dtype), self._shared_blob(x_root) if peers else None,
                                 self._shared_blob(y) if peers else None)]
                    else:                                            # CPU / gloo: named shared-memory blocks; the input is copied in
                        slot = self._out_cache.get(key)
                        if slot is None:
                            for v in self._out_cache.values():
                                for o in v[0]:
                                    o.close()
                            slot = [[_ShmTensor(x_root.shape, x_root.dtype), _ShmTensor(x_root.shape, x_root.dtype),
                                     _ShmTensor(x_root.shape, x_root.dtype)], 0]    # the input's copy and two outputs
                            self._out_cache = {key: slot}
                        xin, yout = slot[0][0], slot[0][1 + slot[1]]
                        slot[1] ^= 1
                        xin.tensor.copy_(x_root)
                        x_root, y = xin.tensor, yout.tensor
                        meta = [(tuple(x_root.shape), str(x_root.dtype), xin.handle, yout.handle)]
                except Exception as e:
                    bad_input = "cannot share the root's tensors: %s: %s" % (type(e).__name__, e)
                    meta = [("error", bad_input)]
        if self.world > 1:
            dist.broadcast_object_list(meta, src=self._ranks[root], group=self.group)
        if meta[0][0] == "error":
            if self.rank == root:
                raise ValueError(bad_input)
            raise PeerForwardError("forward_from_root: the root refused its input (%s)" % meta[0][1])
        shape, dtype_name, xb, yb = meta[0]
        n = shape[0]
        bounds = shard_bounds(n, self.world)
        do_scatter, do_gather, do_compute = direction != "gather", direction != "scatter", direction == "both"
        err = None
        lo, hi = bounds[self.rank]
        pieces = chunk_bounds(lo, hi, chunks)
        if self.rank == root:
            if do_compute:
                try:
                    for i, (a, b) in enumerate(pieces):
                        if self.forward_into is not None:
                            self.forward_into(x_root[a:b], y[a:b])
                        else:
                            y[a:b] = self.forward(x_root[a:b])
                        self._note("fwd", i)
                except Exception as e:
                    err = e
            self._finish(err)          # the peers join it when their last push has completed
            if cuda:
                torch.cuda.current_stream().synchronize()
            return y
        if not pieces:
            self._finish(None)
            return None
        # Everything this peer does between here and the status all-reduce is inside ONE try: an allocation that fails, a
        # mapping or a peer copy that raises, must still reach _finish -- the root and the other peers are waiting in it.
        try:
            if cuda:                   # views of the root's memory: an opened handle is kept, keyed by its bytes, until close()
                views = []
                for blob in (xb, yb):
                    v = self._opened.get(blob)
                    if v is None:
                        if len(self._opened) >= 16:
                            self._opened.pop(next(iter(self._opened)))
                        v = self._opened[blob] = _open_shared(blob)
                    views.append(v)
                xv, yv = views
            else:
                for h in (xb, yb):
                    if h[1] not in self._opened:
                        self._opened[h[1]] = _ShmTensor(handle=h)
                xv, yv = self._opened[xb[1]].tensor, self._opened[yb[1]].tensor
            dtype = xv.dtype
            t = shape[1]
            alloc = torch.empty if do_scatter else torch.zeros
            bufs = [alloc((b - a, t) + tuple(shape[2:]), dtype=dtype, device=self.device) for a, b in pieces]
            if cuda:
                if self._streams is None:
                    self._streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
                pull_s, push_s = self._streams
                cur = torch.cuda.current_stream()
                pull_s.wait_stream(cur)
                push_s.wait_stream(cur)
                pulled = []
                if do_scatter:
                    for (a, b), buf in zip(pieces, bufs):     # every pull is queued up front: chunk c+1 lands while chunk c computes
                        with torch.cuda.stream(pull_s):
                            buf.copy_(xv[a:b], non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(pull_s)
                        pulled.append(ev)
            outs = []
            for c, ((a, b), buf) in enumerate(zip(pieces, bufs)):
                if do_scatter:
                    if cuda:
                        cur.wait_event(pulled[c])
                    else:
                        buf.copy_(xv[a:b])
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
                    if err is not None:
                        out = torch.zeros_like(buf)
                    self._note("fwd", c)
                if do_gather:
                    outs.append(out)
                    if cuda:
                        done = torch.cuda.Event()
                        done.record(cur)
                        with torch.cuda.stream(push_s):
                            push_s.wait_event(done)
                            yv[a:b].copy_(out, non_blocking=True)
                    else:
                        yv[a:b].copy_(out)
                    self._note("send", c)
            if cuda:
                cur.wait_stream(push_s)        # the status all-reduce below is behind the last push on this rank's stream ...
                cur.wait_stream(pull_s)
                cur.synchronize()              # ... and this rank does not enter it before its copies have completed
        except Exception as e:
            err = err or e
        self._finish(err)
        return None
