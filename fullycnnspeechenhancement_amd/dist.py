"""Batch-axis sharding of the forward pass over the GPUs of one node (one process per GPU).

The reference has no parallelism of any kind (single tf.Session on one device: SURVEY F9); what
shards naturally here is the utterance (batch) axis of `sess.run(pred, {input_x: x})`
(model_utils/tester.py:85-90): inference BatchNorm uses moving statistics, so utterances are
independent and NO collective is needed on the data path when each rank already owns its shard
(`forward_resident`, what bench.py times).  When one rank (the reference's single host process)
holds the whole batch, `forward_from_root` scatters contiguous batch slices and gathers the masks
back with grouped point-to-point transfers (ncclSend/ncclRecv under torch.distributed's "nccl"
backend = RCCL; one peer per xGMI link, so the root's 7 links work in parallel), optionally in
chunks so that transfers overlap compute.  The same code runs on the "gloo" backend with CPU
tensors, which is how tests/ cover the world_size > 1 logic without GPUs.
"""

import torch
import torch.distributed as dist


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


class BatchShardedForward(object):
    """forward: callable mapping a [n, T, 129, 1] tensor on this rank's device to the same shape
    (a fullycnnspeechenhancement_amd model on GPU; any stand-in under gloo in tests)."""

    def __init__(self, forward, group=None, device=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (one process per GPU)")
        self.forward = forward
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device) if device is not None else torch.device("cpu")

    # -- every rank already holds its own utterances: no data-path collective -----------------
    def forward_resident(self, x_local):
        return self.forward(x_local)

    # -- one rank holds the whole batch (the reference's calling convention) -------------------
    def forward_from_root(self, x_root, root=0, chunks=1):
        """x_root: [N, T, 129, 1] on `root` (ignored elsewhere).  Returns [N, T, 129, 1] on root, None
        on the other ranks.  `chunks` > 1 pipelines each peer's slice: receive chunk c+1 while
        computing chunk c, send results back as they finish."""
        meta = [None]
        if self.rank == root:
            if x_root.dim() != 4 or x_root.shape[2] != 129 or x_root.shape[3] != 1:
                raise ValueError("input must be [N, T, 129, 1], got %s" % (tuple(x_root.shape),))
            meta = [(tuple(x_root.shape), str(x_root.dtype))]
        dist.broadcast_object_list(meta, src=root, group=self.group)
        shape, dtype_name = meta[0]
        dtype = getattr(torch, dtype_name.split(".")[-1])
        n, t = shape[0], shape[1]
        bounds = shard_bounds(n, self.world)

        if self.rank == root:
            x_root = x_root.contiguous()
            y = torch.empty_like(x_root)
            sends, recvs = [], []
            for r in range(self.world):
                if r == root:
                    continue
                for lo, hi in chunk_bounds(*bounds[r], chunks):
                    sends.append(dist.P2POp(dist.isend, x_root[lo:hi], r, self.group))
                    recvs.append(dist.P2POp(dist.irecv, y[lo:hi], r, self.group))
            works = dist.batch_isend_irecv(sends + recvs) if sends else []
            lo, hi = bounds[root]
            if hi > lo:   # the root's own slice computes while its links carry the others'
                for a, b in chunk_bounds(lo, hi, chunks):
                    y[a:b] = self.forward(x_root[a:b])
            for w in works:
                w.wait()
            return y

        lo, hi = bounds[self.rank]
        pieces = chunk_bounds(lo, hi, chunks)
        if not pieces:
            return None
        bufs = [torch.empty((b - a, t) + tuple(shape[2:]), dtype=dtype, device=self.device) for a, b in pieces]
        recv_works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, root, self.group) for buf in bufs])
        send_works, outs = [], []
        for i, buf in enumerate(bufs):
            # batch_isend_irecv may return one work for the whole group (nccl) or one per op (gloo)
            recv_works[min(i, len(recv_works) - 1)].wait()
            out = self.forward(buf).contiguous()
            outs.append(out)   # keep alive until sent
            send_works += dist.batch_isend_irecv([dist.P2POp(dist.isend, out, root, self.group)])
        for w in send_works:
            w.wait()
        return None
