"""Multi-GPU layout of the hot path: one process per GPU, work split along independent units, no data-path collective.

SURVEY.md §8(e): every (sequence, target, reference) frame pair is independent for correlation / warp / flow utilities,
and context matching couples only the references of ONE target frame.  So the shard unit is the frame pair (or the
target frame) and ranks never exchange data on the hot path.  The only collective of the whole system is the gradient
all-reduce of the data-parallel inpainting training step (BASELINE.json configs[4]); `GradientBuckets` implements it as
bucketed, asynchronous all-reduces over torch.distributed ("nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  The reference itself has no distributed code (single process, single device).
"""
import torch
import torch.distributed as dist


def partition(n_units, world_size, rank):
    """Contiguous split of n_units over world_size ranks -> (start, stop) of `rank`; sizes differ by at most one."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank %d / world size %d" % (rank, world_size))
    base, extra = divmod(int(n_units), world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard(tensors, world_size, rank, dim=0):
    """This rank's slice along `dim` (the sequence / batch axis) of a tensor or of every tensor in a list / tuple / dict."""
    def one(t):
        if t is None:
            return None
        a, b = partition(t.shape[dim], world_size, rank)
        return t.narrow(dim, a, b - a)
    if isinstance(tensors, dict):
        return {k: one(v) for k, v in tensors.items()}
    if isinstance(tensors, (list, tuple)):
        return type(tensors)(one(t) for t in tensors)
    return one(tensors)


def world():
    """(rank, world_size) of the running job; (0, 1) outside torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_shards(local, dim=0):
    """Concatenate every rank's result along `dim` (tests / evaluation only: the hot path itself never gathers)."""
    rank, ws = world()
    if ws == 1:
        return local
    sizes = [None] * ws
    dist.all_gather_object(sizes, int(local.shape[dim]))
    longest = max(sizes)
    padded = local.contiguous()
    if padded.shape[dim] < longest:   # all_gather needs equal shapes: pad along dim, trim after
        pad_shape = list(padded.shape)
        pad_shape[dim] = longest - padded.shape[dim]
        padded = torch.cat([padded, padded.new_zeros(pad_shape)], dim=dim)
    outs = [torch.empty_like(padded) for _ in range(ws)]
    dist.all_gather(outs, padded)
    return torch.cat([o.narrow(dim, 0, n) for o, n in zip(outs, sizes)], dim=dim)


class GradientBuckets:
    """Bucketed gradient all-reduce for the data-parallel training step (each rank's gradient / world_size, summed over ranks).

    Parameters are packed, in reverse registration order (the order backward produces gradients in), into flat fp32
    buckets of ~bucket_mb MiB.  On 8 x MI355X each GPU has 7 xGMI links of ~153 GB/s; the 58.76 MB gradient of the
    inpainting network (14,690,883 parameters) is three 25 MiB buckets, each far above the message size where RCCL's ring
    reaches its per-link bandwidth.

    overlap=False: `reduce()` (after backward) packs every bucket and launches one asynchronous all-reduce per bucket, so the
        collective of bucket k overlaps the packing of bucket k+1; `wait()` copies the averaged gradients back.
    overlap=True (built BEFORE backward): every `p.grad` becomes a view into its bucket (no pack / unpack copies) and a
        post-accumulate hook launches a bucket's all-reduce the moment its last gradient has been accumulated, so the
        collectives run under the rest of backward; `reduce()` then only launches buckets that backward did not complete
        (parameters without a gradient count as zeros) and `wait()` only waits.  Buckets are launched strictly in index
        order on every rank, whatever order the hooks fire in.  One backward per reduce()/wait() cycle;
        use `zero_grad(set_to_none=False)` so that the views survive (a replaced gradient is copied in and out instead)."""

    def __init__(self, params, bucket_mb=25.0, overlap=False):
        self.params = [p for p in params if p.requires_grad]
        limit = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, cur, cur_n = [], [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > limit:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        nb = len(self.buckets)
        self._flat = [None] * nb
        self._handles = [None] * nb
        self._launched = [False] * nb
        self._ready = [0] * nb
        self._next = 0          # first bucket not yet launched in this cycle: launches happen in index order only
        self.overlap = bool(overlap)
        self._hooks = []
        if self.overlap:
            for i, bucket in enumerate(self.buckets):
                self._bind(i)
                for p in bucket:
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i)))

    # -- flat storage
    def _flat_of(self, i):
        bucket = self.buckets[i]
        ref = bucket[0]
        if self._flat[i] is None or self._flat[i].device != ref.device:
            self._flat[i] = torch.zeros(sum(p.numel() for p in bucket), dtype=torch.float32, device=ref.device)
        return self._flat[i]

    def _slots(self, i):
        flat, off = self._flat_of(i), 0
        for p in self.buckets[i]:
            yield p, flat[off:off + p.numel()]
            off += p.numel()

    @staticmethod
    def _is_view(p, slot):
        return p.grad is not None and p.grad.data_ptr() == slot.data_ptr() and p.grad.dtype == slot.dtype and p.grad.is_contiguous()

    def _bind(self, i):
        for p, slot in self._slots(i):
            if not self._is_view(p, slot):
                if p.grad is not None:
                    slot.copy_(p.grad.reshape(-1))
                p.grad = slot.view_as(p)

    # -- launching
    def _on_grad(self, i):
        # Collectives are matched across ranks purely by ISSUE ORDER, so buckets are launched strictly in bucket-index order
        # (as DDP does): a bucket that completes early waits until every bucket before it has been launched.  A rank whose
        # graph leaves a parameter without a gradient simply stops launching from hooks at that bucket; reduce() flushes the
        # rest in the same order, so every rank issues bucket 0, 1, 2, ... whatever its graph looked like.
        self._ready[i] += 1
        assert self._ready[i] <= len(self.buckets[i]), \
            "bucket %d saw more gradient accumulations than parameters: one backward per reduce()/wait() cycle" % i
        while self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next]):
            self._launch(self._next)

    def _launch(self, i):
        rank, ws = world()
        flat = self._flat_of(i)
        for p, slot in self._slots(i):      # gradients that are not views of the bucket: pack them (zeros if absent)
            if not self._is_view(p, slot):
                if p.grad is None:
                    slot.zero_()
                else:
                    slot.copy_(p.grad.reshape(-1))
        if ws > 1:
            flat.div_(ws)
            self._handles[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._launched[i] = True
        self._next = i + 1

    def reduce(self):
        for i in range(self._next, len(self.buckets)):
            self._launch(i)
        return self

    def wait(self):
        for i in range(len(self.buckets)):
            if self._handles[i] is not None:
                self._handles[i].wait()
            for p, slot in self._slots(i):
                if self._is_view(p, slot):
                    continue
                if self.overlap:
                    p.grad = slot.view_as(p)      # re-establish the view a zero_grad(set_to_none=True) dropped
                else:
                    if p.grad is None:
                        p.grad = torch.empty_like(p)
                    p.grad.copy_(slot.view_as(p))
            self._handles[i], self._launched[i], self._ready[i] = None, False, 0
        self._next = 0

    def launched(self):
        """how many buckets are already in flight (overlap mode: counted after backward, before reduce())"""
        return sum(self._launched)

    def nbytes(self):
        return 4 * sum(p.numel() for p in self.params)
