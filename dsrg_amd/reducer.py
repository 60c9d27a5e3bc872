"""Gradient all-reduce for data-parallel training (the reference trains on one GPU — training/tools/train.py:26-28 —
so this is new work): flat fp32 buckets, one RCCL all-reduce per bucket launched while backward is still running, and NO copy
of a gradient into its bucket.

torch's DistributedDataParallel copies every gradient into its bucket as the gradient becomes ready (Reducer::
mark_variable_ready_dense: one `mul` launch per parameter, 47 per train-s step, +3.8 % at one rank).  Here a parameter's slot of
the bucket IS where its gradient is written:

  * before backward every LARGE parameter (the convolution kernels: 99.7 % of the bytes) carries `p._dsrg_grad_out` = the view of
    its slot (the parameter's own strides); the weight-gradient launches of this package write there (ops.conv_igemm_wgrad /
    conv_igemm_backward / conv3x3_wgrad, `out=`) and the node returns a fresh alias, which autograd's AccumulateGrad adopts as
    `p.grad` without a copy;
  * a post-accumulate hook per large parameter counts the bucket's gradients; a gradient that did not land in its slot (some
    torch op produced it) is copied there by the hook — the fallback, one launch;
  * the small tensors (biases, the 21-output classifiers: ~50 tensors, < 1 MB) share one last bucket, filled by ONE multi-tensor
    copy after backward;
  * a bucket whose gradients are all there is divided by the world size (one launch per bucket, none at one rank) and all-reduced
    asynchronously; buckets are launched in bucket order on every rank;
  * finish() (after backward) launches what is left, waits for the collectives on the compute stream and leaves `p.grad` = the
    reduced slot for the optimizer.

Parameters are bucketed in REVERSE registration order (the order backward produces their gradients, to a first approximation).
"""
import torch


def grad_destination(param, shape=None, dtype=torch.float32):
    """the tensor a gradient kernel should write `param`'s gradient into when a reducer has given the parameter a slot (a fresh
    alias of the slot: autograd adopts a gradient only when nobody else holds the tensor object), or None"""
    out = getattr(param, "_dsrg_grad_out", None)
    if out is None or out.dtype != dtype or (shape is not None and tuple(out.shape) != tuple(shape)):
        return None
    return out.detach()


class BucketedAllReduce(object):
    def __init__(self, params, world_size=None, bucket_cap_mb=32.0, process_group=None, broadcast=True, small_numel=32768):
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = process_group
        self.world = world_size if world_size is not None else (self.dist.get_world_size(process_group) if self.dist else 1)
        self.params = [p for p in params if p.requires_grad]
        if broadcast and self.dist is not None and self.world > 1:
            with torch.no_grad():
                for p in self.params:                          # same initial weights on every rank
                    self.dist.broadcast(p.data, 0, group=process_group)
        cap = max(1, int(bucket_cap_mb * (1 << 20)) // 4)       # elements per bucket
        self.buckets = []                                       # [flat tensor, [params], pending count, launched?]
        # small tensors (biases, the 21-output classifiers: ~50 tensors, < 1 MB together) share ONE last bucket that is filled by a
        # single multi-tensor copy in finish(): a launch per tensor would cost more than the bytes
        self.small = [p for p in self.params if p.numel() < small_numel or p.dtype != torch.float32]
        small_ids = set(id(p) for p in self.small)
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if id(p) in small_ids:
                continue
            n = p.numel()
            if cur and cur_n + n > cap:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += (n + 63) // 64 * 64                        # slots start on 256-byte boundaries
        if cur:
            self._close(cur, cur_n)
        self.n_big = len(self.buckets)
        if self.small:
            self._close(self.small, sum((p.numel() + 63) // 64 * 64 for p in self.small))
        self._slot = {}
        for bi, (flat, ps, _, _) in enumerate(self.buckets):
            off = 0
            for p in ps:
                n = p.numel()
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                view = flat[off:off + n].as_strided(p.shape, p.stride()) if dense else flat[off:off + n].view(p.shape)
                self._slot[p] = (bi, view)
                off += (n + 63) // 64 * 64
        self._handles = []
        self._next = 0
        self.copies = 0                                         # gradients that had to be copied into their slot (tests, tools)
        self.copied = []                                        # ... and whose
        self.launch_log = []                                    # bucket indices in launch order of the last step (tests)
        self._hooks = [p.register_post_accumulate_grad_hook(self._ready) for p in self.params if id(p) not in small_ids]

    def _close(self, ps, n):
        dev = ps[0].device
        self.buckets.append([torch.zeros(n, dtype=torch.float32, device=dev), list(ps), len(ps), False])

    def prepare(self):
        """before backward: gradients unset, every parameter knows its slot"""
        for b in self.buckets:
            b[2], b[3] = len(b[1]), False
        self._next = 0
        self._handles = []
        self.launch_log = []
        self._seen = set()
        small_ids = set(id(p) for p in self.small)
        for p in self.params:
            p.grad = None
            p._dsrg_grad_out = self._slot[p][1] if id(p) not in small_ids else None

    @torch.no_grad()
    def _ready(self, p):
        bi, view = self._slot[p]
        g = p.grad
        if g is None or id(p) in self._seen:                    # (one gradient per parameter and step: no accumulation across calls)
            return
        self._seen.add(id(p))
        if not (g.dtype == torch.float32 and g.data_ptr() == view.data_ptr() and g.stride() == view.stride()):
            view.copy_(g)                                       # (a gradient some torch op produced: the one copy DDP makes for all)
            self.copies += 1
            self.copied.append(p)
        p.grad = view
        b = self.buckets[bi]
        b[2] -= 1
        self._launch_ready()

    def _launch_ready(self):
        while self._next < len(self.buckets) and self.buckets[self._next][2] <= 0 and not self.buckets[self._next][3]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi):
        b = self.buckets[bi]
        b[3] = True
        self.launch_log.append(bi)
        if self.dist is None or self.world == 1:
            return                                              # one rank: nothing to reduce (a 1-rank RCCL all-reduce still costs a launch per bucket)
        flat = b[0]
        flat.div_(self.world)                                   # once per bucket (SUM of the pre-divided shards = the mean)
        self._handles.append(self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    @torch.no_grad()
    def finish(self):
        """after backward: parameters that received no gradient count as zero, the remaining buckets go out in order, the compute
        stream waits for every collective; p.grad = the reduced slot"""
        for bi in range(self._next, len(self.buckets)):
            b = self.buckets[bi]
            if b[3]:
                continue
            dst, src = [], []
            for p in b[1]:
                view = self._slot[p][1]
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr() or p.grad.stride() != view.stride():
                    dst.append(view); src.append(p.grad)
                p.grad = view
            if dst:
                torch._foreach_copy_(dst, src)                  # one multi-tensor launch for the whole bucket
                if bi < self.n_big:
                    self.copies += len(dst)
            b[2] = 0
            self._launch(bi)
        self._next = len(self.buckets)
        for h in self._handles:
            h.wait()
        self._handles = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            p._dsrg_grad_out = None
