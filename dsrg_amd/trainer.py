"""One DSRG train-s step on MI355X: backbone forward (MIOpen, bf16 autocast) -> supervision
hot path (libdsrg_hip.so) -> backward -> Caffe-style SGD (solver-s.prototxt).  Data parallel
over images: one process per GPU, gradients all-reduced by RCCL in buckets the weight-gradient kernels write into (reducer.py).
"""
import os

import torch

from .backbone import VGG16ASPP
from .ops import dsrg_supervision_loss


_SGD_PACK = os.environ.get("DSRG_SGD_PACK", "1") != "0"     # the update + weight packing kernel of csrc/sgd_pack.hip (0: torch._fused_sgd_)


class CaffeSGD(object):
    """Caffe's SGDSolver update (solver-s.prototxt:5-14): V <- m V + lr*lr_mult*(g + wd*decay_mult*W);
    W <- W - V; lr = base_lr * gamma^floor(iter/stepsize).

    Kept in the equivalent form B = V / lr (B <- m B + (g + wd W); W <- W - lr B), which is what the one-pass fused
    multi-tensor kernel `torch._fused_sgd_` computes; when the learning rate steps, B is rescaled by lr_old / lr_new so
    that the trajectory stays exactly Caffe's (V carries the old rate in its history)."""

    def __init__(self, groups, base_lr=5e-4, momentum=0.9, weight_decay=5e-4, gamma=0.33, stepsize=1000):
        self.groups = groups
        self.base_lr, self.momentum, self.wd, self.gamma, self.stepsize = base_lr, momentum, weight_decay, gamma, stepsize
        self.iter = 0
        for g in self.groups:
            g["bufs"] = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in g["params"]]
            g["buf_lr"] = None
        if _SGD_PACK:
            # every write step() makes to a parameter is either dsrg_sgd_pack_f32's (which rewrites the packed bf16 kernels the
            # convolution nodes keep) or is followed by a bump of its version counter: the nodes may keep their packs
            from .ops import keep_weight_packs
            keep_weight_packs([p for g in self.groups for p in g["params"] if p.is_cuda], self)

    def lr(self):
        return self.base_lr * self.gamma ** (self.iter // self.stepsize)

    @torch.no_grad()
    def step(self):
        lr = self.lr()
        own = ([], [], [], [], [])                 # params, grads, bufs, lrs, wds of the tensors dsrg_sgd_pack_f32 updates
        for g in self.groups:
            ps = [p for p in g["params"] if p.grad is not None]
            if not ps:
                continue
            bufs = [b for p, b in zip(g["params"], g["bufs"]) if p.grad is not None]
            grads = [p.grad for p in ps]
            local_lr, local_wd = lr * g["lr_mult"], self.wd * g["decay_mult"]
            if g["buf_lr"] is not None and g["buf_lr"] != local_lr:
                torch._foreach_mul_(g["bufs"], g["buf_lr"] / local_lr)
            g["buf_lr"] = local_lr
            if _SGD_PACK and ps[0].is_cuda:
                # one launch series for all groups (the rates ride per tensor), the packed bf16 kernels of the convolution nodes
                # rewritten in the same pass (ops.sgd_pack_step); whatever does not fit (another dtype, a gradient laid out
                # differently) stays with torch below
                from .ops import sgd_pack_eligible
                rest = ([], [], [])
                for p, gr, b in zip(ps, grads, bufs):
                    if sgd_pack_eligible(p, gr, b):
                        for lst, v in zip(own, (p, gr, b, local_lr, local_wd)):
                            lst.append(v)
                    else:
                        for lst, v in zip(rest, (p, gr, b)):
                            lst.append(v)
                ps, grads, bufs = rest
                if not ps:
                    continue
            try:
                torch._fused_sgd_(ps, grads, bufs, weight_decay=local_wd, momentum=self.momentum, lr=local_lr,
                                  dampening=0.0, nesterov=False, maximize=False, is_first_step=False)
            except (RuntimeError, NotImplementedError):          # no fused kernel for this device / dtype mix
                if local_wd != 0.0:
                    grads = torch._foreach_add(grads, ps, alpha=local_wd)
                torch._foreach_mul_(bufs, self.momentum)
                torch._foreach_add_(bufs, grads)
                torch._foreach_add_(ps, bufs, alpha=-local_lr)
            torch.autograd.graph.increment_version(ps)           # torch._fused_sgd_ writes without bumping the version counters
        if own[0]:
            from .ops import sgd_pack_step
            sgd_pack_step(own[0], own[1], own[2], own[3], own[4], self.momentum)
        self.iter += 1

    def zero_grad(self):
        for g in self.groups:
            for p in g["params"]:
                p.grad = None

    def state_dict(self):
        """Caffe's .solverstate: iteration + momentum history (here B = V / lr and the rate it is scaled by)"""
        return {"iter": self.iter, "base_lr": self.base_lr,
                "groups": [{"lr_mult": g["lr_mult"], "decay_mult": g["decay_mult"], "buf_lr": g["buf_lr"],
                            "bufs": [b.detach().cpu() for b in g["bufs"]]} for g in self.groups]}

    @torch.no_grad()
    def load_state_dict(self, st):
        if len(st["groups"]) != len(self.groups):
            raise ValueError("solver state has %d parameter groups, the net %d" % (len(st["groups"]), len(self.groups)))
        for g, sg in zip(self.groups, st["groups"]):
            if (g["lr_mult"], g["decay_mult"]) != (sg["lr_mult"], sg["decay_mult"]) or len(g["bufs"]) != len(sg["bufs"]):
                raise ValueError("solver state does not match the parameter groups of this net")
            for b, sb in zip(g["bufs"], sg["bufs"]):
                b.copy_(sb.to(b.device, b.dtype))
            g["buf_lr"] = sg["buf_lr"]
        self.iter = int(st["iter"])
        # base_lr is NOT restored: Caffe's Solver::Restore takes the iteration and the history from the snapshot and the
        # learning-rate policy from the solver prototxt, so a run resumed with another rate uses the new one (the history
        # B = V / lr is rescaled at the next step through buf_lr)


@torch.no_grad()
def params_checksum(tensors):
    """-> int64[2] on the tensors' device: (sum of the 32-bit patterns, sum of pattern x (1 + index mod 65521)), both mod 2^64"""
    dev = tensors[0].device
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    for k, t in enumerate(tensors):
        t = t.detach().contiguous().view(-1)
        if t.element_size() != 4:
            t = t.float()
        bits = t.view(torch.int32).to(torch.int64)
        idx = torch.arange(bits.numel(), device=dev, dtype=torch.int64)
        acc[0] += bits.sum() + (k + 1)
        acc[1] += (bits * (1 + (idx + 7919 * k) % 65521)).sum()
    return acc


def distributed_rank():
    """this process's rank in the default process group (0 outside one)"""
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class DSRGTrainer(object):
    def __init__(self, device, world_size=1, seed=0, amp_dtype=torch.bfloat16, channels_last=True,
                 loss_fn=None, net=None, ddp=None, weights=None, snapshot=None, bucket_cap_mb=32):
        """loss_fn(logits, images, labels, cues) -> (total, losses); defaults to the HIP supervision
        path.  (Tests inject a torch loss to exercise the data-parallel plumbing on CPU/gloo.)
        weights: `train.py --weights` (run.sh:5: ../../vgg16_20M_mc.caffemodel) — a .caffemodel / .npz / torch file
        copied by layer name before training; snapshot: `train.py --snapshot` — a solverstate written by save()."""
        torch.manual_seed(seed)            # same initial weights on every rank (the reducer also broadcasts)
        self.device = device
        self.amp_dtype = amp_dtype
        self.channels_last = channels_last
        self.loss_fn = loss_fn or dsrg_supervision_loss
        # the bilateral lattices depend only on the images: build them on a side stream while the
        # backbone forward runs (HIP path only)
        self.overlap_build = loss_fn is None and device.type == "cuda"
        self.side = torch.cuda.Stream(device=device) if self.overlap_build else None
        net = net if net is not None else VGG16ASPP()
        if weights is not None:
            from .checkpoint import load_weights
            self.loaded_layers = load_weights(net, weights)
        net = net.to(device)
        if channels_last:
            net = net.to(memory_format=torch.channels_last)
        self.net = net
        self.model = net
        self.reducer = None
        if (world_size > 1) if ddp is None else ddp:
            # 151.5 MB of fp32 gradients per step; 32 MB buckets -> 5 all-reduces overlapped with backward, the weight-gradient
            # kernels writing straight into the buckets (dsrg_amd/reducer.py; torch's DistributedDataParallel copies every
            # gradient into its bucket: 47 launches per step, +3.8 % at one rank — DSRG_TORCH_DDP=1 restores it for A/B runs)
            if os.environ.get("DSRG_TORCH_DDP") == "1":
                from torch.nn.parallel import DistributedDataParallel as DDP
                self.model = DDP(net, device_ids=[device.index] if device.type == "cuda" else None, bucket_cap_mb=bucket_cap_mb,
                                 gradient_as_bucket_view=True)
            else:
                from .reducer import BucketedAllReduce
                self.reducer = BucketedAllReduce(list(net.parameters()), bucket_cap_mb=bucket_cap_mb)
        self.opt = CaffeSGD(net.caffe_param_groups())
        self.defer_bias = device.type == "cuda" and os.environ.get("DSRG_DEFER_REDUCTIONS", "1") != "0"      # 0: tools, A/B
        if snapshot is not None:
            self.load(snapshot)
        # per-rank dropout stream: seeded with the DISTRIBUTED rank, not the device ordinal — under a launcher that shows every
        # process one GPU (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES per rank) all of them are cuda:0, and they must still draw
        # different masks for their different images
        self.dropout_stream_seed = seed + 1 + distributed_rank()
        torch.manual_seed(self.dropout_stream_seed)

    def save(self, prefix="models/model-s"):
        """solver-s.prototxt:16-17 `snapshot_prefix`: <prefix>_iter_N.caffemodel + .solverstate.pt (rank 0 writes)"""
        from .checkpoint import save_snapshot
        return save_snapshot(self, prefix)

    def load(self, state_path):
        from .checkpoint import load_snapshot
        it = load_snapshot(self, state_path)
        if self.device.type == "cuda":                       # (load_state_dict bumps the version counters; belt and braces)
            from .ops import forget_weight_packs
            forget_weight_packs(self.net.parameters())
        return it

    def reduce_losses(self, losses):
        """the logging all-reduce of SURVEY 8e: `losses` of step() are this rank's shard means; their mean over ranks is the
        global-batch loss (both losses are means over images and the shards are equal).  One 8-byte all-reduce, outside the
        gradient path; a single process returns its input."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return losses
        out = losses.detach().clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out / dist.get_world_size()

    def weights_checksum(self):
        """two 64-bit words over the BIT PATTERNS of every parameter and momentum buffer, in parameter order: the plain sum and a
        position-weighted sum (so equal values in another order do not pass), exact integer arithmetic on the device"""
        return params_checksum(list(self.net.parameters()) + [b for g in self.opt.groups for b in g["bufs"]])

    def weights_equal_across_ranks(self):
        """data-parallel replicas must hold bit-identical weights after every step (same initial weights, same
        all-reduced gradients, same update).  All-gathers weights_checksum() (16 bytes per rank) -> (equal?, [[sum, weighted
        sum] per rank]); collective — every rank calls it; a single process returns (True, [its own])."""
        import torch.distributed as dist
        mine = self.weights_checksum()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return True, [[int(v) for v in mine.cpu()]]
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        words = [[int(v) for v in p.cpu()] for p in parts]
        return all(w == words[0] for w in words), words

    def step(self, images, labels, cues):
        """images (B,3,321,321) f32 mean-subtracted, labels (B,1,1,21), cues (B,21,41,41) -> losses[2] (this rank's shard;
        reduce_losses() gives the global-batch figure for logging)"""
        if self.reducer is not None:
            self.reducer.prepare()                          # gradients unset, every large parameter knows its bucket slot
        else:
            self.opt.zero_grad()
        x = images.contiguous(memory_format=torch.channels_last) if self.channels_last else images
        if self.overlap_build:
            from .ops import crf_prepare
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)                    # previous step no longer reads the lattices
            with torch.cuda.stream(self.side):
                crf_prepare(images, cues.shape[1], cues.shape[2], cues.shape[3])
        with torch.autocast(self.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            logits = self.model(x)
        logits = logits.float().contiguous()
        if self.overlap_build:
            torch.cuda.current_stream().wait_stream(self.side)
            total, losses = self.loss_fn(logits, images, labels, cues, prepared=True)
        else:
            total, losses = self.loss_fn(logits, images, labels, cues)
        # the fifteen 5-7 us passes that finish a bias gradient are recorded during backward and run as one launch behind it
        # (ops.deferred_reductions: same bits; nothing reads a bias gradient before the reducer / the update)
        from .ops import deferred_reductions
        with deferred_reductions(self.defer_bias):
            total.backward()
        if self.reducer is not None:
            self.reducer.finish()                           # the rest of the buckets out, the collectives joined; p.grad = the mean
        self.opt.step()
        return losses
