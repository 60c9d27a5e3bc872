"""CPU, world_size 2 over gloo: the data-parallel plumbing of the trainer.  The HIP
supervision path cannot run without a GPU, so a torch loss with the same batch structure
(mean over images, like both DSRG losses) is injected; what is checked is the part that is new
relative to the single-GPU reference: shard -> DDP all-reduce -> Caffe SGD == one process on the
global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from dsrg_amd.trainer import DSRGTrainer, CaffeSGD


class TinyNet(nn.Module):
    """same parameter-group structure as VGG16ASPP (weights/biases, fc8 lr multipliers), tiny"""

    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.MaxPool2d(3, 2, 1, ceil_mode=True))
        self.branches = nn.ModuleList([nn.Sequential(nn.Conv2d(8, 8, 3, padding=d, dilation=d), nn.ReLU(), nn.Identity(),
                                                     nn.Conv2d(8, 8, 1), nn.ReLU(), nn.Identity(), nn.Conv2d(8, 5, 1))
                                       for d in (1, 2)])

    def forward(self, x):
        f = self.features(x)
        return self.branches[0](f) + self.branches[1](f)

    caffe_param_groups = __import__("dsrg_amd.backbone", fromlist=["VGG16ASPP"]).VGG16ASPP.caffe_param_groups


def torch_loss(logits, images, labels, cues):
    """mean over images of a per-image normalised cross-entropy on the cues (structure of pylayers.py:136-137)"""
    lp = torch.log_softmax(logits, 1)
    per = -(cues * lp).sum((1, 2, 3)) / cues.sum((1, 2, 3)).clamp(min=1e-4)
    l = per.mean()
    return l, torch.stack([l.detach(), l.detach() * 0])


def make_data(B):
    g = torch.Generator().manual_seed(7)
    images = torch.randn(B, 3, 17, 17, generator=g)
    cues = (torch.rand(B, 5, 9, 9, generator=g) < 0.1).float()
    labels = torch.ones(B, 1, 1, 5)
    return images, labels, cues


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    # a bucket cap of ~1 KB splits TinyNet's 14 tensors over several buckets, as 32 MB does with the 37.9 M parameters of VGG16
    # (small_numel: TinyNet's kernels count as "large" tensors with slots of their own; its biases share the last bucket)
    tr = DSRGTrainer(torch.device("cpu"), world_size=world, seed=0, amp_dtype=None, channels_last=False,
                     loss_fn=torch_loss, net=TinyNet(), bucket_cap_mb=0.001)
    tr.reducer.remove()
    from dsrg_amd.reducer import BucketedAllReduce
    tr.reducer = BucketedAllReduce(list(tr.net.parameters()), bucket_cap_mb=0.001, small_numel=64)
    # the first thing the trainer's dropout stream yields on this rank (both ranks sit on "cpu": no device ordinal tells them apart,
    # as with one visible GPU per process)
    mask = torch.nn.functional.dropout(torch.ones(256), 0.5).clone()
    # the overlap of the gradient all-reduce with backward, observed: a communication hook (the default all-reduce, plus a
    # log line) fires per bucket as soon as the bucket's gradients exist; the first layer's weight gradient is the LAST thing
    # backward computes, so bucket events in front of it are all-reduces issued while backward was still running
    events = []
    launch = tr.reducer._launch

    def logged_launch(bi):
        events.append("bucket")
        return launch(bi)
    tr.reducer._launch = logged_launch
    tr.net.features[0].weight.register_hook(lambda g: events.append("first_layer_grad"))
    images, labels, cues = make_data(4)
    sh = slice(rank * 2, rank * 2 + 2)                       # rank r takes images [2r, 2r+2)
    shard_losses, reduced = [], []
    for _ in range(3):
        events.append("step")
        l = tr.step(images[sh], labels[sh], cues[sh])
        shard_losses.append(l.clone())
        reduced.append(tr.reduce_losses(l))                  # the 8-byte logging all-reduce
    equal, words = tr.weights_equal_across_ranks()          # what bench.py --gpus N prints and exits on
    if rank == 1:
        with torch.no_grad():
            tr.net.features[0].bias[0] += 1e-7               # one parameter of one replica drifts by an ulp-sized amount
    equal_after_drift, words_drift = tr.weights_equal_across_ranks()
    # gradients are views of the buckets; the fallback copies are counted (torch ops produce every gradient of this CPU net)
    assert all(p.grad.data_ptr() == tr.reducer._slot[p][1].data_ptr() for p in tr.net.parameters())
    torch.save({"w": [p.detach().clone() for p in tr.net.parameters()], "events": events, "shard": shard_losses,
                "nbuckets": len(tr.reducer.buckets), "launch_log": list(tr.reducer.launch_log),
                "reduced": reduced, "equal": equal, "words": words, "equal_after_drift": equal_after_drift,
                "words_drift": words_drift, "mask": mask, "dropout_seed": tr.dropout_stream_seed}, os.path.join(out_dir, "w%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process_global_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "w0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "w1.pt"))
    w0, w1 = r0["w"], r1["w"]
    for k, (a, b) in enumerate(zip(w0, w1)):
        if k == 1:                                           # features.0.bias: rank 1 injected a drift into its first element below
            assert torch.equal(a[1:], b[1:]) and a[0] != b[0] and abs(float(a[0] - b[0])) < 1e-6
            b[0] = a[0]
        assert torch.equal(a, b)                             # replicas stay in lock step
    # the self-check of the multi-GPU bench line: equal on both ranks while the replicas agree, unequal — on BOTH ranks — as soon
    # as one bit of one replica differs
    for r in (r0, r1):
        assert r["equal"] is True and r["words"][0] == r["words"][1] and len(r["words"]) == 2
        assert r["equal_after_drift"] is False and r["words_drift"][0] != r["words_drift"][1]
    assert r0["words"] == r1["words"] and r0["words_drift"] == r1["words_drift"]
    # per-rank dropout streams: seeded by the distributed rank (not by the device ordinal), so the ranks' masks differ
    assert r1["dropout_seed"] == r0["dropout_seed"] + 1 and not torch.equal(r0["mask"], r1["mask"])
    torch.manual_seed(0)
    tr = DSRGTrainer(torch.device("cpu"), world_size=1, seed=0, amp_dtype=None, channels_last=False,
                     loss_fn=torch_loss, net=TinyNet())
    images, labels, cues = make_data(4)
    global_losses = [tr.step(images, labels, cues).clone() for _ in range(3)]
    for a, b in zip(w0, tr.net.parameters()):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-6)
    # the logging all-reduce: the mean of the two shard losses = the global-batch loss of the single process, on both ranks
    for it in range(3):
        assert torch.allclose(r0["reduced"][it], r1["reduced"][it])
        assert torch.allclose(r0["reduced"][it], (r0["shard"][it] + r1["shard"][it]) / 2)
        assert torch.allclose(r0["reduced"][it], global_losses[it], rtol=1e-5, atol=1e-6)
    assert tr.reduce_losses(global_losses[0]) is global_losses[0]            # a single process: no collective
    # bucketed all-reduce overlapped with backward: several buckets per step, launched in bucket order on both ranks, all but the
    # last ones issued before backward has produced the first layer's gradient; the small-tensor bucket goes last
    for r in (r0, r1):
        steps = " ".join(r["events"]).split("step")[1:]
        assert len(steps) == 3 and r["nbuckets"] >= 3 and r["launch_log"] == list(range(r["nbuckets"]))
        for st in steps:
            ev = st.split()
            assert ev.count("first_layer_grad") == 1 and ev.count("bucket") == r["nbuckets"], ev
            assert ev[:ev.index("first_layer_grad")].count("bucket") >= 1, ev
            assert ev[-1] == "bucket", ev


def test_caffe_sgd_matches_hand_computation():
    p = nn.Parameter(torch.tensor([1.0, -2.0]))
    q = nn.Parameter(torch.tensor([0.5]))
    opt = CaffeSGD([dict(params=[p], lr_mult=1.0, decay_mult=1.0), dict(params=[q], lr_mult=2.0, decay_mult=0.0)],
                   base_lr=0.1, momentum=0.9, weight_decay=0.01, gamma=0.5, stepsize=2)
    v_p, v_q, w_p, w_q = np.zeros(2), np.zeros(1), np.array([1.0, -2.0]), np.array([0.5])
    for it in range(5):
        gp, gq = np.array([0.3, -0.1]) * (it + 1), np.array([0.2])
        p.grad, q.grad = torch.tensor(gp, dtype=torch.float32), torch.tensor(gq, dtype=torch.float32)
        lr = 0.1 * 0.5 ** (it // 2)                                   # lr_policy "step" (solver-s.prototxt:5-8)
        v_p = 0.9 * v_p + lr * (gp + 0.01 * w_p); w_p = w_p - v_p
        v_q = 0.9 * v_q + 2 * lr * gq; w_q = w_q - v_q
        opt.step()
        assert np.allclose(p.detach().numpy(), w_p, atol=1e-6) and np.allclose(q.detach().numpy(), w_q, atol=1e-6)


def _save_worker(rank, world, port, out_dir, bad):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = DSRGTrainer(torch.device("cpu"), world_size=world, seed=0, amp_dtype=None, channels_last=False,
                     loss_fn=torch_loss, net=TinyNet(), bucket_cap_mb=0.001)
    images, labels, cues = make_data(4)
    sh = slice(rank * 2, rank * 2 + 2)
    tr.step(images[sh], labels[sh], cues[sh])
    res = {}
    if bad:
        # rank 0 cannot write (the "directory" is a file): EVERY rank must raise instead of hanging in a barrier
        blocker = os.path.join(out_dir, "blocker")
        if rank == 0:
            open(blocker, "w").close()
        dist.barrier()
        try:
            tr.save(os.path.join(blocker, "sub", "model"))
            res["raised"] = False
        except RuntimeError as e:
            res["raised"] = "not written" in str(e)
    else:
        _, state = tr.save(os.path.join(out_dir, "snap", "model"))      # collective: both ranks
        res["exists"] = os.path.exists(state)                            # ... and the file is there when ANY rank returns
        tr2 = DSRGTrainer(torch.device("cpu"), world_size=world, seed=5, amp_dtype=None, channels_last=False,
                          loss_fn=torch_loss, net=TinyNet(), bucket_cap_mb=0.001)
        res["iter"] = tr2.load(state)
        res["same"] = all(torch.equal(a, b) for a, b in zip(tr.net.state_dict().values(), tr2.net.state_dict().values()))
    torch.save(res, os.path.join(out_dir, "save%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("bad", [False, True])
def test_two_rank_snapshot_is_collective_and_fails_on_every_rank(tmp_path, bad):
    mp.spawn(_save_worker, args=(2, _free_port(), str(tmp_path), bad), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(str(tmp_path), "save%d.pt" % r))
        if bad:
            assert res["raised"] is True
        else:
            assert res["exists"] and res["iter"] == 1 and res["same"]
