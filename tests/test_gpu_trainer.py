"""GPU (-m gpu): the configurations BASELINE.json quotes, end to end.

  configs[1]  VGG16 forward + CRF + SRG, batch 1 (inference-only supervision path) on the backbone's own scores
  configs[2]  full seed_mc train-s step, batch 16: DSRGTrainer.step (side-stream lattice build, autocast, CaffeSGD)
  configs[3]  the same step under torch.distributed.run with the nccl (= RCCL) backend and DDP, one rank
  configs[4]  RetrainTrainer.step with the full ResNet-101 DeepLab at 513x513
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from dsrg_amd import synthetic as S
from test_gpu_parity import dev, _check_fused_step, CRF_TOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def ops():
    from dsrg_amd import ops, _lib
    _lib.require_gpu()
    return ops


def _batch(B, seed=41):
    b = S.make_batch(seed, B)
    return dev(b["images"]), dev(b["labels"]), dev(b["cues"])


def _dropout_off(net):
    from dsrg_amd.backbone import GemmConv2d
    for m in net.modules():
        if isinstance(m, GemmConv2d):
            m.fuse_dropout = 0.0


def test_config2_backbone_forward_then_supervision_b1(ops, O):
    """BASELINE configs[1]: one 321x321 image through the VGG16-ASPP forward (bf16 autocast, fp32 heads), then
    Softmax -> CRF -> SRG (+ losses) on the backbone's OWN scores, against the oracle layer by layer"""
    from dsrg_amd.backbone import VGG16ASPP
    torch.manual_seed(0)
    net = VGG16ASPP().cuda().to(memory_format=torch.channels_last).eval()
    b = S.make_batch(77, 1)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        scores = net(dev(b["images"]).contiguous(memory_format=torch.channels_last))
    assert scores.dtype == torch.float32 and scores.shape == (1, 21, 41, 41)
    # a random-init net scores ~1e-2 (fc8-SEC ~ N(0, 0.01)): stretch to the range of a trained net so that the CRF and
    # the region growing have decisions to take; both sides see the same numbers
    scores = (scores - scores.mean()) * (8.0 / scores.std().clamp(min=1e-12))
    # ... and lean towards the cued classes around their cues, as a net a few hundred iterations into training does, so
    # that the region growing has something to grow
    from scipy.ndimage import gaussian_filter
    bump = np.stack([gaussian_filter(c, 3.0) for c in b["cues"][0]]).astype(np.float32)
    bump = (bump / np.maximum(bump.max((1, 2), keepdims=True), 1e-12))[None]               # peak 1 for every cued class
    logits = np.ascontiguousarray(scores.cpu().numpy() + 60.0 * bump)
    _check_fused_step(ops, O, logits, b["images"], b["labels"], b["cues"], "config 2 (B=1, backbone scores)")
    probs = O.softmax_forward(logits)
    refined, _ = O.crf_refine_batch(probs, b["images"], 12.0, 10)
    grown = int(O.srg_grow_batch(b["labels"], b["cues"], refined).sum() - b["cues"].sum())
    print("config 2: max refined %.4f, grown %d" % (refined.max(), grown))
    assert grown > 0


def test_config3_train_step_b16(ops):
    """BASELINE configs[2]: DSRGTrainer.step at batch 16 exactly as bench.py runs it (bf16 autocast, dropout on, lattices
    built on the side stream): finite losses, every parameter moves, and the first update obeys Caffe's rule with the
    lr / decay multipliers of train-s.prototxt (weights 1/1, biases 2/0, fc8-SEC 10/1 and 20/0; solver-s.prototxt:5-14)"""
    from dsrg_amd.trainer import DSRGTrainer
    device = torch.device("cuda", 0)
    tr = DSRGTrainer(device, seed=0)
    assert tr.overlap_build
    images, labels, cues = _batch(16)
    before = [p.detach().clone() for g in tr.opt.groups for p in g["params"]]
    l0 = tr.step(images, labels, cues)
    torch.cuda.synchronize()
    assert torch.isfinite(l0).all() and l0.shape == (2,)
    lr, wd = tr.opt.base_lr, tr.opt.wd
    seen = set()
    i = moved = 0
    for g in tr.opt.groups:
        seen.add((g["lr_mult"], g["decay_mult"]))
        for p in g["params"]:
            w0 = before[i]
            i += 1
            assert p.grad is not None and torch.isfinite(p.grad).all()
            want = w0 - lr * g["lr_mult"] * (p.grad + wd * g["decay_mult"] * w0)      # V0 = 0 (solver-s.prototxt: momentum 0.9)
            assert torch.allclose(p.detach(), want, rtol=1e-5, atol=1e-8)
            moved += int(not torch.equal(p.detach(), w0))
            if g["decay_mult"] == 1.0:
                assert not torch.equal(p.detach(), w0)          # every weight tensor moves (a bias whose update is below
    assert moved >= 0.9 * len(before)                           # half an ulp of its value may stay put in fp32)
    assert seen == {(1.0, 1.0), (2.0, 0.0), (10.0, 1.0), (20.0, 0.0)}
    l1 = tr.step(images, labels, cues)
    assert torch.isfinite(l1).all() and tr.opt.iter == 2
    print("config 3 losses:", [float(x) for x in l0], [float(x) for x in l1])


def test_trainer_prepared_lattices_equal_inline_and_fp32_twin(ops):
    """(a) lattices built on the side stream under the backbone forward == built inline after it (bit-equal losses and
    weights); (b) the bf16-autocast step against an fp32 twin of the same net on the same batch (dropout off): the fp32
    heads keep the two loss values close, and the fp32 path itself runs (bench.py's fp32 headline uses it)"""
    from dsrg_amd.trainer import DSRGTrainer
    device = torch.device("cuda", 0)
    images, labels, cues = _batch(4, seed=43)

    def run(amp, overlap, steps=2):
        tr = DSRGTrainer(device, seed=5, amp_dtype=amp)
        _dropout_off(tr.net)
        tr.overlap_build = overlap
        out = [tr.step(images, labels, cues).clone() for _ in range(steps)]
        torch.cuda.synchronize()
        return out, tr
    la, ta = run(torch.bfloat16, True)
    lb, tb = run(torch.bfloat16, False)
    for x, y in zip(la, lb):
        assert torch.equal(x, y)
    for pa, pb in zip(ta.net.parameters(), tb.net.parameters()):
        assert torch.equal(pa, pb)
    lc, tc = run(None, True)
    for x, y in zip(la, lc):
        assert torch.isfinite(y).all()
        assert torch.allclose(x, y, rtol=0.03, atol=3e-3), (x, y)
    print("bf16 losses", [[float(v) for v in x] for x in la], "fp32 losses", [[float(v) for v in x] for x in lc])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        ta.net.eval()
        s16 = ta.net(images.contiguous(memory_format=torch.channels_last))
    assert s16.dtype == torch.float32                                    # heads and their sum stay fp32 under autocast


DDP_WORKER = r"""
import os, sys, json
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from dsrg_amd import synthetic as S
from dsrg_amd.trainer import DSRGTrainer
rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
device = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=device)
b = S.make_batch(51, 16)                                   # (16 images: every wide layer on the implicit-GEMM route, as in the bench)
d = lambda a: torch.from_numpy(a).to(device)
images, labels, cues = d(b["images"]), d(b["labels"]), d(b["cues"])
events = []
def run(ddp):
    tr = DSRGTrainer(device, world_size=dist.get_world_size(), seed=9, ddp=ddp)
    if ddp:
        # one all-reduce per gradient bucket, issued while backward is still running (logged); the first convolution's weight
        # gradient is the last one backward produces
        launch = tr.reducer._launch
        def logged(bi):
            events.append("bucket")
            return launch(bi)
        tr.reducer._launch = logged
        next(tr.net.parameters()).register_hook(lambda g: events.append("first_layer_grad"))
    out = []
    for _ in range(3):
        events.append("step")
        l = tr.step(images, labels, cues)
        out.append([float(v) for v in tr.reduce_losses(l)])
    torch.cuda.synchronize()
    if ddp:
        red = tr.reducer
        info.update(copies=red.copies, nbuckets=len(red.buckets), nbig=sum(len(b[1]) for b in red.buckets[:red.n_big]),
                    nsmall=len(red.small), aliased=all(p.grad.data_ptr() == red._slot[p][1].data_ptr() for p in tr.net.parameters()))
    return out, [p.detach().clone() for p in tr.net.parameters()]
info = {}
l_ddp, w_ddp = run(True)
ddp_events = list(events)
l_one, w_one = run(False)
t = torch.ones(1, device=device); dist.all_reduce(t)
same_w = all(torch.equal(a, b) for a, b in zip(w_ddp, w_one))
print("DDPRESULT " + json.dumps({"ddp": l_ddp, "one": l_one, "same_weights": bool(same_w), "allreduce": float(t.item()),
                                 "backend": dist.get_backend(), "events": ddp_events, "info": info}))
dist.destroy_process_group()
"""


def test_config4_ddp_rccl_single_rank(tmp_path):
    """BASELINE configs[3] on the one GPU there is: bench.py's launch line (torch.distributed.run, nccl = RCCL, VGG16ASPP with
    the custom autograd functions, side-stream lattice build, gradients landing in the reducer's buckets — dsrg_amd/reducer.py)
    for 3 steps == the same 3 steps without a process group, bit for bit; and NO gradient of a large parameter is copied into its
    bucket (torch's DistributedDataParallel made 47 copies per step: +3.8 % at one rank)"""
    import json
    import socket
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER % {"root": ROOT})
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MIOPEN_FIND_MODE=os.environ.get("MIOPEN_FIND_MODE", "2"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DDPRESULT ")][-1]
    res = json.loads(line[len("DDPRESULT "):])
    assert res["backend"] == "nccl" and res["allreduce"] == 1.0
    assert np.allclose(res["ddp"], res["one"], rtol=1e-5, atol=1e-6), res
    assert res["same_weights"]
    assert np.isfinite(res["ddp"]).all()
    # every weight-gradient kernel wrote into its parameter's slot: zero fallback copies over the three steps; the small tensors
    # (biases, classifiers) share the last bucket; p.grad is a view of the bucket for every parameter
    info = res["info"]
    assert info["copies"] == 0 and info["aliased"] and info["nbig"] >= 20 and info["nsmall"] >= 25, info
    # 151.5 MB of fp32 gradients in 32 MB buckets: five or six all-reduces per step, in bucket order, all but the last two issued
    # before backward has reached conv1_1
    steps = " ".join(res["events"]).split("step")[1:]
    assert len(steps) == 3
    for st in steps:
        ev = st.split()
        assert ev.count("bucket") == info["nbuckets"] >= 5 and ev.count("first_layer_grad") == 1, ev
        assert ev[:ev.index("first_layer_grad")].count("bucket") >= 3, ev


DDP2_WORKER = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from dsrg_amd import synthetic as S
from dsrg_amd.trainer import DSRGTrainer
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
device = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=device)
b = S.make_batch(61, 2 * world)
d = lambda a: torch.from_numpy(a).to(device)
images, labels, cues = d(b["images"]), d(b["labels"]), d(b["cues"])
sh = slice(2 * rank, 2 * rank + 2)                           # rank r takes images [2r, 2r+2) of the global batch
def run(ddp, imgs, labs, cs):
    tr = DSRGTrainer(device, world_size=world if ddp else 1, seed=9, ddp=ddp)
    tr.net.eval()                                            # no dropout: the shards' masks would differ from the global batch's
    w0 = [p.detach().clone() for p in tr.net.parameters()]
    out = []
    for _ in range(2):
        out.append([float(v) for v in tr.reduce_losses(tr.step(imgs, labs, cs))] if ddp else [float(v) for v in tr.step(imgs, labs, cs)])
    torch.cuda.synchronize()
    return out, w0, [p.detach().clone() for p in tr.net.parameters()]
l_ddp, w0, w_ddp = run(True, images[sh], labels[sh], cues[sh])
# replicas in lock step: every rank's weights bit-equal to rank 0's
flat = torch.cat([p.flatten() for p in w_ddp])
ref = flat.clone(); dist.broadcast(ref, 0)
same = torch.tensor([float(torch.equal(flat, ref))], device=device); dist.all_reduce(same, op=dist.ReduceOp.MIN)
res = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "lockstep": float(same.item())}
if rank == 0:
    l_one, w0b, w_one = run(False, images, labels, cues)     # one process on the global batch
    worst = 0.0
    for a0, a, b_ in zip(w0, w_ddp, w_one):
        upd = (b_ - a0).abs().max().item()
        if upd > 0:
            worst = max(worst, (a - b_).abs().max().item() / upd)
    res.update({"ddp": l_ddp, "one": l_one, "worst_update_gap": worst})
    print("DDP2RESULT " + json.dumps(res))
dist.barrier()
dist.destroy_process_group()
"""


def test_config4_ddp_rccl_two_ranks(tmp_path):
    """BASELINE configs[3] at the smallest world size that has a real exchange: two ranks over RCCL, each with its shard of a
    4-image batch, two train steps; the replicas stay bit-equal and equal the single-process run on the global batch (both
    losses are means over images, so averaged shard gradients are the global-batch gradient — up to the bf16 rounding of
    each shard's weight-gradient GEMMs).  Needs two GPUs; bench.py --gpus 2 is exercised the same way."""
    import json
    import socket
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-rank RCCL step needs two")
    script = tmp_path / "ddp2_worker.py"
    script.write_text(DDP2_WORKER % {"root": ROOT})
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MIOPEN_FIND_MODE=os.environ.get("MIOPEN_FIND_MODE", "2"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("DDP2RESULT ")][-1][len("DDP2RESULT "):])
    assert res["ranks"] == 2 and res["backend"] == "nccl" and res["lockstep"] == 1.0, res
    assert np.allclose(res["ddp"], res["one"], rtol=2e-3, atol=1e-4), res
    assert res["worst_update_gap"] < 0.05, res                # bf16 gradient GEMMs per shard vs per global batch
    # and bench.py's own launcher: plain `python bench.py --gpus 2` must start both ranks and report them
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--no-profile"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["global_batch"] == 4
    assert line["weights_equal_across_ranks"] is True        # the ranks all-gathered their checksums after the timed steps


def test_bench_line_under_the_launcher_checks_the_replicas(tmp_path):
    """the driver's N>1 launch line at the one world size a one-GPU box has: `python -m torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1` goes through the RCCL process group, DDP and the checksum all-gather that N=8 goes through, and the line
    says so (rccl_ranks, weights_equal_across_ranks); a plain run reports no process group and a trivially equal replica set"""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTORCH_TUNABLEOP_ENABLED="0")
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-fp32",
            "--no-cpu-baseline", "--no-modes", "--no-profile"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port)] + tail, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["weights_equal_across_ranks"] is True and len(line["weights_checksum_rank0"]) == 16
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 2 and np.isfinite(line["losses"]).all()


def test_bf16_route_computes_the_float32_gradient():
    """Round-4 review item 2: per-parameter COSINE between the shipped bf16 route's gradient and the float32 backbone's (the
    reference's Caffe precision) at an ImageNet-scale initialisation, batch 16 (BASELINE configs[2]), Dropout off, the same
    train-s score gradient fed to both (dsrg_amd/fidelity.py).  Bars: every parameter >= 0.98, the whole gradient >= 0.998; no
    worse than torch's own bf16 autocast on the same net; and within a hair of the yardstick — the float32 gradient itself
    after the weights alone were rounded to bf16 once (a ReLU net's gradient is piecewise constant: any 2^-9 perturbation
    flips the masks of the units that sit at zero, which is where all of the distance comes from — profiles/r05_grad_fidelity*)."""
    from dsrg_amd.fidelity import gradient_fidelity
    r = gradient_fidelity(16, ("bf16", "stock", "f32,w16"))
    worst = {t: min(r["cos"][t].values()) for t in r["cos"]}
    print("min cosine per leg:", worst, "whole gradient:", r["cos_all"])
    assert worst["bf16"] >= 0.98, (worst, min(r["cos"]["bf16"], key=r["cos"]["bf16"].get))
    assert r["cos_all"]["bf16"] >= 0.998
    assert worst["bf16"] >= worst["stock"] - 0.01                       # as faithful as stock bf16 autocast
    assert worst["bf16"] >= worst["f32,w16"] - 0.015                    # and about as close as float32 is to itself 2^-9 away
    # the upper layers, where no deep chain of masks sits in between, agree to three digits and more
    assert all(c >= 0.999 for n, c in r["cos"]["bf16"].items() if n.startswith("branches.") and n.split(".")[2] in ("3", "6"))


def test_train_step_with_dropout_on_trains_like_the_float32_leg():
    """the shipped configuration — Dropout 0.5 drawn inside the convolution epilogues, ReLU / Dropout backward riding in the data
    gradients — repeated on one batch: the total loss falls as the float32 backbone's does (torch's own Dropout there: other
    masks, same statistics).  The Dropout-off comparison of the two legs' arithmetic is bench.py's `legs.loss_gap_300_steps`
    and profiles/r05_overfit_two_legs.txt; this is the default-test-set check that the fused mask path trains at all."""
    from dsrg_amd import synthetic as S
    from dsrg_amd.backbone import VGG16ASPP
    from dsrg_amd.trainer import DSRGTrainer
    dev = torch.device("cuda", 0)
    b = S.make_batch(7, 4)
    images, labels, cues = (torch.from_numpy(b[k]).to(dev) for k in ("images", "labels", "cues"))
    tot = {}
    for tag, amp in (("bf16", torch.bfloat16), ("fp32", None)):
        torch.manual_seed(0)
        tr = DSRGTrainer(dev, amp_dtype=amp, seed=0, net=VGG16ASPP(dropout=0.5))
        hist = torch.stack([tr.step(images, labels, cues).detach() for _ in range(60)]).sum(1).cpu().numpy()
        assert np.isfinite(hist).all()
        tot[tag] = hist
        del tr
    for tag, h in tot.items():
        assert h[-1] < 0.85 * h[0], (tag, h[0], h[-1])
    gap = np.abs(tot["bf16"] - tot["fp32"]) / np.abs(tot["fp32"])
    assert gap.max() < 0.02, float(gap.max())                  # (observed at batch 8, 300 steps: 6e-4)


def test_sgd_pack_kernel_equals_torch_fused_sgd_and_the_pack_kernel(ops):
    """dsrg_sgd_pack_f32 on a list of tensors of all the kinds the net has (packed 3x3 / 1x1 implicit-GEMM kernels, a direct-route
    kernel, a kernel that is not packed, biases, a 21 x 1024 classifier — sizes off every multiple of 4096) against
    torch._fused_sgd_ with the same rates per group, two steps (momentum history in play): parameters and history agree to the last
    bit or the one before it (torch's kernel may contract a*b+c), and the packed bf16 forms it leaves are bit for bit those of
    dsrg_pack_conv_weight_f32 on the updated parameter"""
    from dsrg_amd import _lib
    dev_ = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(5)
    cl = torch.channels_last
    shapes = [((256, 128, 3, 3), 0), ((128, 256, 1, 1), 0), ((128, 64, 3, 3), 1), ((64, 3, 3, 3), None), ((256,), None), ((21, 1024, 1, 1), None),
              ((21,), None), ((4099,), None), ((192, 64, 3, 3), 0)] + [((64,), None)] * 12          # 21 tensors: two launches
    mk = lambda sh: (torch.randn(sh, generator=g).to(dev_).contiguous(memory_format=cl) if len(sh) == 4 else torch.randn(sh, generator=g).to(dev_))
    ps = [torch.nn.Parameter(mk(sh)) for sh, _ in shapes]
    odd = torch.randn(4 * 21 + 1, generator=g).to(dev_)
    ps[6] = torch.nn.Parameter(odd[1:22])                             # a slice that is not 16-byte aligned (the scalar path)
    assert ps[6].data_ptr() % 16 == 4
    assert ops.pack_conv_weight_pair(ps[0], True, True)[0] is not ops.pack_conv_weight_pair(ps[0], True, True)[0]   # nothing kept unasked
    class _Owner(object):
        pass
    owner = _Owner()
    ops.keep_weight_packs(ps, owner)
    ref = [p.detach().clone(memory_format=torch.preserve_format) for p in ps]
    bufs, rbufs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    lrs = [1e-2 * (1 + (i % 3)) for i in range(len(ps))]
    wds = [5e-4 if i % 2 == 0 else 0.0 for i in range(len(ps))]
    for i, (p, (sh, plain)) in enumerate(zip(ps, shapes)):                               # the nodes' first forward leaves the packs
        if plain == 0:
            ops.pack_conv_weight_pair(p, True, True)
        elif plain == 1:
            ops.pack_direct_weight_pair(p, True)
    assert sum(1 for p in ps if p.data_ptr() in ops._weight_packs) == 4
    for step in range(2):
        grads = [mk(sh) for sh, _ in shapes]
        for i in (0, 2, 4, 5):                                        # views 4 bytes into a larger buffer, strides kept: what a
            flat = torch.empty(grads[i].numel() + 1, device=dev_)     # DistributedDataParallel bucket hands out
            view = flat[1:].as_strided(grads[i].shape, grads[i].stride())
            view.copy_(grads[i])
            grads[i] = view
            assert view.data_ptr() % 16 == 4 and ops.sgd_pack_eligible(ps[i], view, bufs[i])
        v0 = [p._version for p in ps]
        ops.sgd_pack_step(ps, grads, bufs, lrs, wds, 0.9)
        assert all(p._version > v for p, v in zip(ps, v0))
        for i in range(len(ps)):
            torch._fused_sgd_([ref[i]], [grads[i]], [rbufs[i]], weight_decay=wds[i], momentum=0.9, lr=lrs[i], dampening=0.0,
                              nesterov=False, maximize=False, is_first_step=False)
        for p, r, b, rb in zip(ps, ref, bufs, rbufs):
            assert p.stride() == r.stride()
            assert torch.allclose(p.detach(), r, rtol=1e-6, atol=1e-6) and torch.allclose(b, rb, rtol=1e-6, atol=1e-6)
        for p, (sh, plain) in zip(ps, shapes):
            if plain is None:
                continue
            e = ops._weight_packs[p.data_ptr()]
            assert e.version == p._version
            w = p.detach().clone(memory_format=torch.preserve_format)                    # not a Parameter: packed afresh
            want = ops.pack_conv_weight_pair(w, True, True) if plain == 0 else ops.pack_direct_weight_pair(w, True)
            assert torch.equal(e.fwd, want[0]) and torch.equal(e.dg, want[1])
            got = ops.pack_conv_weight_pair(p, True, True) if plain == 0 else ops.pack_direct_weight_pair(p, True)
            assert got[0] is e.fwd and got[1] is e.dg                                    # the next forward takes them as they are
    # a write the optimizer did not make (load_state_dict, a broadcast): the node packs again
    with torch.no_grad():
        ps[0].mul_(0.5)
    fwd, dgp = ops.pack_conv_weight_pair(ps[0], True, True)
    want = ops.pack_conv_weight_pair(ps[0].detach().clone(memory_format=torch.preserve_format), True, True)
    assert torch.equal(fwd, want[0]) and torch.equal(dgp, want[1])
    # the claim ends with its owner: nothing is kept for the parameter afterwards
    del owner
    assert ops.pack_conv_weight_pair(ps[8], True, True)[0] is not ops.pack_conv_weight_pair(ps[8], True, True)[0]
    # malformed lists are refused
    lib = _lib.lib()
    import ctypes
    arr = (ctypes.c_void_p * 1)(ps[4].data_ptr() + 2)
    one = (ctypes.c_longlong * 1)(8)
    assert lib.dsrg_sgd_pack_f32(1, arr, arr, arr, None, None, None, one, None, None, 0.9, None) != 0     # not aligned to a float


def test_trainer_steps_with_the_sgd_pack_kernel_equal_torch_fused_sgd(ops, monkeypatch):
    """three DSRGTrainer steps with the update + packing kernel (the default) against the same steps with torch._fused_sgd_ and
    the packs made inside each forward: same losses and weights up to the last-bit differences of the two update kernels carried
    through three steps; from the second step on no convolution node packs a kernel (every pack is found fresh)"""
    from dsrg_amd import trainer as T
    from dsrg_amd.backbone import VGG16ASPP
    device = torch.device("cuda", 0)
    images, labels, cues = _batch(4, seed=9)
    out = {}
    for on in (True, False):
        monkeypatch.setattr(T, "_SGD_PACK", on)
        torch.manual_seed(0)                                          # the same initial weights for both
        tr = T.DSRGTrainer(device, seed=0, net=VGG16ASPP(dropout=0.0))
        losses = [tr.step(images, labels, cues).detach().cpu() for _ in range(3)]
        params = [p for g in tr.opt.groups for p in g["params"]]
        kept = [e for e, p in ((ops._weight_packs.get(p.data_ptr()), p) for p in params) if e is not None and e.ref() is p]
        if on:
            assert len(kept) >= 8 and all(e.version == e.ref()._version for e in kept)   # (which layers take a packed route depends on the batch)
        else:
            assert not kept
        out[on] = (torch.stack(losses), [p.detach().clone() for p in params])
        del tr
    assert torch.equal(out[True][0][0], out[False][0][0])             # the first step's forward is the same launches
    assert torch.allclose(out[True][0], out[False][0], rtol=2e-3, atol=1e-6), (out[True][0], out[False][0])
    for a, b in zip(out[True][1], out[False][1]):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5)


def test_trainer_steps_with_deferred_bias_reductions_are_bit_identical(ops, monkeypatch):
    """four DSRGTrainer steps at batch 16 with the bias gradients' finishing passes recorded during backward and run as ONE launch
    behind it (ops.deferred_reductions, the default) against the same steps with every pass launched where it arises: the same device
    code on the same partial rows — losses, weights and momentum buffers bit-equal; the library's list is empty and nothing is kept
    afterwards; and a pass recorded beyond the list's capacity simply runs at once"""
    from dsrg_amd import trainer as T, _lib
    from dsrg_amd.backbone import VGG16ASPP
    device = torch.device("cuda", 0)
    images, labels, cues = _batch(16, seed=11)
    out = {}
    for on in ("1", "0"):
        monkeypatch.setenv("DSRG_DEFER_REDUCTIONS", on)
        torch.manual_seed(0)
        tr = T.DSRGTrainer(device, seed=0, net=VGG16ASPP(dropout=0.0))
        assert tr.defer_bias == (on == "1")
        losses = [tr.step(images, labels, cues).detach().cpu() for _ in range(4)]
        torch.cuda.synchronize()
        out[on] = (torch.stack(losses), tr.weights_checksum().cpu(), [p.detach().clone() for p in tr.net.parameters()])
        assert not ops._defer[0] and not ops._defer_keep
        del tr
    assert torch.equal(out["1"][0], out["0"][0]) and torch.equal(out["1"][1], out["0"][1])
    for a, b in zip(out["1"][2], out["0"][2]):
        assert torch.equal(a, b)
    # the op level: sixty bias gradients inside one block (the list holds 48), all right after the block
    g = torch.randn(2, 64, 9, 11, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    want = ops.bias_grad(g).clone()
    with ops.deferred_reductions():
        got = [ops.bias_grad(g) for _ in range(60)]
    assert all(torch.equal(t, want) for t in got)


def test_bench_gpus_beyond_the_visible_ones_fails_in_one_line():
    """`python bench.py --gpus N` with fewer than N GPUs on the node: no traceback, one line naming the reason"""
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    err = [l for l in r.stderr.splitlines() if l.strip()]
    assert "Traceback" not in r.stderr and err and "only %d GPU(s) visible" % (n - 1) in err[-1], r.stderr[-500:]


def test_config5_retrain_step_resnet101_513():
    """BASELINE configs[4]: RetrainTrainer.step with the full ResNet-101 DeepLab (3,4,23,3) at 513x513, batch 2: the
    65x65 score map, a finite SoftmaxWithLoss (ignore 255) on the 1/8-shrunk label map, parameters move, poly rate"""
    from dsrg_amd.retrain import RetrainTrainer, poly_lr
    device = torch.device("cuda", 0)
    tr = RetrainTrainer(device, backbone="resnet101", seed=0)
    g = torch.Generator().manual_seed(3)
    images = torch.randn(2, 3, 513, 513, generator=g).to(device)
    label = torch.randint(0, 21, (2, 1, 513, 513), generator=g).float()
    label[:, :, :40] = 255.0
    label = label.to(device)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert tr.net(images.contiguous(memory_format=torch.channels_last)).shape == (2, 21, 65, 65)
    params = [p for p in tr.net.parameters() if p.requires_grad]      # (the BatchNorm maps are constants: gamma / beta do not train)
    w0 = [p.detach().clone() for p in params]
    l0 = tr.step(images, label)
    l1 = tr.step(images, label)
    torch.cuda.synchronize()
    assert torch.isfinite(l0) and torch.isfinite(l1) and 2.0 < float(l0) < 4.5          # ~log(21) at initialisation
    moved = sum(int(not torch.equal(a, b.detach())) for a, b in zip(w0, params))
    assert moved == len(w0) and len(w0) == 104 + 4 * 2, (moved, len(w0))
    assert tr.opt.iter == 2 and abs(tr.opt.base_lr - poly_lr(tr.base_lr, 1, tr.max_iter)) < 1e-12
    print("train-f ResNet-101 513x513 losses:", float(l0), float(l1))
