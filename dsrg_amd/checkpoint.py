"""Weights in, snapshots out: what `tools/train.py --weights X.caffemodel` / `--snapshot` and the solver's
`snapshot: 8000`, `snapshot_prefix: "models/model-s"` do in the reference (training/tools/train.py:53-63,
training/experiment/seed_mc/run.sh:5,9, solver-s.prototxt:16-17, solver-f.prototxt:15-16).

  read_caffemodel / write_caffemodel   a NetParameter file, by name -> [blobs]; a minimal protobuf wire-format codec
                                       (no caffe, no caffe.proto needed): NetParameter.layer = 100 (LayerParameter:
                                       name = 1, blobs = 7) and the legacy NetParameter.layers = 2 (V1LayerParameter:
                                       name = 4, blobs = 6); BlobProto: shape = 7 (dim = 1), data = 5, double_data = 8,
                                       legacy num/channels/height/width = 1..4
  load_weights(net, path)              Caffe's copy-by-layer-name: layers present in both are copied, shapes must match,
                                       the rest keep their initialisation (so vgg16_20M_mc.caffemodel fills conv1_1 ...
                                       fc7_k and leaves the fc8-SEC_k heads at N(0, 0.01), and stage 2 starts from
                                       models/model-s_iter_8000.caffemodel); also .npz ("name/0", "name/1") and torch files
  save_weights(net, path)              the same formats, chosen by extension
  trainer snapshots                    save_snapshot / load_snapshot: weights + momentum history + iteration (Caffe's
                                       .solverstate), rank 0 writes under DDP
"""
import os
import struct

import numpy as np
import torch


# ------------------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """yield (field number, wire type, value) over one message; length-delimited values as memoryviews"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, val


def _parse_blob(buf):
    shape, legacy, chunks, dchunks, scalars = None, {}, [], [], []
    for field, wt, val in _fields(buf):
        if field == 7 and wt == 2:                                 # BlobShape { repeated int64 dim = 1 [packed] }
            dims = []
            for f2, wt2, v2 in _fields(val):
                if f2 == 1 and wt2 == 2:
                    p = 0
                    while p < len(v2):
                        d, p = _varint(v2, p)
                        dims.append(d)
                elif f2 == 1 and wt2 == 0:
                    dims.append(v2)
            shape = tuple(dims)
        elif field == 5 and wt == 2:
            chunks.append(np.frombuffer(val, dtype="<f4"))
        elif field == 5 and wt == 5:
            scalars.append(struct.unpack("<f", bytes(val))[0])
        elif field == 8 and wt == 2:
            dchunks.append(np.frombuffer(val, dtype="<f8"))
        elif field in (1, 2, 3, 4) and wt == 0:
            legacy[field] = val
    if chunks or scalars:
        data = np.concatenate(chunks + ([np.asarray(scalars, np.float32)] if scalars else []))
    elif dchunks:
        data = np.concatenate(dchunks).astype(np.float32)
    else:
        data = np.zeros(0, np.float32)
    if shape is None:
        shape = tuple(legacy.get(k, 1) for k in (1, 2, 3, 4)) if legacy else (data.size,)
    return np.array(data, dtype=np.float32).reshape(shape)


def read_caffemodel(path):
    """-> {layer name: [numpy blobs]} for every layer that carries blobs"""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    out = {}
    for field, wt, val in _fields(buf):
        if wt != 2 or field not in (100, 2):
            continue
        name_field, blob_field = (1, 7) if field == 100 else (4, 6)
        name, blobs = None, []
        for f2, wt2, v2 in _fields(val):
            if f2 == name_field and wt2 == 2:
                name = bytes(v2).decode("utf-8")
            elif f2 == blob_field and wt2 == 2:
                blobs.append(_parse_blob(v2))
        if name is not None and blobs:
            out[name] = blobs
    return out


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_ld(field, payload):
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def write_caffemodel(path, layers, net_name="DSRG"):
    """layers: ordered {layer name: [numpy blobs]} -> NetParameter { name, layer { name, type, blobs { shape, data } } }"""
    msg = bytearray(_enc_ld(1, net_name.encode("utf-8")))
    for name, blobs in layers.items():
        lay = bytearray(_enc_ld(1, name.encode("utf-8")) + _enc_ld(2, b"Convolution"))
        for b in blobs:
            a = np.ascontiguousarray(b, dtype="<f4")
            shape = _enc_ld(1, b"".join(_enc_varint(int(d)) for d in a.shape))
            lay += _enc_ld(7, _enc_ld(7, shape) + _enc_ld(5, a.tobytes()))
        msg += _enc_ld(100, bytes(lay))
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(bytes(msg))
    os.replace(tmp, path)


# ------------------------------------------------------------------------------------------ name mapping
def caffe_layer_map(net):
    """{Caffe layer name: module} for the nets of this package: VGG16ASPP -> conv1_1 ... conv5_3, fc6_k, fc7_k,
    fc8-SEC_k (train-s.prototxt:44-714; train-f.prototxt uses the same names with fc8-SEC -> fc8_k heads handled by
    `rename`); any other module with parameters -> its dotted torch name."""
    from .backbone import VGG16ASPP
    out = {}
    if isinstance(net, VGG16ASPP):
        vgg = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3",
               "conv5_1", "conv5_2", "conv5_3"]
        convs = [m for m in net.features if isinstance(m, torch.nn.Conv2d)]
        assert len(convs) == len(vgg)
        out.update(zip(vgg, convs))
        for k, br in enumerate(net.branches):
            out["fc6_%d" % (k + 1)], out["fc7_%d" % (k + 1)], out["fc8-SEC_%d" % (k + 1)] = br[0], br[3], br[6]
        return out
    for name, m in net.named_modules():
        if any(True for _ in m.parameters(recurse=False)):
            out[name] = m
    return out


def _module_blobs(m):
    """Caffe blob order: weight, bias; frozen BN: (weight, bias, running_mean, running_var) under the module's name"""
    names = [n for n in ("weight", "bias", "running_mean", "running_var") if getattr(m, n, None) is not None]
    return [(n, getattr(m, n)) for n in names]


def net_to_layers(net):
    return {name: [t.detach().float().cpu().numpy() for _, t in _module_blobs(m)] for name, m in caffe_layer_map(net).items()}


@torch.no_grad()
def load_layers(net, layers, rename=None, strict_shapes=True):
    """Caffe's Net::CopyTrainedLayersFrom: copy by layer name, ignore source layers the net lacks, fail on a shape
    mismatch.  `rename` maps source names to target names (e.g. {"fc8-SEC_1": "fc8_1"}).  -> names copied"""
    copied = []
    lm = caffe_layer_map(net)
    for src, blobs in layers.items():
        name = (rename or {}).get(src, src)
        m = lm.get(name)
        if m is None:
            continue
        targets = _module_blobs(m)
        if len(blobs) > len(targets):
            raise ValueError("layer %s: %d source blobs for %d parameters" % (name, len(blobs), len(targets)))
        for (pname, t), b in zip(targets, blobs):
            b = np.asarray(b)
            if b.size != t.numel() or (strict_shapes and b.ndim == t.dim() and tuple(b.shape) != tuple(t.shape)):
                raise ValueError("layer %s.%s: source shape %s does not match %s" % (name, pname, b.shape, tuple(t.shape)))
            t.copy_(torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).reshape(t.shape).to(t.device, t.dtype))
        copied.append(name)
    return copied


def _report_copied(net, path, copied, n_source):
    """Caffe logs "Ignoring source layer ..." and starts from the filler values silently only for layers it names; a file that
    fills NOTHING is almost certainly the wrong file (or names layers the other framework's way) — refuse it, and say which
    target layers stayed at their initial values otherwise."""
    import warnings
    if not copied:
        raise ValueError("%s: none of its %d layers matches a layer of %s by name — training would start from the random "
                         "initialisation (for nets other than VGG16ASPP the layer names are the torch module paths)" % (
                             path, n_source, type(net).__name__))
    missing = sorted(set(caffe_layer_map(net)) - set(copied))
    if missing:
        warnings.warn("%s fills %d layers of %s; %d keep their initial values: %s" % (
            path, len(copied), type(net).__name__, len(missing), ", ".join(missing[:12]) + (" ..." if len(missing) > 12 else "")))
    return copied


def load_weights(net, path, rename=None):
    """--weights: .caffemodel (by layer name), .npz ("<layer>/<blob index>"), or a torch file (state_dict or a snapshot
    written by save_snapshot; loaded non-strictly by key, like Caffe by name).  -> names copied.  A file that matches no
    layer raises; layers left at their initial values are reported by a warning."""
    try:
        return _load_weights(net, path, rename)
    finally:
        if any(p.is_cuda for p in net.parameters()):         # kept bf16 kernels of the old values (ops.keep_weight_packs)
            from .ops import forget_weight_packs
            forget_weight_packs(net.parameters())


def _load_weights(net, path, rename=None):
    if path.endswith(".caffemodel"):
        layers = read_caffemodel(path)
        return _report_copied(net, path, load_layers(net, layers, rename), len(layers))
    if path.endswith(".npz"):
        z = np.load(path)
        layers = {}
        for key in z.files:
            name, idx = key.rsplit("/", 1)
            layers.setdefault(name, {})[int(idx)] = z[key]
        return _report_copied(net, path, load_layers(net, {n: [d[i] for i in sorted(d)] for n, d in layers.items()}, rename),
                              len(layers))
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = sd.get("net", sd)
    own = net.state_dict()
    use = {k: v for k, v in sd.items() if k in own}
    for k, v in use.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise ValueError("%s: source shape %s does not match %s" % (k, tuple(v.shape), tuple(own[k].shape)))
    if not use:
        raise ValueError("%s: none of its %d tensors matches a parameter of %s by name" % (path, len(sd), type(net).__name__))
    net.load_state_dict(use, strict=False)
    return sorted(use)


def save_weights(net, path):
    if path.endswith(".caffemodel"):
        write_caffemodel(path, net_to_layers(net))
    elif path.endswith(".npz"):
        np.savez(path, **{"%s/%d" % (n, i): b for n, blobs in net_to_layers(net).items() for i, b in enumerate(blobs)})
    else:
        torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, path)


# ------------------------------------------------------------------------------------------ solver snapshots
def _is_rank0():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_snapshot(trainer, prefix):
    """Caffe's Solver::Snapshot: `<prefix>_iter_<N>.caffemodel` (weights; for VGG16ASPP with the reference's layer names and
    blob order, i.e. readable by the reference's tools — other nets are written with their torch module paths as layer names
    and are readable by load_weights only) + `<prefix>_iter_<N>.solverstate.pt` (weights, momentum history, iteration).
    COLLECTIVE under torch.distributed: every rank calls it (rank 0 writes, the others wait) and every rank returns after the
    files exist — or every rank raises when rank 0 could not write them (its outcome is broadcast; a rank-0 failure never
    leaves the others blocked in a barrier).  Calling it from rank 0 only (`if rank == 0: trainer.save()`) would hang rank 0 in
    that broadcast: save on all ranks.  -> (caffemodel path, solverstate path)"""
    import torch.distributed as dist
    it = trainer.opt.iter
    model_path, state_path = "%s_iter_%d.caffemodel" % (prefix, it), "%s_iter_%d.solverstate.pt" % (prefix, it)
    err = None
    if _is_rank0():
        try:
            d = os.path.dirname(model_path)
            if d:
                os.makedirs(d, exist_ok=True)
            save_weights(trainer.net, model_path)
            tmp = state_path + ".tmp"
            torch.save({"net": {k: v.detach().cpu() for k, v in trainer.net.state_dict().items()},
                        "opt": trainer.opt.state_dict(), "iter": it}, tmp)
            os.replace(tmp, state_path)
        except Exception as e:                           # noqa: BLE001 - reported on every rank below
            err = "%s: %s" % (type(e).__name__, e)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        box = [err]
        dist.broadcast_object_list(box, src=0)          # also the barrier: a rank that loads the snapshot right away finds it
        err = box[0]
    if err is not None:
        raise RuntimeError("snapshot %s not written (rank 0: %s)" % (state_path, err))
    return model_path, state_path


def load_snapshot(trainer, state_path):
    """Solver::Restore (train.py:57-58): weights, momentum history and the iteration counter"""
    st = torch.load(state_path, map_location="cpu", weights_only=True)
    trainer.net.load_state_dict(st["net"])
    trainer.opt.load_state_dict(st["opt"])
    return st["iter"]
