"""Checkpoint I/O around train_step: the diffusers / transformers Flax directory layout the
reference reads (`load_models`, training_utils.py:177-250) and writes (`save_model`, training_utils.py:986-1025, called
from training.py:151-184 and :262-299), plus what the reference lacks: the optimizer state (8-bit Lion codes + scales,
fp32 momenta, EMA, step count) and the sampling RNG, so a run can resume bit-exactly.

The reference delegates the file formats to third-party code that is not vendored under /root/reference:
  * flax.serialization.to_bytes / from_bytes (flax 0.7.x, pinned by diffusers 0.21.4 / requirements.txt): msgpack of the
    nested parameter dict; an ndarray is ExtType(1, msgpack((shape, dtype.name, C-order bytes))), a numpy scalar ExtType(3,
    same payload), a python complex ExtType(2, msgpack((re, im))); arrays above 2**30 bytes are replaced by
    {"__msgpack_chunked_array__": True, "shape": {"0": ..}, "chunks": {"0": ..}}.  Restated here from that published
    format (flax is not installed in this image: the byte-level known-answer test in tests/test_checkpoint_cpu.py is built
    from the format description, not from flax output - "parity unpinned" for this file format).
  * diffusers FlaxModelMixin.save_pretrained / FlaxDiffusionPipeline.save_pretrained: <sub>/config.json +
    <sub>/diffusion_flax_model.msgpack, model_index.json; transformers FlaxPreTrainedModel: config.json + flax_model.msgpack.

Host-side code: parameters cross PCIe once per save (the reference's save forces the same device->host transfer).
"""
import hashlib
import json
import os

import msgpack
import numpy as np
import torch

from . import nets
from .params import EmaView, ParamStore

DIFFUSERS_VERSION = "0.21.4"  # the version the reference pins
MAX_CHUNK_SIZE = 2 ** 30      # flax.serialization.MAX_CHUNK_SIZE
_EXT_NDARRAY, _EXT_NATIVE_COMPLEX, _EXT_NPSCALAR = 1, 2, 3

UNET_WEIGHTS = "diffusion_flax_model.msgpack"   # diffusers FLAX_WEIGHTS_NAME
CLIP_WEIGHTS = "flax_model.msgpack"             # transformers FLAX_WEIGHTS_NAME


# ----------------------------------------------------------------------------- flax msgpack
def _ndarray_payload(arr):
    arr = np.asarray(arr)
    if arr.dtype.hasobject:
        raise ValueError("object arrays cannot be serialised")
    return msgpack.packb((arr.shape, arr.dtype.name, arr.tobytes("C")), use_bin_type=True)


def _ext_pack(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_payload(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _ndarray_payload(x))
    if isinstance(x, complex):
        return msgpack.ExtType(_EXT_NATIVE_COMPLEX, msgpack.packb((x.real, x.imag)))
    return x


def _ndarray_from_payload(data):
    shape, dtype_name, buf = msgpack.unpackb(data, raw=True)
    name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    if name == "bfloat16":  # numpy has no bfloat16: widen to float32 (exact)
        u = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
        return u.view(np.float32).reshape(shape)
    return np.frombuffer(buf, dtype=np.dtype(name)).reshape(shape, order="C")


def _ext_unpack(code, data):
    if code == _EXT_NDARRAY:
        return _ndarray_from_payload(data)
    if code == _EXT_NPSCALAR:
        return _ndarray_from_payload(data)[()]
    if code == _EXT_NATIVE_COMPLEX:
        re, im = msgpack.unpackb(data)
        return complex(re, im)
    return msgpack.ExtType(code, data)


def _chunk_leaves(tree):
    if isinstance(tree, dict):
        return {k: _chunk_leaves(v) for k, v in tree.items()}
    if isinstance(tree, np.ndarray) and tree.size * tree.dtype.itemsize > MAX_CHUNK_SIZE:
        per = max(1, MAX_CHUNK_SIZE // tree.dtype.itemsize)
        flat = tree.reshape(-1)
        return {"__msgpack_chunked_array__": True, "shape": {str(i): int(d) for i, d in enumerate(tree.shape)},
                "chunks": {str(i): flat[o: o + per] for i, o in enumerate(range(0, flat.size, per))}}
    return tree


def _unchunk_leaves(tree):
    if not isinstance(tree, dict):
        return tree
    if "__msgpack_chunked_array__" in tree:
        shape = tuple(tree["shape"][str(i)] for i in range(len(tree["shape"])))
        return np.concatenate([tree["chunks"][str(i)] for i in range(len(tree["chunks"]))]).reshape(shape)
    return {k: _unchunk_leaves(v) for k, v in tree.items()}


def flax_to_bytes(tree):
    """flax.serialization.to_bytes of a nested dict of numpy arrays / scalars."""
    return msgpack.packb(_chunk_leaves(tree), default=_ext_pack, strict_types=True)


def _bin_header(n):
    if n < 256:
        return bytes([0xC4, n])
    if n < 65536:
        return bytes([0xC5]) + n.to_bytes(2, "big")
    return bytes([0xC6]) + n.to_bytes(4, "big")


def _ext_header(n, code):
    if n in (1, 2, 4, 8, 16):
        return bytes([{1: 0xD4, 2: 0xD5, 4: 0xD6, 8: 0xD7, 16: 0xD8}[n], code])
    if n < 256:
        return bytes([0xC7, n, code])
    if n < 65536:
        return bytes([0xC8]) + n.to_bytes(2, "big") + bytes([code])
    return bytes([0xC9]) + n.to_bytes(4, "big") + bytes([code])


def flax_write(f, tree):
    """flax_to_bytes(tree) streamed into the binary file f, byte for byte the same output, without materialising it: array
    payloads go from the numpy buffer straight to the file (a 3.4 GB UNet would otherwise be copied three times)."""
    packer = msgpack.Packer(default=_ext_pack, strict_types=True, autoreset=True)

    def emit(node):
        if isinstance(node, dict):
            f.write(packer.pack_map_header(len(node)))
            for k, v in node.items():
                f.write(packer.pack(k))
                emit(v)
        elif isinstance(node, np.ndarray) and not node.dtype.hasobject and node.nbytes >= 4096:
            arr = np.ascontiguousarray(node)
            prefix = bytes([0x93]) + msgpack.packb(arr.shape) + msgpack.packb(arr.dtype.name) + _bin_header(arr.nbytes)
            f.write(_ext_header(len(prefix) + arr.nbytes, _EXT_NDARRAY))
            f.write(prefix)
            f.write(memoryview(arr).cast("B"))
        else:
            f.write(packer.pack(node))

    emit(_chunk_leaves(tree))


def flax_from_bytes(data):
    """flax.serialization.msgpack_restore: nested dict of numpy arrays."""
    return _unchunk_leaves(msgpack.unpackb(data, ext_hook=_ext_unpack, raw=False, strict_map_key=False))


# ----------------------------------------------------------------------------- trees
def flatten_tree(tree, prefix=""):
    """Nested Flax parameter dict -> {"a/b/kernel": leaf} (the path form ParamStore.load and create_mask use)."""
    out = {}
    for k, v in tree.items():
        p = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten_tree(v, p))
        else:
            out[p] = v
    return out


def unflatten_tree(flat):
    out = {}
    for p, v in flat.items():
        node = out
        parts = p.split("/")
        for k in parts[:-1]:
            node = node.setdefault(k, {})
        node[parts[-1]] = v
    return out


def _to_numpy(v):
    if torch.is_tensor(v):
        v = v.detach()
        if v.dtype == torch.bfloat16:
            v = v.float()
        return v.cpu().numpy()
    return np.asarray(v)


def params_to_tree(params):
    """Whatever the training loop holds as `params` -> nested dict of host numpy arrays: a ParamStore / TrainState (its fp32
    master), an EmaView (the EMA buffer), or a flat / nested dict of tensors."""
    if hasattr(params, "store") and isinstance(params.store, ParamStore) and not isinstance(params, EmaView):
        params = params.store
    if isinstance(params, ParamStore) and getattr(params, "full_tree", None) is not None:
        flat = flatten_tree(params.full_tree)  # frozen VAE: the device store holds the encoder half only
    elif isinstance(params, EmaView):
        flat = params.store.export_host("ema")
    elif isinstance(params, ParamStore):
        flat = params.export_host("master")
    elif isinstance(params, dict):
        flat = flatten_tree(params)
    else:
        raise TypeError(f"cannot serialise parameters of type {type(params).__name__}")
    return unflatten_tree({p: _to_numpy(v) for p, v in flat.items()})


# ----------------------------------------------------------------------------- configs
def _jsonable(cfg):
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}


def _write_json(path, obj):
    with open(path, "w") as f:
        json.dump(obj, f, indent=2, sort_keys=True)
        f.write("\n")


def _read_config(path, defaults=None):
    with open(path) as f:
        raw = json.load(f)
    cfg = dict(defaults or {})
    for k, v in raw.items():
        if not k.startswith("_"):
            cfg[k] = tuple(v) if isinstance(v, list) else v
    return cfg


def _write_submodel(out_dir, sub, config, header, weights_name, params):
    d = os.path.join(out_dir, sub)
    os.makedirs(d, exist_ok=True)
    cfg = dict(header)
    cfg.update(_jsonable(config))
    _write_json(os.path.join(d, "config.json"), cfg)
    tmp = os.path.join(d, weights_name + ".tmp")
    with open(tmp, "wb") as f:
        flax_write(f, params_to_tree(params))
    os.replace(tmp, os.path.join(d, weights_name))  # a crash mid-write never leaves a truncated checkpoint under the final name


def _model_config(model_object_dict, key):
    m = model_object_dict[key]
    if isinstance(m, dict) and "config" in m and isinstance(m["config"], dict):
        return m["config"]
    if isinstance(m, dict):
        return m
    return dict(getattr(m, "config"))


# ----------------------------------------------------------------------------- the reference's two entry points
def save_model(model_object_dict, tokenizer_object, unet_params, text_encoder_params, vae_params, output_dir):
    """training_utils.py:986-1025: write a FlaxStableDiffusionPipeline directory.  The scheduler entry is the reference's
    hard-coded DDIM placeholder (scaled_linear, v_prediction, :998-1004), whatever the run trained with."""
    os.makedirs(output_dir, exist_ok=True)
    dv = {"_diffusers_version": DIFFUSERS_VERSION}
    _write_submodel(output_dir, "unet", _model_config(model_object_dict, "unet"),
                    {"_class_name": "FlaxUNet2DConditionModel", **dv}, UNET_WEIGHTS, unet_params)
    _write_submodel(output_dir, "vae", _model_config(model_object_dict, "vae"),
                    {"_class_name": "FlaxAutoencoderKL", **dv}, UNET_WEIGHTS, vae_params)
    _write_submodel(output_dir, "text_encoder", _model_config(model_object_dict, "text_encoder"),
                    {"architectures": ["CLIPTextModel"], "model_type": "clip_text_model"}, CLIP_WEIGHTS, text_encoder_params)
    os.makedirs(os.path.join(output_dir, "scheduler"), exist_ok=True)
    _write_json(os.path.join(output_dir, "scheduler", "scheduler_config.json"),
                {"_class_name": "FlaxDDIMScheduler", **dv, "beta_start": 0.00085, "beta_end": 0.012,
                 "beta_schedule": "scaled_linear", "num_train_timesteps": 1000, "prediction_type": "v_prediction",
                 "set_alpha_to_one": True, "steps_offset": 0, "trained_betas": None})
    index = {"_class_name": "FlaxStableDiffusionPipeline", **dv,
             "scheduler": ["diffusers", "FlaxDDIMScheduler"], "text_encoder": ["transformers", "FlaxCLIPTextModel"],
             "tokenizer": ["transformers", "CLIPTokenizer"], "unet": ["diffusers", "FlaxUNet2DConditionModel"],
             "vae": ["diffusers", "FlaxAutoencoderKL"],
             # save_pretrained records every registered module, None ones as [null, null]; from_pretrained expects the entries
             "safety_checker": [None, None], "feature_extractor": [None, None]}
    if tokenizer_object is not None:
        tokenizer_object.save_pretrained(os.path.join(output_dir, "tokenizer"))
    _write_json(os.path.join(output_dir, "model_index.json"), index)
    print("f model saved ")  # the reference's message, verbatim (training_utils.py:1025)


def load_models(training_config, load_tokenizer=True):
    """training_utils.py:177-250: read the pipeline directory at training_config.model_path.  Returns the dict
    on_device_model_training_state() takes: {"unet": {"unet_params", "config"}, "vae": {...}, "text_encoder": {...},
    "tokenizer": CLIPTokenizer | None}; parameter trees are flat {"path/with/slashes": float32 tensor} on the host (the
    reference loads fp32 masters and computes in bf16, :209-222)."""
    root = training_config.model_path

    def weights(sub, name):
        path = os.path.join(root, sub, name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: expected Flax weights (the reference loads Flax checkpoints, training_utils.py:209-222)")
        with open(path, "rb") as f:
            flat = flatten_tree(flax_from_bytes(f.read()))
        return {p: torch.from_numpy(np.array(v, dtype=np.float32)) for p, v in flat.items()}  # a writable fp32 copy

    unet_cfg = _read_config(os.path.join(root, "unet", "config.json"), nets._UNET_DEFAULTS)
    vae_cfg = _read_config(os.path.join(root, "vae", "config.json"))
    clip_cfg = _read_config(os.path.join(root, "text_encoder", "config.json"))
    tokenizer = None
    if load_tokenizer and os.path.isdir(os.path.join(root, "tokenizer")):
        from transformers import CLIPTokenizer
        tokenizer = CLIPTokenizer.from_pretrained(root, subfolder="tokenizer")
    return {"unet": {"unet_params": weights("unet", UNET_WEIGHTS), "config": unet_cfg},
            "vae": {"vae_params": weights("vae", UNET_WEIGHTS), "config": vae_cfg},
            "text_encoder": {"text_encoder_params": weights("text_encoder", CLIP_WEIGHTS), "config": clip_cfg},
            "tokenizer": tokenizer}


# ----------------------------------------------------------------------------- optimizer / RNG state (not in the reference)
def _layout_digest(store):
    h = hashlib.sha256()
    for p in store.order:
        lf = store.leaves[p]
        h.update(f"{p}:{lf.shape}:{lf.offset}:{int(lf.quantised)}:{int(lf.decayed)};".encode())
    h.update(f"bs={store.block_size};total={store.total}".encode())
    return h.hexdigest()


_STATE_BUFFERS = ("master", "codes", "inv_scale", "mom", "ema")
# -2: flat-buffer layout of round 2 (8-element leaf alignment, segment ends at multiples of 2048, time_emb_proj / cross-attention
# to_k / to_v grouped per width).  -1 files (round 1) hold the same tensors at other offsets and cannot be re-dealt.
STATE_FORMAT = "sdt-training-state-2"
_OLD_FORMATS = {"sdt-training-state-1": "the flat-buffer layout changed after that format (leaf alignment, grouped projection leaves); "
                                        "re-export the weights with save_model from the build that wrote the file and restart the optimizer state"}


def save_training_state(path, unet_state, text_encoder_state, train_rng=None, rng_states=None):
    """Everything train_step mutates, so that load_training_state + the same batches continue the run: fp32 masters, 8-bit
    Lion codes + per-block scales, fp32 momenta of the unquantised leaves, EMA, step counts, and the sampling generator(s).
    Data-parallel runs draw different noise / timesteps on every rank: pass rng_states = the list of ALL ranks' generator states
    (gather_rng_states) so that each rank resumes its own stream; train_rng alone stores this process's generator (world 1)."""
    from safetensors.torch import save_file
    tensors, meta = {}, {"format": STATE_FORMAT}
    for name, st in (("unet", unet_state), ("text_encoder", text_encoder_state)):
        store = st.store if hasattr(st, "store") else st
        store._gather()  # sharded optimizer: raises unless GradReducer.gather_state() (collective, all ranks) made the state whole
        for b in _STATE_BUFFERS:
            t = getattr(store, b)
            if t is not None:
                tensors[f"{name}.{b}"] = t.detach().cpu().contiguous()
        meta[f"{name}.count"] = str(int(store.count))
        meta[f"{name}.layout"] = _layout_digest(store)
    if rng_states is not None:
        meta["train_rng.world"] = str(len(rng_states))
        for r, stt in enumerate(rng_states):
            tensors[f"train_rng.state.{r}"] = stt.cpu()
    elif train_rng is not None:
        meta["train_rng.world"] = "1"
        tensors["train_rng.state.0"] = train_rng.get_state().cpu()
    tmp = path + ".tmp"
    save_file(tensors, tmp, metadata=meta)
    os.replace(tmp, path)


def gather_rng_states(train_rng):
    """All ranks' generator states, in rank order, on every rank (a collective when torch.distributed is initialised)."""
    import torch
    import torch.distributed as dist
    mine = train_rng.get_state().cpu()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [mine]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, mine)
    return out


def load_training_state(path, unet_state, text_encoder_state, train_rng=None, rank=0, world=1):
    """Inverse of save_training_state, into states built for the same model / quantisation settings (checked by a digest of
    the buffer layout).  Rank `rank` of `world` restores ITS generator; a file written by a different world size is refused
    (the ranks' noise streams cannot be re-dealt).  Returns the generator (state restored in place when given).
    With the sharded optimizer every rank calls this with the same file (each loads the whole buffers; the store is whole afterwards);
    a pending all-gather of the weight mirrors is drained first (ParamStore.begin_external_write)."""
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        fmt = meta.get("format")
        if fmt in _OLD_FORMATS:
            raise ValueError(f"{path}: training-state format {fmt}, this build reads {STATE_FORMAT}: {_OLD_FORMATS[fmt]}")
        if fmt != STATE_FORMAT:
            raise ValueError(f"{path}: not a training-state file (format {fmt!r}, expected {STATE_FORMAT})")
        keys = set(f.keys())
        for name, st in (("unet", unet_state), ("text_encoder", text_encoder_state)):
            store = st.store if hasattr(st, "store") else st
            if meta.get(f"{name}.layout") != _layout_digest(store):
                raise ValueError(f"{path}: {name} state was saved for a different parameter layout / quantisation setting "
                                 f"(layout version {STATE_FORMAT}; digest {meta.get(f'{name}.layout', '?')[:12]} != {_layout_digest(store)[:12]}: "
                                 "model config, quantisation / weight-decay exclusion lists and quant_block_size must match)")
            store.begin_external_write()  # sharded optimizer: a mirror all-gather of the last step may still be in flight
            for b in _STATE_BUFFERS:
                dst = getattr(store, b)
                if (dst is not None) != (f"{name}.{b}" in keys):
                    raise ValueError(f"{path}: {name}.{b} present in only one of file / state")
                if dst is not None:
                    dst.copy_(f.get_tensor(f"{name}.{b}"))
            store.count = int(meta[f"{name}.count"])
            if store.device.type == "cuda":
                store.prepare(full=True)  # the masters changed under the bf16 compute copies
            store.state_whole = True      # every rank loaded the whole buffers (sharded optimizer: nothing to gather before a save)
        if train_rng is not None:
            saved_world = int(meta.get("train_rng.world", "0"))
            if saved_world == 0 and "train_rng.state" in keys:  # files of the first format: one generator
                saved_world, key = 1, "train_rng.state"
            else:
                key = f"train_rng.state.{rank}"
            if saved_world == 0:
                raise ValueError(f"{path}: holds no sampling-generator state (saved without train_rng / rng_states): pass train_rng=None to "
                                 "load the optimizer state alone")
            if saved_world != world:
                raise ValueError(f"{path}: generator states of {saved_world} rank(s), this run has {world}: every rank draws its own "
                                 "noise / timestep stream and the streams cannot be re-dealt")
            if key not in keys:
                raise ValueError(f"{path}: no generator state saved for rank {rank}")
            train_rng.set_state(f.get_tensor(key))
    return train_rng
