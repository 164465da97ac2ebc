"""Checkpoint I/O: the Flax msgpack container, the diffusers pipeline directory save_model writes and
load_models reads (training_utils.py:177-250, 986-1025), and the optimizer / RNG state file.  Host-side: runs without a GPU.

flax is not installed here, so the byte-level vector below is assembled from the published format (ExtType 1 =
msgpack((shape, dtype.name, bytes))), not captured from flax: the container's parity is unpinned (see checkpoint.py)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from stable_diffusion_training_amd import checkpoint as ck
from stable_diffusion_training_amd import nets, params


def test_flax_msgpack_known_bytes():
    tree = {"w": np.array([1.0, 2.0], dtype=np.float32)}
    payload = bytes([0x93, 0x91, 0x02, 0xA7]) + b"float32" + bytes([0xC4, 0x08]) + np.array([1.0, 2.0], "<f4").tobytes()
    expect = bytes([0x81, 0xA1]) + b"w" + bytes([0xC7, len(payload), 0x01]) + payload
    assert ck.flax_to_bytes(tree) == expect
    back = ck.flax_from_bytes(expect)
    assert back["w"].dtype == np.float32 and back["w"].tolist() == [1.0, 2.0]


def test_flax_msgpack_reader_on_hand_assembled_nested_tree():
    """A nested tree with a bfloat16 leaf, a NumPy scalar (ext type 3) and a chunked leaf, every byte written out from the
    msgpack spec (fixmap 0x8n, fixstr 0xAn, fixarray 0x9n, bin8 0xC4, ext8 0xC7, true 0xC3) and flax's serialization layout
    (ext 1 = ndarray as msgpack((shape, dtype.name, bytes)); ext 3 = NumPy scalar, same payload; arrays over the chunk limit as
    {"__msgpack_chunked_array__": true, "shape": {"0": d0, ...}, "chunks": {"0": flat piece, ...}}) - NOT produced by this
    package's writer, so the reader is checked against bytes it did not write itself.  (Still not flax output: unpinned.)"""
    def fixstr(t):
        assert len(t) < 32
        return bytes([0xA0 | len(t)]) + t.encode()

    bf16_raw = bytes([0x80, 0x3F, 0x20, 0xC0, 0x49, 0x40, 0x00, 0x3F])          # 1.0, -2.5, 3.140625, 0.5 (little endian)
    p_h = bytes([0x93, 0x92, 0x02, 0x02]) + fixstr("bfloat16") + bytes([0xC4, 0x08]) + bf16_raw
    p_n = bytes([0x93, 0x90]) + fixstr("int32") + bytes([0xC4, 0x04, 0x07, 0x00, 0x00, 0x00])
    def f32_vec(vals):
        body = b"".join(np.float32(v).tobytes() for v in vals)
        pay = bytes([0x93, 0x91, len(vals)]) + fixstr("float32") + bytes([0xC4, len(body)]) + body
        return bytes([0xC7, len(pay), 0x01]) + pay

    blk = bytes([0x82]) + fixstr("h") + bytes([0xC7, len(p_h), 0x01]) + p_h + fixstr("n") + bytes([0xC7, len(p_n), 0x03]) + p_n
    big = (bytes([0x83]) + fixstr("__msgpack_chunked_array__") + bytes([0xC3])
           + fixstr("shape") + bytes([0x82]) + fixstr("0") + bytes([0x02]) + fixstr("1") + bytes([0x03])
           + fixstr("chunks") + bytes([0x82]) + fixstr("0") + f32_vec([1, 2, 3]) + fixstr("1") + f32_vec([4, 5, 6]))
    blob = bytes([0x82]) + fixstr("blk") + blk + fixstr("big") + big
    assert len(p_h) == 23 and len(p_n) == 14
    back = ck.flax_from_bytes(blob)
    assert back["blk"]["h"].shape == (2, 2) and back["blk"]["h"].tolist() == [[1.0, -2.5], [3.140625, 0.5]]
    assert back["blk"]["n"] == 7 and back["blk"]["n"].dtype == np.int32
    assert back["big"].dtype == np.float32 and back["big"].tolist() == [[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]


def test_flax_msgpack_round_trip_nested_scalars_bf16_chunks(monkeypatch):
    rng = np.random.default_rng(0)
    tree = {"a": {"kernel": rng.standard_normal((3, 3, 4, 8)).astype(np.float32), "bias": rng.standard_normal(8).astype(np.float32)},
            "codes": rng.integers(-128, 127, (5, 16)).astype(np.int8), "step": np.int32(7), "scale": np.float32(0.5), "z": 1 + 2j,
            "empty": np.zeros((0, 4), np.float32)}
    back = ck.flax_from_bytes(ck.flax_to_bytes(tree))
    assert back["a"]["kernel"].shape == (3, 3, 4, 8) and np.array_equal(back["a"]["kernel"], tree["a"]["kernel"])
    assert np.array_equal(back["a"]["bias"], tree["a"]["bias"]) and np.array_equal(back["codes"], tree["codes"])
    assert back["step"] == 7 and back["step"].dtype == np.int32 and back["scale"] == np.float32(0.5) and back["z"] == 1 + 2j
    assert back["empty"].shape == (0, 4)
    # a bfloat16 leaf (jax writes dtype.name "bfloat16") widens exactly to float32
    import msgpack
    vals = torch.tensor([1.0, -2.5, 3.140625], dtype=torch.bfloat16)
    raw = vals.view(torch.int16).numpy().tobytes()
    blob = msgpack.packb({"h": msgpack.ExtType(1, msgpack.packb(((3,), "bfloat16", raw), use_bin_type=True))})
    assert ck.flax_from_bytes(blob)["h"].tolist() == vals.float().tolist()
    # arrays above MAX_CHUNK_SIZE travel as {"__msgpack_chunked_array__", "shape", "chunks"}
    monkeypatch.setattr(ck, "MAX_CHUNK_SIZE", 64)
    big = {"k": rng.standard_normal((7, 9)).astype(np.float32), "small": np.arange(4, dtype=np.float32)}
    data = ck.flax_to_bytes(big)
    raw_tree = msgpack.unpackb(data, ext_hook=ck._ext_unpack, raw=False)
    assert raw_tree["k"]["__msgpack_chunked_array__"] is True and raw_tree["k"]["shape"] == {"0": 7, "1": 9}
    assert len(raw_tree["k"]["chunks"]) == 4 and isinstance(raw_tree["small"], np.ndarray)
    assert np.array_equal(ck.flax_from_bytes(data)["k"], big["k"])


def test_flax_write_streams_the_same_bytes(tmp_path, monkeypatch):
    rng = np.random.default_rng(1)
    tree = {"big": rng.standard_normal((64, 40)).astype(np.float32), "sub": {"small": np.arange(5, dtype=np.float32), "n": np.int32(3)},
            "i8": rng.integers(-128, 127, (9000,)).astype(np.int8), "wide": rng.standard_normal((300, 300)).astype(np.float32),
            "strided": rng.standard_normal((80, 64)).astype(np.float32)[:, ::2]}
    for limit in (2 ** 30, 20000):  # plain, and with the 360 kB leaf chunked
        monkeypatch.setattr(ck, "MAX_CHUNK_SIZE", limit)
        path = tmp_path / f"t{limit}.msgpack"
        with open(path, "wb") as f:
            ck.flax_write(f, tree)
        assert path.read_bytes() == ck.flax_to_bytes(tree)


def test_flatten_unflatten():
    flat = {"a/b/kernel": 1, "a/b/bias": 2, "c": 3}
    tree = ck.unflatten_tree(flat)
    assert tree == {"a": {"b": {"kernel": 1, "bias": 2}}, "c": 3} and ck.flatten_tree(tree) == flat


def _tiny_models():
    cfgs = {"unet": nets.unet_config("tiny"), "vae": nets.vae_config("tiny"), "text_encoder": nets.clip_config("tiny")}
    w = {"unet": nets.init_params(nets.unet_spec(cfgs["unet"]), seed=1), "vae": nets.init_params(nets.vae_encoder_spec(cfgs["vae"]), seed=2),
         "text_encoder": nets.init_params(nets.clip_text_spec(cfgs["text_encoder"]), seed=3)}
    return cfgs, w


def test_save_model_layout_and_load_models_round_trip(tmp_path, capsys):
    cfgs, w = _tiny_models()
    # the training loop hands over a ParamStore (its masters), an EmaView, or plain trees
    ustore = params.ParamStore(nets.unet_spec(cfgs["unet"]), device="cpu", quantise=False, with_ema=True)
    ustore.load(w["unet"])
    ustore.ema.mul_(0.5)
    out = str(tmp_path / "model@1")
    ck.save_model(cfgs, None, ustore, w["text_encoder"], ck.unflatten_tree(w["vae"]), out)
    assert "model saved" in capsys.readouterr().out
    for rel in ("model_index.json", "unet/config.json", "unet/diffusion_flax_model.msgpack", "vae/config.json",
                "vae/diffusion_flax_model.msgpack", "text_encoder/config.json", "text_encoder/flax_model.msgpack",
                "scheduler/scheduler_config.json"):
        assert os.path.isfile(os.path.join(out, rel)), rel
    assert not [f for _, _, fs in os.walk(out) for f in fs if f.endswith(".tmp")]
    idx = json.load(open(os.path.join(out, "model_index.json")))
    assert idx["_class_name"] == "FlaxStableDiffusionPipeline" and idx["unet"] == ["diffusers", "FlaxUNet2DConditionModel"]
    assert idx["text_encoder"] == ["transformers", "FlaxCLIPTextModel"] and idx["scheduler"] == ["diffusers", "FlaxDDIMScheduler"]
    assert idx["safety_checker"] == [None, None] and idx["feature_extractor"] == [None, None]  # None modules are recorded too
    sch = json.load(open(os.path.join(out, "scheduler", "scheduler_config.json")))  # the reference's placeholder (:998-1004)
    assert (sch["beta_schedule"], sch["prediction_type"], sch["beta_start"], sch["beta_end"], sch["num_train_timesteps"]) == \
        ("scaled_linear", "v_prediction", 0.00085, 0.012, 1000)
    ucfg = json.load(open(os.path.join(out, "unet", "config.json")))
    assert ucfg["_class_name"] == "FlaxUNet2DConditionModel" and ucfg["block_out_channels"] == [32, 64]
    # the weights file is a nested Flax dict
    nested = ck.flax_from_bytes(open(os.path.join(out, "unet", "diffusion_flax_model.msgpack"), "rb").read())
    assert set(nested) >= {"conv_in", "down_blocks_0", "mid_block", "time_embedding"} and nested["conv_in"]["kernel"].shape == (3, 3, 4, 32)

    models = ck.load_models(types.SimpleNamespace(model_path=out))
    assert models["tokenizer"] is None
    assert models["unet"]["config"] == cfgs["unet"] and models["vae"]["config"] == cfgs["vae"]
    assert {k: v for k, v in models["text_encoder"]["config"].items() if k not in ("architectures", "model_type")} == cfgs["text_encoder"]
    for key, name in (("unet", "unet_params"), ("vae", "vae_params"), ("text_encoder", "text_encoder_params")):
        got = models[key][name]
        assert set(got) == set(w[key])
        for p, t in w[key].items():
            assert got[p].dtype == torch.float32 and torch.equal(got[p], t.float()), p
    # the loaded trees feed ParamStore.load directly
    again = params.ParamStore(nets.unet_spec(models["unet"]["config"]), device="cpu", quantise=False)
    again.load(models["unet"]["unet_params"])
    assert torch.equal(again.master, ustore.master)

    # -EMA checkpoint (training.py:281-296): an EmaView serialises the EMA buffer, not the masters
    ck.save_model(cfgs, None, params.EmaView(ustore), w["text_encoder"], w["vae"], str(tmp_path / "model-EMA@1"))
    ema = ck.load_models(types.SimpleNamespace(model_path=str(tmp_path / "model-EMA@1")))["unet"]["unet_params"]
    assert torch.equal(ema["conv_in/kernel"], ustore.p("conv_in/kernel") * 0.5)


def test_load_models_missing_weights_is_loud(tmp_path):
    cfgs, w = _tiny_models()
    out = str(tmp_path / "m")
    ck.save_model(cfgs, None, w["unet"], w["text_encoder"], w["vae"], out)
    os.remove(os.path.join(out, "vae", "diffusion_flax_model.msgpack"))
    with pytest.raises(FileNotFoundError):
        ck.load_models(types.SimpleNamespace(model_path=out))


def test_training_state_round_trip(tmp_path):
    cfgs, w = _tiny_models()

    def make(quant):
        u = params.ParamStore(nets.unet_spec(cfgs["unet"]), device="cpu", quantise=quant, with_ema=True,
                              quant_excluded=("bias", "scale"), block_size=16)
        t = params.ParamStore(nets.clip_text_spec(cfgs["text_encoder"]), device="cpu", quantise=False)
        return u, t

    u, t = make(True)
    g = torch.Generator().manual_seed(5)
    for st in (u, t):
        for name in ("master", "codes", "inv_scale", "mom", "ema"):
            b = getattr(st, name)
            if b is None:
                continue
            if b.dtype == torch.int8:
                b.copy_(torch.randint(-128, 127, b.shape, generator=g, dtype=torch.int8))
            else:
                b.copy_(torch.randn(b.shape, generator=g))
    u.count, t.count = 41, 41
    rng = torch.Generator().manual_seed(99)
    torch.randn(10, generator=rng)
    path = str(tmp_path / "state.safetensors")
    ck.save_training_state(path, u, t, rng)
    expect_next = torch.randn(4, generator=rng)

    u2, t2 = make(True)
    rng2 = ck.load_training_state(path, u2, t2, torch.Generator())
    for a, b in ((u, u2), (t, t2)):
        assert a.count == b.count == 41
        for name in ("master", "codes", "inv_scale", "mom", "ema"):
            x, y = getattr(a, name), getattr(b, name)
            assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), name
    assert torch.equal(torch.randn(4, generator=rng2), expect_next)

    u3, t3 = make(False)  # other quantisation setting -> other buffer layout
    with pytest.raises(ValueError):
        ck.load_training_state(path, u3, t3)


def test_training_state_keeps_one_generator_per_rank(tmp_path):
    """Data-parallel resume: every rank draws its own noise / timestep stream (examples/train_synthetic.py seeds
    master_seed * 1009 + rank), so the state file holds ALL ranks' generator states and rank r gets state r back; a file written
    by another world size is refused."""
    spec = [("a/kernel", (16, 16)), ("a/bias", (16,))]
    u = params.ParamStore(spec, device="cpu", quantise=True, quant_excluded=("bias",), block_size=16)
    t = params.ParamStore(spec, device="cpu", quantise=False)
    gens = [torch.Generator().manual_seed(5 * 1009 + r) for r in range(2)]
    for g in gens:
        torch.randn(7, generator=g)
    path = str(tmp_path / "state2.safetensors")
    ck.save_training_state(path, u, t, rng_states=[g.get_state() for g in gens])
    expect = [torch.randn(3, generator=g) for g in gens]
    assert not torch.equal(expect[0], expect[1])
    for r in range(2):
        got = ck.load_training_state(path, u, t, torch.Generator(), rank=r, world=2)
        assert torch.equal(torch.randn(3, generator=got), expect[r]), f"rank {r} resumed another rank's generator"
    with pytest.raises(ValueError):
        ck.load_training_state(path, u, t, torch.Generator(), rank=0, world=1)
    with pytest.raises(ValueError):
        ck.load_training_state(path, u, t, torch.Generator(), rank=0, world=4)
    assert len(ck.gather_rng_states(gens[0])) == 1  # no process group: this process's state alone


def test_training_state_errors_name_their_cause(tmp_path):
    """a round-1 file (format -1: other leaf alignment / grouping) is refused with a message that names the format; a
    file saved without generator state says so instead of 'generator states of 0 rank(s)'."""
    from safetensors.torch import save_file
    spec = [("a/kernel", (16, 16)), ("a/bias", (16,))]
    u = params.ParamStore(spec, device="cpu", quantise=True, quant_excluded=("bias",), block_size=16)
    t = params.ParamStore(spec, device="cpu", quantise=False)
    old = str(tmp_path / "old.safetensors")
    save_file({"unet.master": torch.zeros(4)}, old, metadata={"format": "sdt-training-state-1"})
    with pytest.raises(ValueError, match="sdt-training-state-1.*sdt-training-state-2"):
        ck.load_training_state(old, u, t)
    path = str(tmp_path / "norng.safetensors")
    ck.save_training_state(path, u, t)  # no generator
    ck.load_training_state(path, u, t)  # optimizer state alone: fine
    with pytest.raises(ValueError, match="no sampling-generator state"):
        ck.load_training_state(path, u, t, torch.Generator())
