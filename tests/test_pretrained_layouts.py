"""`load_unet` reads what the reference reads (/root/reference/ddpo/utils/serialization.py:322-371 hands `pretrained_model` to
`FlaxStableDiffusionPipeline.from_pretrained`): an HF repository layout — Flax msgpack trees or diffusers-PyTorch state dicts —
found as a local directory or as a hub-cache snapshot, plus tokenizer/ and text_encoder/.  CPU tests on the `tiny` family:
the SAME seeded parameter tree is written in every layout and must load back value for value; partial directories and missing
weights are refused; `dtype` follows the reference's contract."""
import json
import os

import numpy as np
import pytest
import torch

from ddpo_amd.models.unet import ParamStore, UNetConfig, unet_param_shapes
from ddpo_amd.models.vae import VAEConfig, vae_decoder_param_shapes
from ddpo_amd.utils import flax_msgpack as FM
from ddpo_amd.utils import serialization as S


def _tree(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    return {n: torch.randn(shp, generator=g) for n, shp in shapes.items()}


def _to_torch_layout(tree, vae_new_names=False):
    """Inverse of serialization.torch_to_flax_tree: Flax names / layouts -> a diffusers-PyTorch state dict."""
    import re
    sd = {}
    for name, t in tree.items():
        stem, leaf = name.rsplit(".", 1)
        if vae_new_names and ".mid_block.attentions_0." in name:
            for new, old in S._VAE_ATTN.items():
                stem = stem.replace("attentions_0." + old, "attentions_0." + new)
        key = re.sub(r"_(\d+)(?=\.|$)", r".\1", stem)          # blocks_0 -> blocks.0, to_out_0 -> to_out.0, net_2 -> net.2
        if leaf == "kernel":
            w = t.permute(3, 2, 0, 1) if t.dim() == 4 else t.t()
            sd[key + ".weight"] = w.contiguous()
        elif leaf == "scale":
            sd[key + ".weight"] = t
        else:
            sd[key + ".bias"] = t
    return sd


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return [chr(c) for c in cs]


def _write_tokenizer(d):
    os.makedirs(d, exist_ok=True)
    vocab = {}
    for ch in _bytes_to_unicode():
        vocab[ch] = len(vocab)
    for ch in _bytes_to_unicode():
        vocab[ch + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    json.dump(vocab, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n")
    json.dump({"model_max_length": 77}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    return len(vocab)


def _text_model(vocab, hidden=64, seed=9):
    from transformers import CLIPTextConfig, CLIPTextModel
    torch.manual_seed(seed)
    return CLIPTextModel(CLIPTextConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                        max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=hidden)).eval()


def _torch_clip_to_flax_tree(model):
    """Inverse of serialization.flax_clip_text_to_torch (what an HF Flax repository's text_encoder/flax_model.msgpack holds)."""
    flat = {}
    for k, v in model.state_dict().items():
        if k.endswith("position_ids"):
            continue
        v = v.detach().numpy()
        if k.endswith("embedding.weight"):
            flat[k[:-len("weight")] + "embedding"] = v
        elif "layer_norm" in k and k.endswith(".weight"):
            flat[k[:-len("weight")] + "scale"] = v
        elif k.endswith(".weight"):
            flat[k[:-len("weight")] + "kernel"] = v.T.copy()
        else:
            flat[k] = v
    return flat


def _write_repo(root, layout, utree, vtree, text="torch", with_encoder_junk=True):
    from safetensors.torch import save_file
    os.makedirs(root, exist_ok=True)
    if layout == "hf-flax":
        for sub, tree in (("unet", utree), ("vae", vtree)):
            os.makedirs(os.path.join(root, sub), exist_ok=True)
            nested = FM.nest({n: t.numpy() for n, t in tree.items()})
            if sub == "vae" and with_encoder_junk:      # a real VAE tree also carries the encoder half: must be ignored
                nested["encoder"] = {"conv_in": {"kernel": np.zeros((3, 3, 3, 8), np.float32)}}
                nested["quant_conv"] = {"kernel": np.zeros((1, 1, 8, 8), np.float32)}
            open(os.path.join(root, sub, "diffusion_flax_model.msgpack"), "wb").write(FM.to_bytes(nested))
    elif layout == "hf-pytorch":
        for sub, tree in (("unet", utree), ("vae", vtree)):
            os.makedirs(os.path.join(root, sub), exist_ok=True)
            save_file(_to_torch_layout(tree, vae_new_names=(sub == "vae")), os.path.join(root, sub, "diffusion_pytorch_model.safetensors"))
    elif layout == "flat":
        save_file({n: t.contiguous() for n, t in utree.items()}, os.path.join(root, "unet.safetensors"))
        save_file({n: t.contiguous() for n, t in vtree.items()}, os.path.join(root, "vae.safetensors"))
    vocab = _write_tokenizer(os.path.join(root, "tokenizer"))
    te = os.path.join(root, "text_encoder")
    model = _text_model(vocab)
    if text == "torch":
        model.save_pretrained(te)
    else:
        os.makedirs(te, exist_ok=True)
        model.config.save_pretrained(te)
        open(os.path.join(te, "flax_model.msgpack"), "wb").write(FM.to_bytes(FM.nest(_torch_clip_to_flax_tree(model))))
    return model


@pytest.fixture()
def trees():
    return (_tree(unet_param_shapes(UNetConfig.named("tiny")), 1), _tree(vae_decoder_param_shapes(VAEConfig.named("tiny")), 2))


@pytest.fixture(autouse=True)
def _tiny_family(monkeypatch):
    monkeypatch.setenv("DDPO_MODEL_CONFIG", "tiny")
    monkeypatch.delenv("DDPO_ALLOW_SYNTHETIC", raising=False)       # these tests are about REAL files; synthetic must be asked for
    monkeypatch.setattr(ParamStore, "pack_bf16", lambda self, bwd=True: None)     # no HIP kernels on the CPU host


@pytest.mark.parametrize("layout,text", [("hf-flax", "flax"), ("hf-pytorch", "torch"), ("flat", "torch")])
def test_every_layout_loads_the_same_parameters(tmp_path, trees, layout, text):
    utree, vtree = trees
    ref_text = _write_repo(str(tmp_path / "repo"), layout, utree, vtree, text=text)
    pipe, params = S.load_unet(None, pretrained_model=str(tmp_path / "repo"), device="cpu")
    assert pipe.synthetic_weights is False and layout in pipe.weights_source and pipe.param_dtype == "float32"
    for n, t in utree.items():
        assert torch.equal(params["unet"][n], t), n
    for n, t in vtree.items():
        assert torch.equal(params["vae"][n], t), n
    # tokenizer + text encoder are the directory's own, not the byte-level stand-in / a random tower
    assert not getattr(pipe.tokenizer, "synthetic", True) and params["text_encoder"].synthetic is False
    ids = pipe.prepare_inputs(["a cat", "two dogs"])
    assert ids.shape == (2, 77)
    with torch.no_grad():
        want = ref_text(torch.as_tensor(np.asarray(ids), dtype=torch.long))[0]
    got = params["text_encoder"](ids)
    assert torch.allclose(got, want, atol=1e-6), float((got - want).abs().max())


def test_hub_id_resolves_to_a_cache_snapshot(tmp_path, trees):
    utree, vtree = trees
    snap = tmp_path / "cache" / "models--duongna--stable-diffusion-v1-4-flax" / "snapshots" / "abc123"
    _write_repo(str(snap), "hf-flax", utree, vtree, text="flax")
    assert S.resolve_pretrained("duongna/stable-diffusion-v1-4-flax", str(tmp_path / "cache")) == str(snap)
    pipe, params = S.load_unet(None, pretrained_model="duongna/stable-diffusion-v1-4-flax", cache=str(tmp_path / "cache"), device="cpu")
    assert pipe.synthetic_weights is False
    assert torch.equal(params["unet"]["conv_in.kernel"], utree["conv_in.kernel"])


def test_partial_directory_is_refused(tmp_path, trees):
    utree, vtree = trees
    root = str(tmp_path / "repo")
    _write_repo(root, "hf-flax", utree, vtree)
    import shutil
    shutil.rmtree(os.path.join(root, "text_encoder"))
    with pytest.raises(FileNotFoundError, match="refusing to mix"):
        S.load_unet(None, pretrained_model=root, device="cpu")
    os.remove(os.path.join(root, "vae", "diffusion_flax_model.msgpack"))
    with pytest.raises(FileNotFoundError, match="vae"):
        S.load_unet(None, pretrained_model=root, device="cpu")


def test_missing_weights_fail_unless_synthetic_is_asked_for(tmp_path, monkeypatch):
    with pytest.raises(FileNotFoundError, match="DDPO_ALLOW_SYNTHETIC"):
        S.load_unet(None, pretrained_model="duongna/stable-diffusion-v1-4-flax", cache=str(tmp_path), device="cpu")
    monkeypatch.setenv("DDPO_ALLOW_SYNTHETIC", "1")
    pipe, params = S.load_unet(None, pretrained_model="duongna/stable-diffusion-v1-4-flax", cache=str(tmp_path), device="cpu")
    assert pipe.synthetic_weights is True and pipe.weights_source is None
    ck = S.save_checkpoint(str(tmp_path / "ck"), params["unet"], step=0, synthetic_weights=pipe.synthetic_weights, flax_format=False)
    from safetensors import safe_open
    assert safe_open(ck, "pt").metadata()["synthetic_weights"] == "True"
    assert os.path.exists(tmp_path / "ck" / "SYNTHETIC_WEIGHTS")


def test_dtype_contract(tmp_path, trees, monkeypatch):
    """reference :343-350: parameters are cast to `dtype`.  bfloat16 = bf16-rounded parameters + the single-pass bf16 datapath."""
    from ddpo_amd import lib as L
    utree, vtree = trees
    root = str(tmp_path / "repo")
    _write_repo(root, "flat", utree, vtree)
    pipe, params = S.load_unet(None, pretrained_model=root, dtype="bfloat16", device="cpu")
    assert L.DATAPATH == "bf16" and pipe.param_dtype == "bfloat16"
    w = params["unet"]["conv_in.kernel"]
    assert torch.equal(w, utree["conv_in.kernel"].to(torch.bfloat16).float()) and not torch.equal(w, utree["conv_in.kernel"])
    L.DATAPATH = "fp32"
    pipe, params = S.load_unet(None, pretrained_model=root, dtype="float32", device="cpu")
    assert L.DATAPATH == "fp32" and torch.equal(params["unet"]["conv_in.kernel"], utree["conv_in.kernel"])
    with pytest.raises(ValueError, match="float32 or bfloat16"):
        S.load_unet(None, pretrained_model=root, dtype="float16", device="cpu")


def test_torch_layout_converter_handles_conv_vs_dense_projections():
    """SD-1.x stores proj_in / proj_out as 1x1 convs, SD-2.x as dense layers; either may meet either target shape."""
    shapes = {"a.proj_in.kernel": (8, 16), "b.proj_in.kernel": (1, 1, 8, 16), "n.scale": (8,), "n.bias": (8,)}
    g = torch.Generator().manual_seed(0)
    w = torch.randn(16, 8, generator=g)
    out = S.torch_to_flax_tree({"a.proj_in.weight": w[:, :, None, None].clone(), "b.proj_in.weight": w.clone(), "n.weight": torch.ones(8),
                                "n.bias": torch.zeros(8), "unrelated.weight": torch.zeros(3)}, shapes)
    assert set(out) == set(shapes)
    assert torch.equal(out["a.proj_in.kernel"], w.t()) and torch.equal(out["b.proj_in.kernel"], w.t().reshape(1, 1, 8, 16))


def test_chunked_flax_leaf_uses_flax_shape_encoding(monkeypatch):
    """flax writes the shape of a chunked (> 1 GiB) leaf as {"0": n, "1": m}; both forms are read."""
    import msgpack
    monkeypatch.setattr(FM, "_MAX_CHUNK_BYTES", 64)
    a = np.arange(60, dtype=np.float32).reshape(5, 12)
    raw = msgpack.unpackb(FM.to_bytes({"w": a}), ext_hook=FM._ext_unpack, raw=False, strict_map_key=False)
    assert raw["w"]["shape"] == {"0": 5, "1": 12}
    assert np.array_equal(FM.from_bytes(FM.to_bytes({"w": a}))["w"], a)
    legacy = dict(raw["w"], shape=[5, 12])
    assert np.array_equal(FM._unchunk({"w": legacy})["w"], a)
