"""Pins oracle/clip_vision.py (the restated aesthetic-reward chain, reference callbacks.py:60-95) against independent implementations
that can be executed here: transformers' torch `CLIPVisionModelWithProjection` for the image tower, the installed
`CLIPImageProcessor` for the uint8 -> pixel_values half of the processor, and the reference's own `set_weights` layout rule
(ddpo/models/laion.py:38-51) for the MLP.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import clip_vision as OC


def _hf_model(cfg):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=cfg.hidden, intermediate_size=cfg.mlp, num_hidden_layers=cfg.layers,
                                                        num_attention_heads=cfg.heads, image_size=cfg.image, patch_size=cfg.patch,
                                                        projection_dim=cfg.proj, hidden_act="quick_gelu", layer_norm_eps=cfg.eps)).eval()
    return hf


@pytest.mark.parametrize("name,cfg,n", [("tiny", OC.VIT_TINY, 3), ("vit-l/14", OC.VIT_L14, 1)])
def test_image_tower_matches_transformers_torch_clip(name, cfg, n):
    params = OC.init_params(OC.vision_param_shapes(cfg), seed=4)
    hf = _hf_model(cfg)
    sd = hf.state_dict()
    assert {k for k in sd if not k.endswith("position_ids")} == set(params), "parameter naming differs from transformers' CLIP"
    for k, v in params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    hf.load_state_dict(params, strict=False)
    x = torch.randn(n, 3, cfg.image, cfg.image, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = hf(pixel_values=x).image_embeds
        got = OC.image_features(params, cfg, x)
    assert got.shape == want.shape == (n, cfg.proj)
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1e-5, err
    # float64 evaluation of the restatement = the ground truth the GPU test uses; fp32 sits within rounding of it
    got64 = OC.image_features({k: v.double() for k, v in params.items()}, cfg, x)
    assert float((got64.float() - got).abs().max() / want.abs().max()) < 1e-5


@pytest.mark.parametrize("H,W,S", [(96, 128, 32), (512, 512, 224), (300, 200, 224), (224, 224, 224)])
def test_processor_from_uint8_matches_installed_clip_image_processor(H, W, S):
    from transformers import CLIPImageProcessor
    u8 = np.random.default_rng(H + W).integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    proc = CLIPImageProcessor(size={"shortest_edge": S}, crop_size={"height": S, "width": S})
    want = proc(images=list(u8), return_tensors="np")["pixel_values"]
    got = OC.preprocess(u8, S)
    assert got.shape == want.shape == (2, 3, S, S) and got.dtype == np.float32
    assert float(np.abs(got - want).max()) < 2e-6


def test_processor_float_input_is_truncated_to_uint8_first():
    """transformers 4.28.1 `to_pil_image`: float arrays are multiplied by 255 and cast with astype(uint8) — TRUNCATION, the same
    conversion the reference's jpeg reward applies (ddpo/utils/hdf5.py:33).  (Recalled, see oracle/clip_vision.py.)"""
    x = np.random.default_rng(0).random((1, 64, 64, 3), dtype=np.float32)
    assert np.array_equal(OC.preprocess(x, 32), OC.preprocess((x * 255).astype(np.uint8), 32))
    assert not np.array_equal(OC.preprocess(x, 32), OC.preprocess(np.round(x * 255).astype(np.uint8), 32))


def test_mlp_follows_the_reference_weight_file_layout():
    """laion.set_weights (:38-51): `.pth` keys layers.{0,2,4,6,7}.{weight,bias}, torch (out,in) weights transposed into Dense kernels."""
    mp = OC.init_params(OC.mlp_param_shapes(768), seed=2)
    assert list(mp) == [f"layers.{i}.{p}" for i in (0, 2, 4, 6, 7) for p in ("weight", "bias")]
    assert [tuple(mp[f"layers.{i}.weight"].shape) for i in (0, 2, 4, 6, 7)] == [(1024, 768), (128, 1024), (64, 128), (16, 64), (1, 16)]
    f = torch.randn(5, 768, generator=torch.Generator().manual_seed(0))
    seq = torch.nn.Sequential(torch.nn.Linear(768, 1024), torch.nn.Dropout(0.2), torch.nn.Linear(1024, 128), torch.nn.Dropout(0.2),
                              torch.nn.Linear(128, 64), torch.nn.Dropout(0.1), torch.nn.Linear(64, 16), torch.nn.Linear(16, 1)).eval()
    seq.load_state_dict({k.replace("layers.", ""): v for k, v in mp.items()})      # the published predictor's own module layout
    with torch.no_grad():
        assert torch.allclose(OC.aesthetic_mlp(mp, f), seq(f), atol=1e-6)


def test_scores_end_to_end_shapes_and_determinism():
    cfg = OC.VIT_TINY
    params = OC.init_params(OC.vision_param_shapes(cfg), seed=4)
    mp = OC.init_params(OC.mlp_param_shapes(cfg.proj), seed=5)
    imgs = np.random.default_rng(3).random((4, 80, 64, 3), dtype=np.float32)
    s1 = OC.aesthetic_scores(params, mp, cfg, imgs)
    s2 = OC.aesthetic_scores(params, mp, cfg, imgs)
    assert s1.shape == (4,) and torch.equal(s1, s2) and s1.dtype == torch.float64
