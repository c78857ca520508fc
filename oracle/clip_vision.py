"""Oracle (TEST INFRASTRUCTURE ONLY — imported by tests/, never by the product path): the aesthetic reward of
/root/reference/ddpo/training/callbacks.py:60-95 restated in plain torch-CPU (any float dtype; tests use float64):

    processor(images=list(images))            CLIPProcessor of `openai/clip-vit-large-patch14`           (:88-89)
    FlaxCLIPModel.get_image_features          CLIP ViT-L/14 image tower + visual projection            (:76-77)
    features / ||features||                                                                            (:80-82)
    AestheticClassifier                       /root/reference/ddpo/models/laion.py:7-19 (Dense 768-1024-128-64-16-1, dropouts inert)

The tower and the processor are un-vendored third-party code (transformers==4.28.1).  Pinning status:
  * tower (`image_features`): PINNED against transformers' torch `CLIPVisionModelWithProjection` — an independent implementation of
    the same architecture — executed here with the same seeded weights (tests/test_oracle_clip_vision.py: 1e-5 in fp32 on the full
    ViT-L/14 geometry).  The Flax class the reference calls and the torch class share `modeling_*_clip` semantics: pre-LayerNorm
    after the embeddings, quick_gelu MLP, CLS-token pooling, post-LayerNorm on the pooled token only, bias-free visual projection.
  * processor, from uint8 onwards (PIL bicubic resize of the short side to 224, centre crop, x/255, (x - mean) / std, CHW): PINNED
    against the installed `CLIPImageProcessor` on square / landscape / portrait inputs (4e-7).
  * processor, float input -> uint8: PARITY UNPINNED, recalled from transformers 4.28.1 `image_transforms.to_pil_image` — an array
    whose first element is a Python/numpy float is multiplied by 255 and cast with `.astype(np.uint8)` (truncation) before PIL
    sees it.  The installed 5.x processor no longer does this (it rescales float inputs by 1/255 a second time), so it cannot
    referee this one line; it is the same truncating conversion the reference's own jpeg reward uses (ddpo/utils/hdf5.py:33).
  * LAION MLP: restated from the reference file itself (laion.py:7-51, `set_weights` transposes torch (out,in) into Dense kernels).

Parameter names follow transformers' torch state dict (`vision_model.embeddings.patch_embedding.weight`, ...,
`visual_projection.weight`) so that real checkpoints and the HF module used for pinning share one naming.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class VisionCfg:
    def __init__(self, hidden=1024, layers=24, heads=16, mlp=4096, image=224, patch=14, proj=768, eps=1e-5):
        self.hidden, self.layers, self.heads, self.mlp, self.image, self.patch, self.proj, self.eps = hidden, layers, heads, mlp, image, patch, proj, eps

    @property
    def tokens(self):
        return (self.image // self.patch) ** 2 + 1


VIT_L14 = VisionCfg()
VIT_TINY = VisionCfg(hidden=64, layers=2, heads=4, mlp=128, image=56, patch=14, proj=32)
MLP_DIMS = (1024, 128, 64, 16, 1)


def vision_param_shapes(cfg):
    d = OrderedDict()
    v = "vision_model."
    d[v + "embeddings.class_embedding"] = (cfg.hidden,)
    d[v + "embeddings.patch_embedding.weight"] = (cfg.hidden, 3, cfg.patch, cfg.patch)
    d[v + "embeddings.position_embedding.weight"] = (cfg.tokens, cfg.hidden)
    for n in ("pre_layrnorm", "post_layernorm"):                     # (sic: transformers' spelling)
        d[v + n + ".weight"] = (cfg.hidden,); d[v + n + ".bias"] = (cfg.hidden,)
    for i in range(cfg.layers):
        L = f"{v}encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            d[L + n + ".weight"] = (cfg.hidden,); d[L + n + ".bias"] = (cfg.hidden,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            d[L + f"self_attn.{n}.weight"] = (cfg.hidden, cfg.hidden); d[L + f"self_attn.{n}.bias"] = (cfg.hidden,)
        d[L + "mlp.fc1.weight"] = (cfg.mlp, cfg.hidden); d[L + "mlp.fc1.bias"] = (cfg.mlp,)
        d[L + "mlp.fc2.weight"] = (cfg.hidden, cfg.mlp); d[L + "mlp.fc2.bias"] = (cfg.hidden,)
    d["visual_projection.weight"] = (cfg.proj, cfg.hidden)
    return d


def mlp_param_shapes(in_dim=768):
    d, a = OrderedDict(), in_dim
    for idx, b in zip((0, 2, 4, 6, 7), MLP_DIMS):                      # the .pth names of laion.set_weights (:41-42)
        d[f"layers.{idx}.weight"] = (b, a); d[f"layers.{idx}.bias"] = (b,)
        a = b
    return d


def init_params(shapes, seed=0):
    """Deterministic synthetic weights (no CLIP / LAION checkpoints exist offline): linear / conv weights N(0, 1/fan_in), LayerNorm
    weights 1 + N(0, 0.1^2), biases and embeddings N(0, 0.02^2)."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shp in shapes.items():
        if "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith(".weight") and len(shp) >= 2 and "position_embedding" not in name:
            t = torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        else:
            t = 0.02 * torch.randn(shp, generator=g)
        out[name] = t
    return out


def preprocess(images, size=224):
    """CLIPImageProcessor (transformers 4.28.1) on the float32 (N,H,W,3) arrays in [0,1] the entrypoint hands to the reward:
    -> float32 (N,3,size,size).  See the module docstring for which lines are pinned."""
    from PIL import Image
    mean, std = np.asarray(CLIP_MEAN, np.float32), np.asarray(CLIP_STD, np.float32)
    out = []
    for x in images:
        x = np.asarray(x)
        u8 = (x * 255).astype(np.uint8) if np.issubdtype(x.dtype, np.floating) else x     # to_pil_image: rescale(255) + astype(uint8)
        h, w = u8.shape[:2]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long / short)                              # get_resize_output_image_size, default_to_square=False
        ow, oh = (new_short, new_long) if w <= h else (new_long, new_short)
        r = np.array(Image.fromarray(u8).resize((ow, oh), resample=Image.BICUBIC))
        top, left = (oh - size) // 2, (ow - size) // 2                                    # center_crop
        r = r[top:top + size, left:left + size]
        f = (r.astype(np.float32) * (1 / 255)).astype(np.float32)                         # rescale
        out.append(((f - mean) / std).transpose(2, 0, 1))                                 # normalize, channels first
    return np.stack(out).astype(np.float32)


def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def image_features(params, cfg, pixel_values):
    """CLIPVisionTransformer + visual_projection: (N,3,S,S) -> (N, proj).  `params` values may be any float dtype."""
    P = params
    dt = P["visual_projection.weight"].dtype
    x = torch.as_tensor(pixel_values).to(dt)
    N = x.shape[0]
    v = "vision_model."
    pe = torch.nn.functional.conv2d(x, P[v + "embeddings.patch_embedding.weight"], None, stride=cfg.patch)     # (N, C, g, g), no bias
    pe = pe.flatten(2).transpose(1, 2)                                                                          # (N, g*g, C)
    cls = P[v + "embeddings.class_embedding"].expand(N, 1, -1)
    h = torch.cat([cls, pe], 1) + P[v + "embeddings.position_embedding.weight"][None]
    h = _ln(h, P[v + "pre_layrnorm.weight"], P[v + "pre_layrnorm.bias"], cfg.eps)
    d = cfg.hidden // cfg.heads
    for i in range(cfg.layers):
        L = f"{v}encoder.layers.{i}."
        r = h
        t = _ln(h, P[L + "layer_norm1.weight"], P[L + "layer_norm1.bias"], cfg.eps)
        lin = lambda n, t_: t_ @ P[L + f"self_attn.{n}.weight"].t() + P[L + f"self_attn.{n}.bias"]
        sp = lambda t_: t_.view(N, -1, cfg.heads, d).transpose(1, 2)
        q, k, vv = sp(lin("q_proj", t)) * d ** -0.5, sp(lin("k_proj", t)), sp(lin("v_proj", t))
        a = torch.softmax(q @ k.transpose(-1, -2), -1) @ vv
        a = a.transpose(1, 2).reshape(N, -1, cfg.hidden)
        h = r + lin("out_proj", a)
        r = h
        t = _ln(h, P[L + "layer_norm2.weight"], P[L + "layer_norm2.bias"], cfg.eps)
        t = t @ P[L + "mlp.fc1.weight"].t() + P[L + "mlp.fc1.bias"]
        t = t * torch.sigmoid(1.702 * t)                                                                        # quick_gelu
        h = r + (t @ P[L + "mlp.fc2.weight"].t() + P[L + "mlp.fc2.bias"])
    pooled = _ln(h[:, 0], P[v + "post_layernorm.weight"], P[v + "post_layernorm.bias"], cfg.eps)
    return pooled @ P["visual_projection.weight"].t()


def aesthetic_mlp(mlp_params, feats):
    x = feats
    for idx in (0, 2, 4, 6, 7):
        x = x @ mlp_params[f"layers.{idx}.weight"].t() + mlp_params[f"layers.{idx}.bias"]
    return x


def aesthetic_scores(params, mlp_params, cfg, images, dtype=torch.float64):
    """float (N,H,W,3) in [0,1] -> (N,) scores, the whole callbacks.py:76-92 chain."""
    P = {k: v.to(dtype) for k, v in params.items()}
    M = {k: v.to(dtype) for k, v in mlp_params.items()}
    f = image_features(P, cfg, torch.from_numpy(preprocess(images, cfg.image)))
    f = f / torch.linalg.norm(f, dim=-1, keepdim=True)
    return aesthetic_mlp(M, f)[:, 0]
