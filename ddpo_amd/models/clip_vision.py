"""CLIP ViT image tower on the gfx950 kernels of this library (forward only) — the feature extractor of the aesthetic reward.

Replaces `FlaxCLIPModel.get_image_features` + `CLIPProcessor` of /root/reference/ddpo/training/callbacks.py:60-95 (the reference pmaps it
on the same devices as the sampler; here it runs on the sampling GPU, on a private HIP stream, next to the sampling of the next batch):

    processor   host, PIL: float [0,1] -> uint8 (truncation) -> bicubic resize of the short side to 224 -> centre crop -> /255 -> normalise
    patch embed stride-14 14x14 convolution = ONE GEMM over the (N*256, 3*14*14 -> 608) patch matrix; the position embedding rides in
                the GEMM epilogue as its residual operand
    24 x layer  LayerNorm (bf16 hi/lo planes out) -> q / k / v GEMMs -> flash attention d=64, 257 keys -> out-proj GEMM (+residual)
                -> LayerNorm -> fc1 GEMM -> quick-GELU -> fc2 GEMM (+residual)
    pooling     CLS row -> post LayerNorm -> bias-free visual projection GEMM

Every contraction is `ddpo_gemm_conv_fwd*` (bf16x3-split MFMA when lib.DATAPATH says so, exact-fp32 MFMA otherwise), the norms are
`ddpo_layernorm_fwd*`, attention is `ddpo_attention_fwd*`; torch only reshapes (im2col view, CLS-row copy).  Weights are held in the
engine's (in, out) layout; `load_state_dict` takes transformers' torch names ((out, in) / OIHW), `flax_tree_to_torch_names` converts
an HF Flax tree first.  There is no non-HIP path.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from .. import lib as L
from .unet import ParamStore

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class VisionConfig:
    def __init__(self, hidden=1024, layers=24, heads=16, mlp=4096, image=224, patch=14, proj=768, eps=1e-5):
        self.hidden, self.layers, self.heads, self.mlp, self.image, self.patch, self.proj, self.eps = hidden, layers, heads, mlp, image, patch, proj, eps
        self.grid = image // patch
        self.tokens = self.grid * self.grid + 1
        self.k_patch = 3 * patch * patch
        self.k_pad = (self.k_patch + 31) // 32 * 32          # 588 -> 608: 32-wide k-tiles of the buffer-addressed GEMM

    @staticmethod
    def named(name):
        if name in ("vit-l/14", "openai/clip-vit-large-patch14", "l14"):
            return VisionConfig()
        if name == "tiny":
            return VisionConfig(hidden=64, layers=2, heads=4, mlp=128, image=56, patch=14, proj=32)
        raise KeyError(name)


def preprocess(images, size=224):
    """`CLIPProcessor(images=list(images))` of transformers 4.28.1 (reference callbacks.py:88-89) on float32 (N,H,W,3) arrays in
    [0,1]: float -> uint8 by TRUNCATION (to_pil_image), PIL bicubic resize of the short side to `size`, centre crop, x / 255,
    (x - mean) / std, channels first.  Host work on N small images; PIL makes the resize byte-identical to the reference's."""
    from PIL import Image
    mean, std = np.asarray(CLIP_MEAN, np.float32), np.asarray(CLIP_STD, np.float32)
    out = []
    for x in images:
        x = np.asarray(x)
        u8 = (x * 255).astype(np.uint8) if np.issubdtype(x.dtype, np.floating) else x
        h, w = u8.shape[:2]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long / short)
        ow, oh = (new_short, new_long) if w <= h else (new_long, new_short)
        r = np.array(Image.fromarray(u8).resize((ow, oh), resample=Image.BICUBIC))
        top, left = (oh - size) // 2, (ow - size) // 2
        r = r[top:top + size, left:left + size]
        f = (r.astype(np.float32) * (1 / 255)).astype(np.float32)
        out.append(((f - mean) / std).transpose(2, 0, 1))
    return np.stack(out).astype(np.float32)


def vision_param_shapes(cfg: VisionConfig):
    """Engine layouts: dense kernels (in, out); the patch convolution as a (k_pad, hidden) matrix, rows ordered (channel, ky, kx)."""
    d = OrderedDict()
    C = cfg.hidden
    d["embeddings.class_embedding"] = (C,)
    d["embeddings.patch_embedding.kernel"] = (cfg.k_pad, C)
    d["embeddings.position_embedding"] = (cfg.tokens, C)
    for n in ("pre_layrnorm", "post_layernorm"):
        d[n + ".scale"] = (C,); d[n + ".bias"] = (C,)
    for i in range(cfg.layers):
        p = f"layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            d[p + n + ".scale"] = (C,); d[p + n + ".bias"] = (C,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            d[p + n + ".kernel"] = (C, C); d[p + n + ".bias"] = (C,)
        d[p + "fc1.kernel"] = (C, cfg.mlp); d[p + "fc1.bias"] = (cfg.mlp,)
        d[p + "fc2.kernel"] = (cfg.mlp, C); d[p + "fc2.bias"] = (C,)
    d["visual_projection.kernel"] = (C, cfg.proj)
    return d


def flax_tree_to_torch_names(flat):
    """HF Flax CLIP params (flattened with '.') -> transformers' torch state-dict names / layouts (dense kernel (in,out) -> weight
    (out,in); conv kernel HWIO -> OIHW; LayerNorm scale -> weight; embedding -> weight)."""
    sd = {}
    for name, v in flat.items():
        t = torch.as_tensor(np.asarray(v)).float()
        if name.endswith(".kernel"):
            sd[name[:-len("kernel")] + "weight"] = (t.permute(3, 2, 0, 1) if t.dim() == 4 else t.t()).contiguous()
        elif name.endswith(".scale") or name.endswith(".embedding"):
            sd[name.rsplit(".", 1)[0] + ".weight"] = t
        else:
            sd[name] = t
    return sd


class ClipVisionTower:
    def __init__(self, cfg: VisionConfig, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.params = ParamStore(vision_param_shapes(cfg), self.device)
        self._pos_cache = {}

    def load_state_dict(self, sd):
        """transformers torch names (`vision_model.…`, `visual_projection.weight`); extra keys (the text tower) are ignored."""
        cfg, P = self.cfg, self.params
        g = lambda k: torch.as_tensor(sd[k]).float()
        v = "vision_model."
        tree = {"embeddings.class_embedding": g(v + "embeddings.class_embedding"),
                "embeddings.position_embedding": g(v + "embeddings.position_embedding.weight")}
        w = g(v + "embeddings.patch_embedding.weight")                                  # (C, 3, p, p)
        if tuple(w.shape) != (cfg.hidden, 3, cfg.patch, cfg.patch):
            raise ValueError(f"patch embedding has shape {tuple(w.shape)}, expected {(cfg.hidden, 3, cfg.patch, cfg.patch)}")
        kp = torch.zeros(cfg.k_pad, cfg.hidden)
        kp[:cfg.k_patch] = w.reshape(cfg.hidden, cfg.k_patch).t()
        tree["embeddings.patch_embedding.kernel"] = kp
        for n in ("pre_layrnorm", "post_layernorm"):
            tree[n + ".scale"], tree[n + ".bias"] = g(v + n + ".weight"), g(v + n + ".bias")
        for i in range(cfg.layers):
            src, dst = f"{v}encoder.layers.{i}.", f"layers.{i}."
            for n in ("layer_norm1", "layer_norm2"):
                tree[dst + n + ".scale"], tree[dst + n + ".bias"] = g(src + n + ".weight"), g(src + n + ".bias")
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                tree[dst + n + ".kernel"], tree[dst + n + ".bias"] = g(src + f"self_attn.{n}.weight").t().contiguous(), g(src + f"self_attn.{n}.bias")
            for n in ("fc1", "fc2"):
                tree[dst + n + ".kernel"], tree[dst + n + ".bias"] = g(src + f"mlp.{n}.weight").t().contiguous(), g(src + f"mlp.{n}.bias")
        tree["visual_projection.kernel"] = g("visual_projection.weight").t().contiguous()
        P.load_dict(tree)
        self._pos_cache.clear()
        if L.current_datapath() != "fp32":
            self.pack()

    def pack(self):
        """bf16 hi / lo planes of every contraction weight (frozen reward model: once)."""
        for n, w in self.params.views.items():
            if n.endswith(".kernel"):
                L.pack_weights(w, bwd=False)

    def _pos_rows(self, N):
        """Position embedding of the patch rows tiled over the batch: the residual operand of the patch-embedding GEMM."""
        ent = self._pos_cache.get(N)
        if ent is None:
            pos = self.params["embeddings.position_embedding"]
            ent = (pos[1:].unsqueeze(0).expand(N, -1, -1).reshape(-1, self.cfg.hidden).contiguous(),
                   (self.params["embeddings.class_embedding"] + pos[0]).contiguous())
            self._pos_cache[N] = ent
        return ent

    def forward(self, pixel_values):
        """(N,3,S,S) fp32 on the device -> image_embeds (N, proj).  The reward model keeps fp32 parameters in the reference whatever
        dtype the SD trees are cast to, so it never runs on the single-pass bf16 datapath (lib.fp32_class_datapath)."""
        with L.fp32_class_datapath():
            return self._forward(pixel_values)

    def _forward(self, pixel_values):
        cfg, P = self.cfg, self.params
        N = pixel_values.shape[0]
        if tuple(pixel_values.shape[1:]) != (3, cfg.image, cfg.image):
            raise ValueError(f"pixel_values must be (N, 3, {cfg.image}, {cfg.image}), got {tuple(pixel_values.shape)}")
        g, p, C, T = cfg.grid, cfg.patch, cfg.hidden, cfg.tokens
        # im2col of a stride-p, p x p convolution is a pure re-ordering: rows = patches, columns = (channel, ky, kx), zero-padded to k_pad
        patches = torch.zeros(N * g * g, cfg.k_pad, dtype=torch.float32, device=self.device)
        patches[:, :cfg.k_patch] = pixel_values.reshape(N, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(N * g * g, cfg.k_patch)
        pos_rows, cls_row = self._pos_rows(N)
        pe = L.linear(patches, P["embeddings.patch_embedding.kernel"], residual=pos_rows)            # conv (no bias) + position embedding
        h = torch.empty(N, T, C, dtype=torch.float32, device=self.device)
        h[:, 0] = cls_row
        h[:, 1:] = pe.view(N, g * g, C)
        h = h.view(N * T, C)
        h = L.layernorm(h, P["pre_layrnorm.scale"], P["pre_layrnorm.bias"], cfg.eps)
        d = C // cfg.heads
        for i in range(cfg.layers):
            pre = f"layers.{i}."
            w = lambda n: P[pre + n + ".kernel"]
            b = lambda n: P[pre + n + ".bias"]
            pl = all(L.planes_ok(w(n), C, N * T) for n in ("q_proj", "k_proj", "v_proj"))
            t = L.layernorm(h, P[pre + "layer_norm1.scale"], P[pre + "layer_norm1.bias"], cfg.eps, planes=pl)
            q, k, v = L.linear(t, w("q_proj"), b("q_proj")), L.linear(t, w("k_proj"), b("k_proj")), L.linear(t, w("v_proj"), b("v_proj"))
            a = L.attention(q, k, v, N, cfg.heads, T, T, d)
            h = L.linear(a, w("out_proj"), b("out_proj"), residual=h)
            t = L.layernorm(h, P[pre + "layer_norm2.scale"], P[pre + "layer_norm2.bias"], cfg.eps, planes=L.planes_ok(w("fc1"), C, N * T))
            f = L.linear(t, w("fc1"), b("fc1"))
            L.quick_gelu(f, out=f)
            h = L.linear(f, w("fc2"), b("fc2"), residual=h)
        pooled = h.view(N, T, C)[:, 0].contiguous()
        pooled = L.layernorm(pooled, P["post_layernorm.scale"], P["post_layernorm.bias"], cfg.eps)
        return L.linear(pooled, P["visual_projection.kernel"])

    __call__ = forward
