"""Oracle (test infrastructure only): Stable-Diffusion U-Net and VAE decoder forward, plain torch (CPU, fp32/fp64).

The reference calls diffusers==0.12.1 `FlaxUNet2DConditionModel.apply`
(/root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:219-224,
 /root/reference/ddpo/training/policy_gradient.py:87-102) and `FlaxAutoencoderKL.decode`
(/root/reference/pipeline/policy_gradient.py:174-182).  diffusers is an un-vendored third-party dependency that
cannot be installed here, so the architecture is restated from its published definition and
anchored on the exact parameter counts 859,520,964 (SD-1.x U-Net), 865,910,724 (SD-2.1 U-Net) and 49,490,199
(VAE decoder incl. post_quant_conv).  PARITY UNPINNED against real JAX outputs (no weights, no jax).

Parameters are kept in the Flax layout and naming: conv kernels HWIO, dense kernels (in, out), names like
"down_blocks_0.attentions_1.transformer_blocks_0.attn1.to_q.kernel".
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as TF


@dataclass
class UNetCfg:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttn", "CrossAttn", "CrossAttn", "Plain")
    layers_per_block: int = 2
    attention_head_dim: Tuple[int, ...] = (8, 8, 8, 8)       # = number of heads per level (diffusers 0.12.1 Flax)
    cross_attention_dim: int = 768
    use_linear_projection: bool = False
    norm_groups: int = 32
    prediction_type: str = "epsilon"

    @property
    def up_block_types(self):
        return tuple(reversed(self.down_block_types))


SD15 = UNetCfg()
SD21 = UNetCfg(attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True,
               prediction_type="v_prediction")
TINY = UNetCfg(block_out_channels=(32, 64, 128, 128), attention_head_dim=(8, 8, 8, 8), cross_attention_dim=64)
# SD-2.1-shaped toy: linear projections, v-prediction, head counts growing with width (head dim 16 everywhere)
TINY21 = UNetCfg(block_out_channels=(32, 64, 128, 128), attention_head_dim=(2, 4, 8, 8), cross_attention_dim=96,
                 use_linear_projection=True, prediction_type="v_prediction")


# ------------------------------------------------------------------------------------------------
# parameter inventory
# ------------------------------------------------------------------------------------------------
def _conv(d, name, cin, cout, k):
    d[name + ".kernel"] = (k, k, cin, cout)
    d[name + ".bias"] = (cout,)


def _dense(d, name, cin, cout, bias=True):
    d[name + ".kernel"] = (cin, cout)
    if bias:
        d[name + ".bias"] = (cout,)


def _norm(d, name, c):
    d[name + ".scale"] = (c,)
    d[name + ".bias"] = (c,)


def _resnet(d, name, cin, cout, temb):
    _norm(d, name + ".norm1", cin)
    _conv(d, name + ".conv1", cin, cout, 3)
    if temb:
        _dense(d, name + ".time_emb_proj", temb, cout)
    _norm(d, name + ".norm2", cout)
    _conv(d, name + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, name + ".conv_shortcut", cin, cout, 1)


def _transformer(d, name, c, ctx, linear):
    _norm(d, name + ".norm", c)
    if linear:
        _dense(d, name + ".proj_in", c, c)
    else:
        _conv(d, name + ".proj_in", c, c, 1)
    b = name + ".transformer_blocks_0"
    for a, kv in (("attn1", c), ("attn2", ctx)):
        _dense(d, f"{b}.{a}.to_q", c, c, bias=False)
        _dense(d, f"{b}.{a}.to_k", kv, c, bias=False)
        _dense(d, f"{b}.{a}.to_v", kv, c, bias=False)
        _dense(d, f"{b}.{a}.to_out_0", c, c)
    _dense(d, b + ".ff.net_0.proj", c, 8 * c)
    _dense(d, b + ".ff.net_2", 4 * c, c)
    for n in ("norm1", "norm2", "norm3"):
        _norm(d, f"{b}.{n}", c)
    if linear:
        _dense(d, name + ".proj_out", c, c)
    else:
        _conv(d, name + ".proj_out", c, c, 1)


def unet_param_shapes(cfg: UNetCfg):
    d = OrderedDict()
    boc = cfg.block_out_channels
    temb = boc[0] * 4
    _conv(d, "conv_in", cfg.in_channels, boc[0], 3)
    _dense(d, "time_embedding.linear_1", boc[0], temb)
    _dense(d, "time_embedding.linear_2", temb, temb)
    out_c = boc[0]
    for i, typ in enumerate(cfg.down_block_types):
        in_c, out_c = out_c, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(d, f"down_blocks_{i}.resnets_{j}", in_c if j == 0 else out_c, out_c, temb)
            if typ == "CrossAttn":
                _transformer(d, f"down_blocks_{i}.attentions_{j}", out_c, cfg.cross_attention_dim, cfg.use_linear_projection)
        if i != len(boc) - 1:
            _conv(d, f"down_blocks_{i}.downsamplers_0.conv", out_c, out_c, 3)
    _resnet(d, "mid_block.resnets_0", boc[-1], boc[-1], temb)
    _transformer(d, "mid_block.attentions_0", boc[-1], cfg.cross_attention_dim, cfg.use_linear_projection)
    _resnet(d, "mid_block.resnets_1", boc[-1], boc[-1], temb)
    rev = list(reversed(boc))
    out_c = rev[0]
    for i, typ in enumerate(cfg.up_block_types):
        prev_out = out_c
        out_c = rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = in_c if j == cfg.layers_per_block else out_c
            rin = prev_out if j == 0 else out_c
            _resnet(d, f"up_blocks_{i}.resnets_{j}", rin + skip, out_c, temb)
            if typ == "CrossAttn":
                _transformer(d, f"up_blocks_{i}.attentions_{j}", out_c, cfg.cross_attention_dim, cfg.use_linear_projection)
        if i != len(boc) - 1:
            _conv(d, f"up_blocks_{i}.upsamplers_0.conv", out_c, out_c, 3)
    _norm(d, "conv_norm_out", boc[0])
    _conv(d, "conv_out", boc[0], cfg.out_channels, 3)
    return d


@dataclass
class VAECfg:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.18215


VAE_SD = VAECfg()
VAE_TINY = VAECfg(block_out_channels=(32, 32, 64, 64))


def vae_decoder_param_shapes(cfg: VAECfg):
    d = OrderedDict()
    boc = cfg.block_out_channels
    _conv(d, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    _conv(d, "decoder.conv_in", cfg.latent_channels, boc[-1], 3)
    _resnet(d, "decoder.mid_block.resnets_0", boc[-1], boc[-1], 0)
    a = "decoder.mid_block.attentions_0"
    _norm(d, a + ".group_norm", boc[-1])
    for n in ("query", "key", "value", "proj_attn"):
        _dense(d, f"{a}.{n}", boc[-1], boc[-1])
    _resnet(d, "decoder.mid_block.resnets_1", boc[-1], boc[-1], 0)
    rev = list(reversed(boc))
    out_c = rev[0]
    for i in range(len(boc)):
        prev = out_c
        out_c = rev[i]
        for j in range(cfg.layers_per_block + 1):
            _resnet(d, f"decoder.up_blocks_{i}.resnets_{j}", prev if j == 0 else out_c, out_c, 0)
        if i != len(boc) - 1:
            _conv(d, f"decoder.up_blocks_{i}.upsamplers_0.conv", out_c, out_c, 3)
    _norm(d, "decoder.conv_norm_out", boc[0])
    _conv(d, "decoder.conv_out", boc[0], cfg.out_channels, 3)
    return d


def vae_encoder_param_shapes(cfg: VAECfg, in_channels=3):
    """diffusers 0.12.1 FlaxEncoder + quant_conv (FlaxAutoencoderKL.encode): 34,163,664 parameters for the SD VAE (83,653,863 of the whole AutoencoderKL minus the decoder half 49,490,199)."""
    d = OrderedDict()
    boc = cfg.block_out_channels
    _conv(d, "encoder.conv_in", in_channels, boc[0], 3)
    ch = boc[0]
    for i, out_c in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet(d, f"encoder.down_blocks_{i}.resnets_{j}", ch, out_c, 0)
            ch = out_c
        if i != len(boc) - 1:
            _conv(d, f"encoder.down_blocks_{i}.downsamplers_0.conv", ch, ch, 3)
    _resnet(d, "encoder.mid_block.resnets_0", ch, ch, 0)
    a = "encoder.mid_block.attentions_0"
    _norm(d, a + ".group_norm", ch)
    for n in ("query", "key", "value", "proj_attn"):
        _dense(d, f"{a}.{n}", ch, ch)
    _resnet(d, "encoder.mid_block.resnets_1", ch, ch, 0)
    _norm(d, "encoder.conv_norm_out", ch)
    _conv(d, "encoder.conv_out", ch, 2 * cfg.latent_channels, 3)
    _conv(d, "quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return d


def count_params(shapes):
    return sum(math.prod(s) for s in shapes.values())


def init_params(shapes, seed=0, dtype=torch.float32):
    """Deterministic synthetic weights: kernels N(0, 1/fan_in), biases N(0, 0.02^2), norm scale 1+N(0,0.1^2),
    norm bias N(0, 0.05^2).  (No real checkpoints exist offline.)"""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith(".kernel"):
            fan_in = math.prod(shp[:-1])
            t = torch.randn(shp, generator=g, dtype=torch.float32) / math.sqrt(fan_in)
        elif name.endswith(".scale"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g, dtype=torch.float32)
        elif ".norm" in name or "group_norm" in name or "conv_norm_out" in name:
            t = 0.05 * torch.randn(shp, generator=g, dtype=torch.float32)
        else:
            t = 0.02 * torch.randn(shp, generator=g, dtype=torch.float32)
        out[name] = t.to(dtype)
    return out


# ------------------------------------------------------------------------------------------------
# forward (NCHW torch ops; parameters in Flax layout)
# ------------------------------------------------------------------------------------------------
def _conv2d(p, name, x, stride=1, pad=1):
    w = p[name + ".kernel"].permute(3, 2, 0, 1)          # HWIO -> OIHW
    return TF.conv2d(x, w, p[name + ".bias"], stride=stride, padding=pad)


def _gn(p, name, x, groups, eps):
    return TF.group_norm(x, groups, p[name + ".scale"], p[name + ".bias"], eps)


def _dense_f(p, name, x):
    y = x @ p[name + ".kernel"]
    b = p.get(name + ".bias")
    return y if b is None else y + b


def _resnet_f(p, name, x, temb, groups, eps):
    h = TF.silu(_gn(p, name + ".norm1", x, groups, eps))
    h = _conv2d(p, name + ".conv1", h)
    if temb is not None:
        h = h + _dense_f(p, name + ".time_emb_proj", TF.silu(temb))[:, :, None, None]
    h = TF.silu(_gn(p, name + ".norm2", h, groups, eps))
    h = _conv2d(p, name + ".conv2", h)
    if (name + ".conv_shortcut.kernel") in p:
        x = _conv2d(p, name + ".conv_shortcut", x, pad=0)
    return h + x


# Score matrices above this many elements (SD-2.1 at 96x96 latents: 5 heads x 9216^2 = 425 M per layer, ten such layers alive
# under autograd) are formed one block of query rows at a time, each block under torch.utils.checkpoint when gradients are
# recorded: softmax is row-wise, so every output row is computed by exactly the same operations as in the one-shot form; only
# the probabilities are recomputed in the backward pass instead of being kept (bounds the host memory of the full-size tests).
_ATTN_CHUNK_ELEMS = 1 << 28
_ATTN_CHUNK_ROWS = 1024


def _attention_rows(q, k, v, scale):
    return torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1) @ v


def _attention_rows_chunked(q, k, v, scale):
    from torch.utils.checkpoint import checkpoint
    outs = []
    for r0 in range(0, q.shape[2], _ATTN_CHUNK_ROWS):
        qc = q[:, :, r0:r0 + _ATTN_CHUNK_ROWS]
        if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
            outs.append(checkpoint(_attention_rows, qc, k, v, scale, use_reentrant=False))
        else:
            outs.append(_attention_rows(qc, k, v, scale))
    return torch.cat(outs, dim=2)


def _attention_f(p, name, x, ctx, heads):
    B, N, C = x.shape
    ctx = x if ctx is None else ctx
    q = _dense_f(p, name + ".to_q", x)
    k = _dense_f(p, name + ".to_k", ctx)
    v = _dense_f(p, name + ".to_v", ctx)
    d = C // heads
    sp = lambda t: t.reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    if B * heads * q.shape[2] * k.shape[2] > _ATTN_CHUNK_ELEMS:
        o = _attention_rows_chunked(q, k, v, d ** -0.5)
    else:
        s = (q @ k.transpose(-1, -2)) * (d ** -0.5)
        o = torch.softmax(s, dim=-1) @ v
    o = o.permute(0, 2, 1, 3).reshape(B, N, C)
    return _dense_f(p, name + ".to_out_0", o)


def _transformer_f(p, name, x, ctx, heads, groups, linear):
    B, C, H, W = x.shape
    res = x
    h = _gn(p, name + ".norm", x, groups, 1e-5)
    if linear:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = _dense_f(p, name + ".proj_in", h)
    else:
        h = _conv2d(p, name + ".proj_in", h, pad=0)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    b = name + ".transformer_blocks_0"
    ln = lambda n, t: TF.layer_norm(t, (C,), p[f"{b}.{n}.scale"], p[f"{b}.{n}.bias"], 1e-5)
    h = h + _attention_f(p, b + ".attn1", ln("norm1", h), None, heads)
    h = h + _attention_f(p, b + ".attn2", ln("norm2", h), ctx, heads)
    ff = _dense_f(p, b + ".ff.net_0.proj", ln("norm3", h))
    lin, gate = ff.chunk(2, dim=-1)
    h = h + _dense_f(p, b + ".ff.net_2", lin * TF.gelu(gate, approximate="tanh"))
    if linear:
        h = _dense_f(p, name + ".proj_out", h)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = _conv2d(p, name + ".proj_out", h, pad=0)
    return h + res


def timestep_embedding(t, dim):
    """get_sinusoidal_embeddings(flip_sin_to_cos=True, freq_shift=0): concat([cos, sin])."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.to(torch.float32)[:, None] * freqs[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def unet_forward(p, cfg: UNetCfg, sample, timesteps, context):
    """sample (B,C,H,W), timesteps (B,) int, context (B,L,D) -> (B,C,H,W)."""
    dt = sample.dtype
    boc = cfg.block_out_channels
    G = cfg.norm_groups
    temb = timestep_embedding(timesteps, boc[0]).to(dt)
    temb = _dense_f(p, "time_embedding.linear_2", TF.silu(_dense_f(p, "time_embedding.linear_1", temb)))
    h = _conv2d(p, "conv_in", sample)
    skips = [h]
    for i, typ in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            h = _resnet_f(p, f"down_blocks_{i}.resnets_{j}", h, temb, G, 1e-5)
            if typ == "CrossAttn":
                h = _transformer_f(p, f"down_blocks_{i}.attentions_{j}", h, context, cfg.attention_head_dim[i], G,
                                   cfg.use_linear_projection)
            skips.append(h)
        if i != len(boc) - 1:
            h = _conv2d(p, f"down_blocks_{i}.downsamplers_0.conv", h, stride=2, pad=1)
            skips.append(h)
    h = _resnet_f(p, "mid_block.resnets_0", h, temb, G, 1e-5)
    h = _transformer_f(p, "mid_block.attentions_0", h, context, cfg.attention_head_dim[-1], G, cfg.use_linear_projection)
    h = _resnet_f(p, "mid_block.resnets_1", h, temb, G, 1e-5)
    rev_heads = list(reversed(cfg.attention_head_dim))
    for i, typ in enumerate(cfg.up_block_types):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = _resnet_f(p, f"up_blocks_{i}.resnets_{j}", h, temb, G, 1e-5)
            if typ == "CrossAttn":
                h = _transformer_f(p, f"up_blocks_{i}.attentions_{j}", h, context, rev_heads[i], G, cfg.use_linear_projection)
        if i != len(boc) - 1:
            h = TF.interpolate(h, scale_factor=2, mode="nearest")
            h = _conv2d(p, f"up_blocks_{i}.upsamplers_0.conv", h)
    h = TF.silu(_gn(p, "conv_norm_out", h, G, 1e-5))
    return _conv2d(p, "conv_out", h)


def vae_decode(p, cfg: VAECfg, latents):
    """pipeline/policy_gradient.py:174-182: z/0.18215 -> decode -> (x/2+.5).clip(0,1) -> NHWC."""
    G = cfg.norm_groups
    z = latents / cfg.scaling_factor
    h = _conv2d(p, "post_quant_conv", z, pad=0)
    h = _conv2d(p, "decoder.conv_in", h)
    h = _resnet_f(p, "decoder.mid_block.resnets_0", h, None, G, 1e-6)
    a = "decoder.mid_block.attentions_0"
    B, C, H, W = h.shape
    res = h
    t = _gn(p, a + ".group_norm", h, G, 1e-6).permute(0, 2, 3, 1).reshape(B, H * W, C)
    q, k, v = (_dense_f(p, f"{a}.{n}", t) for n in ("query", "key", "value"))
    scale = 1.0 / math.sqrt(math.sqrt(C))
    s = torch.softmax((q * scale) @ (k * scale).transpose(-1, -2), dim=-1)
    t = _dense_f(p, a + ".proj_attn", s @ v)
    h = t.reshape(B, H, W, C).permute(0, 3, 1, 2) + res
    h = _resnet_f(p, "decoder.mid_block.resnets_1", h, None, G, 1e-6)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet_f(p, f"decoder.up_blocks_{i}.resnets_{j}", h, None, G, 1e-6)
        if i != n - 1:
            h = TF.interpolate(h, scale_factor=2, mode="nearest")
            h = _conv2d(p, f"decoder.up_blocks_{i}.upsamplers_0.conv", h)
    h = TF.silu(_gn(p, "decoder.conv_norm_out", h, G, 1e-6))
    img = _conv2d(p, "decoder.conv_out", h)
    return (img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)


def _vae_mid_attention(p, a, h, G):
    B, C, H, W = h.shape
    t = _gn(p, a + ".group_norm", h, G, 1e-6).permute(0, 2, 3, 1).reshape(B, H * W, C)
    q, k, v = (_dense_f(p, f"{a}.{n}", t) for n in ("query", "key", "value"))
    scale = 1.0 / math.sqrt(math.sqrt(C))
    s = torch.softmax((q * scale) @ (k * scale).transpose(-1, -2), dim=-1)
    t = _dense_f(p, a + ".proj_attn", s @ v)
    return t.reshape(B, H, W, C).permute(0, 3, 1, 2) + h


def vae_encode(p, cfg: VAECfg, images):
    """The `vae` callback of the RWR sampler (/root/reference/ddpo/training/callbacks.py:37-57): images (B,H,W,3) in [0,1] -> NCHW,
    (x - 0.5) / 0.5, FlaxAutoencoderKL.encode -> concat([mean, logvar], -1) (B,h,w,2*latent) NHWC with logvar clipped to [-30, 20]
    (FlaxDiagonalGaussianDistribution clips in its constructor).  Down-samplers pad (0,1,0,1) and convolve with stride 2, no padding."""
    G = cfg.norm_groups
    h = (images.permute(0, 3, 1, 2) - 0.5) / 0.5
    h = _conv2d(p, "encoder.conv_in", h)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = _resnet_f(p, f"encoder.down_blocks_{i}.resnets_{j}", h, None, G, 1e-6)
        if i != n - 1:
            h = _conv2d(p, f"encoder.down_blocks_{i}.downsamplers_0.conv", TF.pad(h, (0, 1, 0, 1)), stride=2, pad=0)
    h = _resnet_f(p, "encoder.mid_block.resnets_0", h, None, G, 1e-6)
    h = _vae_mid_attention(p, "encoder.mid_block.attentions_0", h, G)
    h = _resnet_f(p, "encoder.mid_block.resnets_1", h, None, G, 1e-6)
    h = TF.silu(_gn(p, "encoder.conv_norm_out", h, G, 1e-6))
    h = _conv2d(p, "encoder.conv_out", h)
    m = _conv2d(p, "quant_conv", h, pad=0).permute(0, 2, 3, 1)
    C = cfg.latent_channels
    return torch.cat([m[..., :C], m[..., C:].clamp(-30.0, 20.0)], dim=-1)
