"""Stable-Diffusion U-Net (diffusers 0.12.1 `FlaxUNet2DConditionModel` architecture) executed by the gfx950 kernels.

Replaces the `self.unet.apply(...)` call sites of the reference:
  /root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:219-224  (sampling, batch 2B under CFG)
  /root/reference/ddpo/training/policy_gradient.py:87-102                          (training, cond + uncond passes)
Parameters keep the Flax tree naming / layouts (conv HWIO, dense (in,out)) so a Flax checkpoint dict loads as is;
they live in ONE flat fp32 buffer (one RCCL all-reduce, one fused AdamW launch).
Activations are NHWC rows (B*H*W, C) in fp32; every contraction runs on the exact-fp32 MFMA datapath.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import torch

from .. import lib as L


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    cross_attn_down: Tuple[bool, ...] = (True, True, True, False)
    layers_per_block: int = 2
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)        # diffusers' `attention_head_dim` (= head COUNT in 0.12.1 Flax)
    cross_attention_dim: int = 768
    use_linear_projection: bool = False
    norm_groups: int = 32
    prediction_type: str = "epsilon"

    @staticmethod
    def named(name):
        if name in ("sd15", "sd14", "sd1"):
            return UNetConfig()
        if name == "sd21":
            return UNetConfig(num_heads=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True,
                              prediction_type="v_prediction")
        if name == "tiny":
            return UNetConfig(block_out_channels=(32, 64, 128, 128), cross_attention_dim=64)
        raise KeyError(name)


class ParamStore:
    """Named fp32 tensors carved out of one flat device buffer (16-byte aligned slices)."""

    def __init__(self, shapes, device):
        self.shapes = OrderedDict(shapes)
        self.offsets = OrderedDict()
        off = 0
        for name, shp in self.shapes.items():
            self.offsets[name] = off
            off += (math.prod(shp) + 3) // 4 * 4
        self.numel = off
        self.n_params = sum(math.prod(s) for s in self.shapes.values())
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.views = {n: self.flat[o:o + math.prod(self.shapes[n])].view(self.shapes[n]) for n, o in self.offsets.items()}

    def __getitem__(self, name):
        return self.views[name]

    def __contains__(self, name):
        return name in self.views

    def load_dict(self, tree):
        """tree: {flax_name: array-like in Flax layout}.  Every parameter must be present."""
        missing = [n for n in self.shapes if n not in tree]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        for n, v in self.views.items():
            t = torch.as_tensor(tree[n], dtype=torch.float32)
            if tuple(t.shape) != tuple(self.shapes[n]):
                raise ValueError(f"{n}: expected {self.shapes[n]}, got {tuple(t.shape)}")
            v.copy_(t)

    def init_synthetic(self, seed=0):
        """Random-init weights of the right architecture (no checkpoints are reachable offline)."""
        g = torch.Generator(device=self.flat.device).manual_seed(seed)
        for n, v in self.views.items():
            if n.endswith(".kernel"):
                v.normal_(0.0, 1.0 / math.sqrt(math.prod(v.shape[:-1])), generator=g)
            elif n.endswith(".scale"):
                v.fill_(1.0)
            else:
                v.zero_()


def _add_conv(d, name, cin, cout, k):
    d[name + ".kernel"] = (k, k, cin, cout)
    d[name + ".bias"] = (cout,)


def _add_dense(d, name, cin, cout, bias=True):
    d[name + ".kernel"] = (cin, cout)
    if bias:
        d[name + ".bias"] = (cout,)


def _add_norm(d, name, c):
    d[name + ".scale"] = (c,)
    d[name + ".bias"] = (c,)


def add_resnet(d, name, cin, cout, temb_dim):
    _add_norm(d, name + ".norm1", cin)
    _add_conv(d, name + ".conv1", cin, cout, 3)
    if temb_dim:
        _add_dense(d, name + ".time_emb_proj", temb_dim, cout)
    _add_norm(d, name + ".norm2", cout)
    _add_conv(d, name + ".conv2", cout, cout, 3)
    if cin != cout:
        _add_conv(d, name + ".conv_shortcut", cin, cout, 1)


def _add_transformer(d, name, c, ctx, linear):
    _add_norm(d, name + ".norm", c)
    (_add_dense if linear else lambda dd, n, a, b: _add_conv(dd, n, a, b, 1))(d, name + ".proj_in", c, c)
    tb = name + ".transformer_blocks_0"
    for attn, kv in (("attn1", c), ("attn2", ctx)):
        _add_dense(d, f"{tb}.{attn}.to_q", c, c, bias=False)
        _add_dense(d, f"{tb}.{attn}.to_k", kv, c, bias=False)
        _add_dense(d, f"{tb}.{attn}.to_v", kv, c, bias=False)
        _add_dense(d, f"{tb}.{attn}.to_out_0", c, c)
    _add_dense(d, tb + ".ff.net_0.proj", c, 8 * c)
    _add_dense(d, tb + ".ff.net_2", 4 * c, c)
    for n in ("norm1", "norm2", "norm3"):
        _add_norm(d, f"{tb}.{n}", c)
    (_add_dense if linear else lambda dd, n, a, b: _add_conv(dd, n, a, b, 1))(d, name + ".proj_out", c, c)


def unet_param_shapes(cfg: UNetConfig):
    d = OrderedDict()
    boc = cfg.block_out_channels
    nlev = len(boc)
    temb = 4 * boc[0]
    _add_conv(d, "conv_in", cfg.in_channels, boc[0], 3)
    _add_dense(d, "time_embedding.linear_1", boc[0], temb)
    _add_dense(d, "time_embedding.linear_2", temb, temb)
    ch = boc[0]
    for i in range(nlev):
        for j in range(cfg.layers_per_block):
            add_resnet(d, f"down_blocks_{i}.resnets_{j}", ch, boc[i], temb)
            ch = boc[i]
            if cfg.cross_attn_down[i]:
                _add_transformer(d, f"down_blocks_{i}.attentions_{j}", ch, cfg.cross_attention_dim, cfg.use_linear_projection)
        if i < nlev - 1:
            _add_conv(d, f"down_blocks_{i}.downsamplers_0.conv", ch, ch, 3)
    add_resnet(d, "mid_block.resnets_0", ch, ch, temb)
    _add_transformer(d, "mid_block.attentions_0", ch, cfg.cross_attention_dim, cfg.use_linear_projection)
    add_resnet(d, "mid_block.resnets_1", ch, ch, temb)
    rev = boc[::-1]
    for i in range(nlev):
        out_c = rev[i]
        skip_last = rev[min(i + 1, nlev - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = skip_last if j == cfg.layers_per_block else out_c
            add_resnet(d, f"up_blocks_{i}.resnets_{j}", ch + skip, out_c, temb)
            ch = out_c
            if cfg.cross_attn_down[nlev - 1 - i]:
                _add_transformer(d, f"up_blocks_{i}.attentions_{j}", ch, cfg.cross_attention_dim, cfg.use_linear_projection)
        if i < nlev - 1:
            _add_conv(d, f"up_blocks_{i}.upsamplers_0.conv", ch, ch, 3)
    _add_norm(d, "conv_norm_out", boc[0])
    _add_conv(d, "conv_out", boc[0], cfg.out_channels, 3)
    return d


class Act:
    """An NHWC activation: rows (B*H*W, C)."""
    __slots__ = ("t", "B", "H", "W", "C")

    def __init__(self, t, B, H, W, C):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C

    @property
    def HW(self):
        return self.H * self.W


def resnet_forward(P, name, x: Act, temb_act, groups, eps):
    """FlaxResnetBlock2D: GN-SiLU-conv3x3 (+time proj) - GN-SiLU-conv3x3 (+ shortcut)."""
    cout = P[name + ".conv1.bias"].numel()
    h = L.groupnorm(x.t, x.B, x.HW, P[name + ".norm1.scale"], P[name + ".norm1.bias"], groups, eps, True)
    rowbias = None
    if temb_act is not None:
        rowbias = L.linear(temb_act, P[name + ".time_emb_proj.kernel"], P[name + ".time_emb_proj.bias"])
    h, _, _ = L.conv2d(h, P[name + ".conv1.kernel"], P[name + ".conv1.bias"], x.B, x.H, x.W, x.C, cout, 3,
                       rowbias=rowbias, rows_per_batch=x.HW)
    h = L.groupnorm(h, x.B, x.HW, P[name + ".norm2.scale"], P[name + ".norm2.bias"], groups, eps, True)
    res = x.t
    if (name + ".conv_shortcut.kernel") in P:
        res, _, _ = L.conv2d(x.t, P[name + ".conv_shortcut.kernel"], P[name + ".conv_shortcut.bias"], x.B, x.H, x.W, x.C, cout, 1)
    out, _, _ = L.conv2d(h, P[name + ".conv2.kernel"], P[name + ".conv2.bias"], x.B, x.H, x.W, cout, cout, 3, residual=res)
    return Act(out, x.B, x.H, x.W, cout)


class UNet2DCondition:
    def __init__(self, cfg: UNetConfig, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.params = ParamStore(unet_param_shapes(cfg), self.device)

    # -------------------------------------------------------------------------------- sub-blocks
    def _attention(self, name, x, B, N, C, heads, ctx, ctx_len):
        P = self.params
        q = L.linear(x, P[name + ".to_q.kernel"])
        kv_src = x if ctx is None else ctx
        k = L.linear(kv_src, P[name + ".to_k.kernel"])
        v = L.linear(kv_src, P[name + ".to_v.kernel"])
        return L.attention(q, k, v, B, heads, N, N if ctx is None else ctx_len, C // heads)

    def _transformer(self, name, x: Act, ctx, ctx_len, heads):
        P, cfg = self.params, self.cfg
        C, B, N = x.C, x.B, x.HW
        h = L.groupnorm(x.t, B, N, P[name + ".norm.scale"], P[name + ".norm.bias"], cfg.norm_groups, 1e-5, False)
        if cfg.use_linear_projection:
            h = L.linear(h, P[name + ".proj_in.kernel"], P[name + ".proj_in.bias"])
        else:
            h, _, _ = L.conv2d(h, P[name + ".proj_in.kernel"], P[name + ".proj_in.bias"], B, x.H, x.W, C, C, 1)
        tb = name + ".transformer_blocks_0"
        ln = lambda n, t: L.layernorm(t, P[f"{tb}.{n}.scale"], P[f"{tb}.{n}.bias"], 1e-5)
        a = self._attention(tb + ".attn1", ln("norm1", h), B, N, C, heads, None, 0)
        h = L.linear(a, P[tb + ".attn1.to_out_0.kernel"], P[tb + ".attn1.to_out_0.bias"], residual=h)
        a = self._attention(tb + ".attn2", ln("norm2", h), B, N, C, heads, ctx, ctx_len)
        h = L.linear(a, P[tb + ".attn2.to_out_0.kernel"], P[tb + ".attn2.to_out_0.bias"], residual=h)
        f = L.linear(ln("norm3", h), P[tb + ".ff.net_0.proj.kernel"], P[tb + ".ff.net_0.proj.bias"])
        f = L.geglu(f)
        h = L.linear(f, P[tb + ".ff.net_2.kernel"], P[tb + ".ff.net_2.bias"], residual=h)
        if cfg.use_linear_projection:
            out = L.linear(h, P[name + ".proj_out.kernel"], P[name + ".proj_out.bias"], residual=x.t)
        else:
            out, _, _ = L.conv2d(h, P[name + ".proj_out.kernel"], P[name + ".proj_out.bias"], B, x.H, x.W, C, C, 1, residual=x.t)
        return Act(out, B, x.H, x.W, C)

    # -------------------------------------------------------------------------------- forward
    def forward(self, sample, timesteps, context):
        """sample (B,C,H,W) fp32 NCHW; timesteps (B,) int32; context (B,L,D) fp32 -> (B,C_out,H,W)."""
        P, cfg = self.params, self.cfg
        B, Cin, H, W = sample.shape
        boc = cfg.block_out_channels
        nlev = len(boc)
        G = cfg.norm_groups
        Lc = context.shape[1]
        ctx = context.reshape(B * Lc, context.shape[2]).contiguous()
        timesteps = timesteps.to(torch.int32)

        temb = L.timestep_embedding(timesteps, boc[0])
        temb = L.linear(temb, P["time_embedding.linear_1.kernel"], P["time_embedding.linear_1.bias"])
        temb = L.linear(L.silu(temb), P["time_embedding.linear_2.kernel"], P["time_embedding.linear_2.bias"])
        temb_act = L.silu(temb)          # every ResBlock applies SiLU before its time_emb_proj

        x = L.nchw_to_nhwc(sample.contiguous())
        t, _, _ = L.conv2d(x, P["conv_in.kernel"], P["conv_in.bias"], B, H, W, Cin, boc[0], 3)
        h = Act(t, B, H, W, boc[0])
        skips = [h]
        for i in range(nlev):
            for j in range(cfg.layers_per_block):
                h = resnet_forward(P, f"down_blocks_{i}.resnets_{j}", h, temb_act, G, 1e-5)
                if cfg.cross_attn_down[i]:
                    h = self._transformer(f"down_blocks_{i}.attentions_{j}", h, ctx, Lc, cfg.num_heads[i])
                skips.append(h)
            if i < nlev - 1:
                t, OH, OW = L.conv2d(h.t, P[f"down_blocks_{i}.downsamplers_0.conv.kernel"],
                                     P[f"down_blocks_{i}.downsamplers_0.conv.bias"], B, h.H, h.W, h.C, h.C, 3, stride=2, pad=1)
                h = Act(t, B, OH, OW, h.C)
                skips.append(h)
        h = resnet_forward(P, "mid_block.resnets_0", h, temb_act, G, 1e-5)
        h = self._transformer("mid_block.attentions_0", h, ctx, Lc, cfg.num_heads[-1])
        h = resnet_forward(P, "mid_block.resnets_1", h, temb_act, G, 1e-5)
        for i in range(nlev):
            lvl = nlev - 1 - i
            for j in range(cfg.layers_per_block + 1):
                s = skips.pop()
                cat = torch.empty(B * h.HW, h.C + s.C, dtype=torch.float32, device=self.device)
                L.copy_cols(h.t, cat, 0, B * h.HW, h.C)
                L.copy_cols(s.t, cat, h.C, B * h.HW, s.C)
                h = resnet_forward(P, f"up_blocks_{i}.resnets_{j}", Act(cat, B, h.H, h.W, h.C + s.C), temb_act, G, 1e-5)
                if cfg.cross_attn_down[lvl]:
                    h = self._transformer(f"up_blocks_{i}.attentions_{j}", h, ctx, Lc, cfg.num_heads[lvl])
            if i < nlev - 1:
                t, OH, OW = L.conv2d(h.t, P[f"up_blocks_{i}.upsamplers_0.conv.kernel"], P[f"up_blocks_{i}.upsamplers_0.conv.bias"],
                                     B, h.H, h.W, h.C, h.C, 3, upsample=True)
                h = Act(t, B, OH, OW, h.C)
        t = L.groupnorm(h.t, B, h.HW, P["conv_norm_out.scale"], P["conv_norm_out.bias"], G, 1e-5, True)
        t, _, _ = L.conv2d(t, P["conv_out.kernel"], P["conv_out.bias"], B, h.H, h.W, h.C, cfg.out_channels, 3)
        return L.nhwc_to_nchw(t, B, cfg.out_channels, h.H, h.W)

    __call__ = forward
