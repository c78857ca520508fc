"""VAE decoder and encoder (diffusers 0.12.1 `FlaxAutoencoderKL.decode` / `.encode`) on the gfx950 kernels, forward only.

Decoder: replaces `vae_decode` of the reference (/root/reference/pipeline/policy_gradient.py:174-182):
    latents / 0.18215 -> post_quant_conv -> decoder -> (x / 2 + 0.5).clip(0, 1) -> NHWC
Encoder: the `vae` callback of the RWR sampler (/root/reference/ddpo/training/callbacks.py:37-57) — the posterior moments
    concat([mean, logvar]) that pipeline/sample.py stores per image and the RWR train step samples latents from.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import torch

from .. import lib as L
from .unet import Act, ParamStore, add_resnet, resnet_forward, _add_conv, _add_dense, _add_norm


@dataclass(frozen=True)
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def named(name):
        if name in ("sd", "sd15", "sd21"):
            return VAEConfig()
        if name == "tiny":
            return VAEConfig(block_out_channels=(32, 32, 64, 64))
        raise KeyError(name)


def vae_decoder_param_shapes(cfg: VAEConfig):
    d = OrderedDict()
    boc = cfg.block_out_channels
    top = boc[-1]
    _add_conv(d, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    _add_conv(d, "decoder.conv_in", cfg.latent_channels, top, 3)
    add_resnet(d, "decoder.mid_block.resnets_0", top, top, 0)
    a = "decoder.mid_block.attentions_0"
    _add_norm(d, a + ".group_norm", top)
    for n in ("query", "key", "value", "proj_attn"):
        _add_dense(d, f"{a}.{n}", top, top)
    add_resnet(d, "decoder.mid_block.resnets_1", top, top, 0)
    ch = top
    for i, out_c in enumerate(boc[::-1]):
        for j in range(cfg.layers_per_block + 1):
            add_resnet(d, f"decoder.up_blocks_{i}.resnets_{j}", ch, out_c, 0)
            ch = out_c
        if i < len(boc) - 1:
            _add_conv(d, f"decoder.up_blocks_{i}.upsamplers_0.conv", ch, ch, 3)
    _add_norm(d, "decoder.conv_norm_out", boc[0])
    _add_conv(d, "decoder.conv_out", boc[0], cfg.out_channels, 3)
    return d


class VAEDecoder:
    def __init__(self, cfg: VAEConfig, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.params = ParamStore(vae_decoder_param_shapes(cfg), self.device)
        self._out_pad = None

    def _padded_conv_out(self):
        """conv_out has 3 output channels; the GEMM wants N % 4 == 0, so keep a zero-padded (.., 4) copy (frozen weights)."""
        if self._out_pad is None:
            k = self.params["decoder.conv_out.kernel"]
            b = self.params["decoder.conv_out.bias"]
            n4 = (self.cfg.out_channels + 3) // 4 * 4
            kp = torch.zeros(*k.shape[:3], n4, dtype=torch.float32, device=self.device)
            kp[..., : k.shape[3]] = k
            bp = torch.zeros(n4, dtype=torch.float32, device=self.device)
            bp[: b.numel()] = b
            self._out_pad = (kp, bp, n4)
        return self._out_pad

    def invalidate(self):
        self._out_pad = None

    def _mid_attention(self, x: Act):
        P = self.params
        a = "decoder.mid_block.attentions_0"
        B, N, C = x.B, x.HW, x.C
        t = L.groupnorm(x.t, B, N, P[a + ".group_norm.scale"], P[a + ".group_norm.bias"], self.cfg.norm_groups, 1e-6, False)
        q = L.linear(t, P[a + ".query.kernel"], P[a + ".query.bias"])
        k = L.linear(t, P[a + ".key.kernel"], P[a + ".key.bias"])
        v = L.linear(t, P[a + ".value.kernel"], P[a + ".value.bias"])
        o = torch.empty(B * N, C, dtype=torch.float32, device=self.device)
        scores = torch.empty(N, N, dtype=torch.float32, device=self.device)
        for b in range(B):
            sl = slice(b * N, (b + 1) * N)
            L.gemm_conv(q[sl], k[sl], M=N, N=N, K=C, w_trans=True, alpha=1.0 / math.sqrt(C), out=scores)
            L.softmax_rows_(scores)
            L.gemm_conv(scores, v[sl], M=N, N=C, K=N, out=o[sl])
        out = L.linear(o, P[a + ".proj_attn.kernel"], P[a + ".proj_attn.bias"], residual=x.t)
        return Act(out, B, x.H, x.W, C)

    def decode(self, latents):
        """latents (B,4,h,w) NCHW -> images (B,8h,8w,3) NHWC in [0,1]."""
        P, cfg = self.params, self.cfg
        B, Cl, H, W = latents.shape
        G = cfg.norm_groups
        z = L.nchw_to_nhwc(latents.contiguous())
        z, _, _ = L.conv2d(z, P["post_quant_conv.kernel"], P["post_quant_conv.bias"], B, H, W, Cl, Cl, 1,
                           alpha=1.0 / cfg.scaling_factor)
        top = cfg.block_out_channels[-1]
        t, _, _ = L.conv2d(z, P["decoder.conv_in.kernel"], P["decoder.conv_in.bias"], B, H, W, Cl, top, 3)
        h = Act(t, B, H, W, top)
        h = resnet_forward(P, "decoder.mid_block.resnets_0", h, None, G, 1e-6)
        h = self._mid_attention(h)
        h = resnet_forward(P, "decoder.mid_block.resnets_1", h, None, G, 1e-6)
        n = len(cfg.block_out_channels)
        for i in range(n):
            for j in range(cfg.layers_per_block + 1):
                h = resnet_forward(P, f"decoder.up_blocks_{i}.resnets_{j}", h, None, G, 1e-6)
            if i < n - 1:
                t, OH, OW = L.conv2d(h.t, P[f"decoder.up_blocks_{i}.upsamplers_0.conv.kernel"],
                                     P[f"decoder.up_blocks_{i}.upsamplers_0.conv.bias"], B, h.H, h.W, h.C, h.C, 3, upsample=True)
                h = Act(t, B, OH, OW, h.C)
        t = L.groupnorm(h.t, B, h.HW, P["decoder.conv_norm_out.scale"], P["decoder.conv_norm_out.bias"], G, 1e-6, True)
        kp, bp, n4 = self._padded_conv_out()
        t, _, _ = L.conv2d(t, kp, bp, B, h.H, h.W, h.C, n4, 3)
        img = t[:, : cfg.out_channels].contiguous()
        img = L.scale_shift_clip(img, 0.5, 0.5, 0.0, 1.0)
        return img.view(B, h.H, h.W, cfg.out_channels)


def vae_encoder_param_shapes(cfg: VAEConfig, in_channels=3):
    d = OrderedDict()
    boc = cfg.block_out_channels
    _add_conv(d, "encoder.conv_in", in_channels, boc[0], 3)
    ch = boc[0]
    for i, out_c in enumerate(boc):
        for j in range(cfg.layers_per_block):
            add_resnet(d, f"encoder.down_blocks_{i}.resnets_{j}", ch, out_c, 0)
            ch = out_c
        if i < len(boc) - 1:
            _add_conv(d, f"encoder.down_blocks_{i}.downsamplers_0.conv", ch, ch, 3)
    add_resnet(d, "encoder.mid_block.resnets_0", ch, ch, 0)
    a = "encoder.mid_block.attentions_0"
    _add_norm(d, a + ".group_norm", ch)
    for n in ("query", "key", "value", "proj_attn"):
        _add_dense(d, f"{a}.{n}", ch, ch)
    add_resnet(d, "encoder.mid_block.resnets_1", ch, ch, 0)
    _add_norm(d, "encoder.conv_norm_out", ch)
    _add_conv(d, "encoder.conv_out", ch, 2 * cfg.latent_channels, 3)
    _add_conv(d, "quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return d


class VAEEncoder:
    """images (B,H,W,3) in [0,1] -> posterior moments (B,H/8,W/8,8) NHWC = concat([mean, clip(logvar, -30, 20)])."""

    def __init__(self, cfg: VAEConfig, device="cuda", in_channels=3):
        self.cfg = cfg
        self.device = torch.device(device)
        self.in_channels = in_channels
        self.params = ParamStore(vae_encoder_param_shapes(cfg, in_channels), self.device)
        self._in_pad = None

    def invalidate(self):
        self._in_pad = None

    def _padded_conv_in(self):
        """conv_in has 3 input channels; the loaders fetch float4 rows, so the image gets a zero 4th channel and the kernel a zero tap."""
        if self._in_pad is None:
            k = self.params["encoder.conv_in.kernel"]
            c4 = (self.in_channels + 3) // 4 * 4
            kp = torch.zeros(k.shape[0], k.shape[1], c4, k.shape[3], dtype=torch.float32, device=self.device)
            kp[:, :, : k.shape[2]] = k
            self._in_pad = (kp, c4)
        return self._in_pad

    def encode(self, images):
        P, cfg = self.params, self.cfg
        B, H, W, Ci = images.shape
        G = cfg.norm_groups
        kp, c4 = self._padded_conv_in()
        x = torch.zeros(B * H * W, c4, dtype=torch.float32, device=self.device)
        x[:, :Ci] = (images.to(self.device, torch.float32).reshape(B * H * W, Ci) - 0.5) / 0.5          # normalize(mean 0.5, std 0.5)
        boc = cfg.block_out_channels
        t, _, _ = L.conv2d(x, kp, P["encoder.conv_in.bias"], B, H, W, c4, boc[0], 3)
        h = Act(t, B, H, W, boc[0])
        n = len(boc)
        for i in range(n):
            for j in range(cfg.layers_per_block):
                h = resnet_forward(P, f"encoder.down_blocks_{i}.resnets_{j}", h, None, G, 1e-6)
            if i < n - 1:
                # FlaxDownsample2D: pad one row / column at the bottom / right, 3x3 stride-2 convolution without padding
                xp = torch.zeros(B, h.H + 1, h.W + 1, h.C, dtype=torch.float32, device=self.device)
                xp[:, : h.H, : h.W] = h.t.view(B, h.H, h.W, h.C)
                t, OH, OW = L.conv2d(xp.view(-1, h.C), P[f"encoder.down_blocks_{i}.downsamplers_0.conv.kernel"],
                                     P[f"encoder.down_blocks_{i}.downsamplers_0.conv.bias"], B, h.H + 1, h.W + 1, h.C, h.C, 3, stride=2, pad=0)
                h = Act(t, B, OH, OW, h.C)
        h = resnet_forward(P, "encoder.mid_block.resnets_0", h, None, G, 1e-6)
        h = self._mid_attention(h)
        h = resnet_forward(P, "encoder.mid_block.resnets_1", h, None, G, 1e-6)
        t = L.groupnorm(h.t, B, h.HW, P["encoder.conv_norm_out.scale"], P["encoder.conv_norm_out.bias"], G, 1e-6, True)
        c2 = 2 * cfg.latent_channels
        t, _, _ = L.conv2d(t, P["encoder.conv_out.kernel"], P["encoder.conv_out.bias"], B, h.H, h.W, h.C, c2, 3)
        m, _, _ = L.conv2d(t, P["quant_conv.kernel"], P["quant_conv.bias"], B, h.H, h.W, c2, c2, 1)
        m = m.view(B, h.H, h.W, c2)
        m[..., cfg.latent_channels:].clamp_(-30.0, 20.0)
        return m

    def _mid_attention(self, x: Act):
        P = self.params
        a = "encoder.mid_block.attentions_0"
        B, N, C = x.B, x.HW, x.C
        t = L.groupnorm(x.t, B, N, P[a + ".group_norm.scale"], P[a + ".group_norm.bias"], self.cfg.norm_groups, 1e-6, False)
        q = L.linear(t, P[a + ".query.kernel"], P[a + ".query.bias"])
        k = L.linear(t, P[a + ".key.kernel"], P[a + ".key.bias"])
        v = L.linear(t, P[a + ".value.kernel"], P[a + ".value.bias"])
        o = torch.empty(B * N, C, dtype=torch.float32, device=self.device)
        scores = torch.empty(N, N, dtype=torch.float32, device=self.device)
        for b in range(B):
            sl = slice(b * N, (b + 1) * N)
            L.gemm_conv(q[sl], k[sl], M=N, N=N, K=C, w_trans=True, alpha=1.0 / math.sqrt(C), out=scores)
            L.softmax_rows_(scores)
            L.gemm_conv(scores, v[sl], M=N, N=C, K=N, out=o[sl])
        out = L.linear(o, P[a + ".proj_attn.kernel"], P[a + ".proj_attn.bias"], residual=x.t)
        return Act(out, B, x.H, x.W, C)
