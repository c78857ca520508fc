"""Stable-Diffusion U-Net (diffusers 0.12.1 `FlaxUNet2DConditionModel` architecture) executed by the gfx950 kernels.

Replaces the `self.unet.apply(...)` call sites of the reference:
  /root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:219-224  (sampling, batch 2B under CFG)
  /root/reference/ddpo/training/policy_gradient.py:87-102                          (training, cond + uncond passes)
Parameters keep the Flax tree naming / layouts (conv HWIO, dense (in,out)) so a Flax checkpoint dict loads as is;
they live in ONE flat fp32 buffer (one RCCL all-reduce, one fused AdamW launch).
Activations are NHWC rows (B*H*W, C) in fp32; contractions run on the datapath selected by `lib.DATAPATH` (exact-fp32 MFMA, or
bf16 MFMA with the fp32 operands split into hi + lo planes: weights registered by `ParamStore.pack_bf16`).
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import os

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
        if name == "tiny21":   # SD-2.1-shaped toy: linear projections, v-prediction, head dim 16 at every level
            return UNetConfig(block_out_channels=(32, 64, 128, 128), num_heads=(2, 4, 8, 8), cross_attention_dim=96,
                              use_linear_projection=True, prediction_type="v_prediction")
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
        self.fused_qkv = {}               # "<…>.attn1." -> (K, 3C) fp32 copy [to_q | to_k | to_v] of a self-attention's projections (pack_bf16)

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

    def pack_bf16(self, bwd=True):
        """(Re)build the bf16 hi/lo planes of every contraction weight for the bf16 MFMA datapaths (lib.DATAPATH).
        Call after loading weights and after every optimizer update."""
        for n, v in self.views.items():
            if n.endswith(".kernel"):
                L.pack_weights(v, bwd=bwd)
                if n.endswith(".ff.net_0.proj.kernel"):          # GEGLU feed-forward: extra planes for the fused forward
                    L.pack_weights_geglu(v, self.views[n[:-len("kernel")] + "bias"])
                if QKV_FUSED and n.endswith(".attn1.to_q.kernel"):
                    # self-attention of the SAMPLING forward: q, k, v as ONE (K, 3C) projection of the LayerNorm output (round 5).  Three launches
                    # of N = C columns become one of 3C: at the 16x16 level (M = 4096 at batch 16) that is 128 x 320 tiles instead of 128 x 64
                    # ones — 2.1x fewer operand bytes through the L2 -> LDS stream these short reductions are bound by — and two kernel
                    # boundaries less everywhere.  Same products in the same k order per column; the bits equal the three projections' wherever both
                    # shapes take the same split-K decision (the tile / split choice depends on the column count: at M = 1024, K = N = 1280
                    # the separate projections run 128 x 64 tiles with the reduction split in three, the fused one unsplit — fp32 summation
                    # order differs there, ~1e-7 relative; tests/test_gpu_model.py holds both to the tolerance).  The training forward keeps the
                    # three launches (its backward wants q, k, v as tensors), so sampler and training forward differ by that much in the
                    # self-attention inputs at such shapes — far below the 7e-5 margin of a first-update ratio to the clip boundary.
                    pre = n[:-len("to_q.kernel")]
                    parts = [v, self.views[pre + "to_k.kernel"], self.views[pre + "to_v.kernel"]]
                    buf = self.fused_qkv.get(pre)
                    if buf is None:
                        buf = self.fused_qkv[pre] = torch.empty(v.shape[0], sum(t.shape[1] for t in parts), dtype=torch.float32, device=v.device)
                    torch.cat(parts, dim=1, out=buf)
                    L.pack_weights(buf, bwd=False)

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


# skip concatenations of the sampling forward are formed in place by the producers (UNet2DCondition.forward); DDPO_SKIP_INPLACE=0: copies
SKIP_INPLACE = os.environ.get("DDPO_SKIP_INPLACE", "1") == "1"
# sampling: the second feed-forward GEMM hands its result to proj_out as planes (see _transformer); 0 restores the fp32 hand-over (A/B switch)
H3_PLANES = os.environ.get("DDPO_H3_PLANES", "1") == "1"
# sampling: the attention kernels hand their result to to_out as planes (see _attention); 0 restores the fp32 hand-over (A/B switch)
ATTN_PLANES = os.environ.get("DDPO_ATTN_PLANES", "1") == "1"
QKV_FUSED = os.environ.get("DDPO_QKV_FUSED", "1") == "1"       # sampling self-attention: q / k / v as one (K, 3C) projection (ParamStore.pack_bf16)


class Act:
    """An NHWC activation: rows (B*H*W, C).  `pl`: the same values as bf16 hi / lo planes when the producing GEMM's output stage
    also emitted them (sampling only) — a plane-fed consumer (down / up-sampler convolution) reads those instead of `t`."""
    __slots__ = ("t", "B", "H", "W", "C", "pl")

    def __init__(self, t, B, H, W, C, pl=None):
        self.t, self.B, self.H, self.W, self.C, self.pl = t, B, H, W, C, pl

    @property
    def ld(self):
        """Row stride of `t` in elements when it is a column slice of a wider row-major buffer (a skip tensor stored inside the concat
        buffer of the up block that consumes it), else None."""
        return int(self.t.stride(0)) if (self.t.dim() == 2 and self.t.shape[0] > 1 and self.t.stride(0) != self.C) else None

    @property
    def HW(self):
        return self.H * self.W

    @property
    def M(self):
        return self.B * self.H * self.W


def resnet_forward(P, name, x: Act, temb_act, groups, eps, tape=None, emit_planes=0, dest=None):
    """FlaxResnetBlock2D: GN-SiLU-conv3x3 (+time proj) - GN-SiLU-conv3x3 (+ shortcut).
    emit_planes (sampling; the consuming sampler convolution's planes_pay value): conv2's output stage also writes the block output as
    planes of that format (Act.pl).
    dest (sampling): a row-strided (rows, cout) view the block output is written into — its column range of the concat buffer of the up
    block that will consume it (UNet2DCondition.forward); x.t may be such a view too (GroupNorm / shortcut / residual take row strides)."""
    cout = P[name + ".conv1.bias"].numel()
    ldx = x.ld
    okw = {} if dest is None else dict(out=dest, ld_out=int(dest.stride(0)))
    # inference / sampling (no tape): the two GroupNorm+SiLU results feed only their convolution, so they are written as bf16
    # hi / lo planes and the convolutions run plane-fed (LDS-DMA operands; bit-identical to the fp32-fed kernels)
    # training (tape): the same, the planes are what the weight gradients of conv1 / conv2 read (lib.TRAIN_PLANES)
    pl1 = L.norm_planes(P[name + ".conv1.kernel"], x.C, x.M, tape is not None)         # 0 fp32 / 1 bf16 planes / 2 f16mx planes
    pl2 = L.norm_planes(P[name + ".conv2.kernel"], cout, x.M, tape is not None)
    h1, st1 = L.groupnorm(x.t, x.B, x.HW, P[name + ".norm1.scale"], P[name + ".norm1.bias"], groups, eps, True, return_stats=True,
                          planes=pl1)
    rowbias, rpb = None, x.HW
    if isinstance(temb_act, dict):                 # sampling: this step's time projections were computed once for the whole call
        rowbias, rpb = temb_act[name], x.M         # (1, cout): ONE row shared by every sample of the batch (same timestep)
    elif temb_act is not None:
        rowbias = L.linear(temb_act, P[name + ".time_emb_proj.kernel"], P[name + ".time_emb_proj.bias"])
    c1, _, _ = L.conv2d(h1, P[name + ".conv1.kernel"], P[name + ".conv1.bias"], x.B, x.H, x.W, x.C, cout, 3,
                        rowbias=rowbias, rows_per_batch=rpb)
    h2, st2 = L.groupnorm(c1, x.B, x.HW, P[name + ".norm2.scale"], P[name + ".norm2.bias"], groups, eps, True, return_stats=True,
                          planes=pl2)
    res = x.t
    shortcut = (name + ".conv_shortcut.kernel") in P
    if shortcut:
        res, _, _ = L.conv2d(x.t, P[name + ".conv_shortcut.kernel"], P[name + ".conv_shortcut.bias"], x.B, x.H, x.W, x.C, cout, 1,
                             **({} if ldx is None else dict(ld_src=ldx)))
    elif ldx is not None:
        okw["ld_res"] = ldx
    opl = None
    if emit_planes and tape is None and L.planes_out_ok(P[name + ".conv2.kernel"], cout, x.M, cout):
        (out, opl), _, _ = L.conv2d(h2, P[name + ".conv2.kernel"], P[name + ".conv2.bias"], x.B, x.H, x.W, cout, cout, 3, residual=res,
                                    planes_out="both", planes_fmt=emit_planes, **okw)
    else:
        out, _, _ = L.conv2d(h2, P[name + ".conv2.kernel"], P[name + ".conv2.bias"], x.B, x.H, x.W, cout, cout, 3, residual=res, **okw)
    if tape is not None:
        tape.append(("resnet", dict(name=name, x=x, st1=st1, h1=h1, c1=c1, st2=st2, h2=h2, cout=cout, shortcut=shortcut,
                                    temb=temb_act is not None, groups=groups)))
    return Act(out, x.B, x.H, x.W, cout, opl)


def resnet_backward(P, G, r, d_out, tctx):
    """Returns d_x (rows of x).  Parameter grads are accumulated into G; the time-embedding grad into tctx."""
    name, x, cout = r["name"], r["x"], r["cout"]
    B, H, W, HW = x.B, x.H, x.W, x.HW
    # out = conv2(h2) + res
    L.conv2d_wgrad(r["h2"], d_out, G[name + ".conv2.kernel"], B, H, W, cout, cout, 3, dbias=G[name + ".conv2.bias"])
    d_h2 = L.conv2d_dgrad(d_out, P[name + ".conv2.kernel"], B, H, W, cout, cout, 3)
    if r["shortcut"]:
        L.conv2d_wgrad(x.t, d_out, G[name + ".conv_shortcut.kernel"], B, H, W, x.C, cout, 1, dbias=G[name + ".conv_shortcut.bias"])
        d_res = L.conv2d_dgrad(d_out, P[name + ".conv_shortcut.kernel"], B, H, W, x.C, cout, 1)
    else:
        d_res = d_out
    d_c1 = L.groupnorm_bwd(r["c1"], d_h2, r["st2"], P[name + ".norm2.scale"], B, HW, r["groups"], True,
                           G[name + ".norm2.scale"], G[name + ".norm2.bias"])
    L.conv2d_wgrad(r["h1"], d_c1, G[name + ".conv1.kernel"], B, H, W, x.C, cout, 3)
    if r["temb"]:
        d_tproj = torch.zeros(B, cout, dtype=torch.float32, device=d_c1.device)
        L.colsum_accum(d_c1, d_tproj, rows_per_seg=HW)                      # d(time_emb_proj output)[b] = sum over pixels of b
        L.colsum_accum(d_tproj, G[name + ".conv1.bias"])                    # conv1 bias sees the same sums
        L.linear_wgrad(tctx["temb_act"], d_tproj, G[name + ".time_emb_proj.kernel"], dbias=G[name + ".time_emb_proj.bias"])
        tctx["d_temb_act"] = L.linear_dgrad(d_tproj, P[name + ".time_emb_proj.kernel"], residual=tctx["d_temb_act"])
    else:
        L.colsum_accum(d_c1, G[name + ".conv1.bias"])
    d_h1 = L.conv2d_dgrad(d_c1, P[name + ".conv1.kernel"], B, H, W, x.C, cout, 3)
    return L.groupnorm_bwd(x.t, d_h1, r["st1"], P[name + ".norm1.scale"], B, HW, r["groups"], True,
                           G[name + ".norm1.scale"], G[name + ".norm1.bias"], dx_add=d_res)


class UNet2DCondition:
    def __init__(self, cfg: UNetConfig, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.params = ParamStore(unet_param_shapes(cfg), self.device)
        self.grads = None
        self._ctx_kv = {}                 # (cross-attention name, context rows) -> (K, V) of the text context (precompute_context)
        self._ctx_kv_active = False
        self._temb = None                 # time-projection table of a sampling call (precompute_timesteps)
        self._temb_active = False

    def ensure_grads(self):
        """Flat fp32 gradient-accumulation buffer with the parameter layout (AccumulatingTrainState.grad_acc)."""
        if self.grads is None:
            self.grads = ParamStore(self.params.shapes, self.device)
        return self.grads

    # -------------------------------------------------------------------------------- attention
    def _attention(self, name, x, B, N, C, heads, ctx, ctx_len, rec=None):
        """Returns the attention output in front of to_out: fp32 (rows, C), or — sampling, where to_out is faster plane-fed (the 64x64 level:
        103 -> 62 us per launch in the model, profiles/r04_timeline_sampling_step.txt) — the bf16 hi / lo planes the attention kernel's output
        stage writes instead (the same values, so the block's result does not change by a bit)."""
        P = self.params
        po = bool(rec is None and ATTN_PLANES and L.PLANES_OUT and L.attention_planes_ok(C // heads) and
                  L.planes_pay(P[name + ".to_out_0.kernel"], C, B * N) == 1)
        fq = P.fused_qkv.get(name + ".") if (QKV_FUSED and ctx is None and rec is None and L.current_datapath() != "fp32") else None
        if fq is not None and L.PACKED.get(fq.data_ptr()) is not None:      # sampling self-attention: one projection launch for q, k, v
            qkv = L.linear(x, fq)
            return L.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, heads, N, N, C // heads, planes_out=po, ldq=3 * C, ldk=3 * C, ldv=3 * C)
        q = L.linear(x, P[name + ".to_q.kernel"])
        cached = self._ctx_kv.get((name, ctx.shape[0])) if (ctx is not None and rec is None and self._ctx_kv_active) else None
        if cached is not None:                     # text-context K / V were projected once for this sampling call
            k, v = cached[0], cached[1]
            if len(cached) > 2 and L.kv_images_valid(cached[2]):      # ... and packed once into the attention kernels' K / V^T images (under THIS datapath)
                return L.attention_from_images(q, cached[2], B, heads, N, ctx_len, C // heads, planes_out=po)
        else:
            kv_src = x if ctx is None else ctx
            k = L.linear(kv_src, P[name + ".to_k.kernel"])
            v = L.linear(kv_src, P[name + ".to_v.kernel"])
        Nk = N if ctx is None else ctx_len
        if rec is None:
            return L.attention(q, k, v, B, heads, N, Nk, C // heads, planes_out=po)
        o, lse = L.attention(q, k, v, B, heads, N, Nk, C // heads, return_lse=True)
        rec.update(q=q, k=k, v=v, o=o, lse=lse, Nk=Nk)
        return o

    def _attention_backward(self, name, rec, x_in, kv_in, d_o, B, N, C, heads, self_attn):
        """d_o: grad of the attention output (before to_out).  Returns the grad w.r.t. the (normed) attention input."""
        P, G = self.params, self.grads
        dq, dk, dv = L.attention_bwd(rec["q"], rec["k"], rec["v"], rec["o"], d_o, rec["lse"], B, heads, N, rec["Nk"], C // heads)
        L.linear_wgrad(x_in, dq, G[name + ".to_q.kernel"])
        L.linear_wgrad(kv_in, dk, G[name + ".to_k.kernel"])
        L.linear_wgrad(kv_in, dv, G[name + ".to_v.kernel"])
        d_x = L.linear_dgrad(dq, P[name + ".to_q.kernel"])
        if self_attn:
            d_x = L.linear_dgrad(dk, P[name + ".to_k.kernel"], residual=d_x)
            d_x = L.linear_dgrad(dv, P[name + ".to_v.kernel"], residual=d_x)
        return d_x

    def _transformer(self, name, x: Act, ctx, ctx_len, heads, tape=None, emit_planes=0, dest=None):
        """dest / strided x.t: as in resnet_forward (sampling: the block output lands in its consumer's concat buffer)."""
        P, cfg = self.params, self.cfg
        C, B, N = x.C, x.B, x.HW
        rec = None if tape is None else dict(name=name, x=x, heads=heads, ctx=ctx, ctx_len=ctx_len)
        tb = name + ".transformer_blocks_0"
        inf = tape is None                         # plane-fed GEMMs behind the norms (see resnet_forward)
        npf = lambda n: L.norm_planes(P[n], C, B * N, not inf)      # what the norm in front of layer n emits: 0 fp32 / 1 bf16 planes / 2 f16mx planes
        pl_in = npf(name + ".proj_in.kernel")
        pl_1 = min(npf(f"{tb}.attn1.{n}.kernel") for n in ("to_q", "to_k", "to_v"))     # one LayerNorm feeds all three: the same format or fp32
        if len({npf(f"{tb}.attn1.{n}.kernel") for n in ("to_q", "to_k", "to_v")}) > 1:
            pl_1 = 0
        pl_2 = npf(tb + ".attn2.to_q.kernel")
        pl_3 = npf(tb + ".ff.net_0.proj.kernel")
        if inf and pl_3 != 1 and L.geglu_tall_pays(P[tb + ".ff.net_0.proj.kernel"], B * N):
            pl_3 = 1                               # FF1 + GEGLU on the 256 x 320 tile is plane-fed: norm3 writes bf16 hi / lo planes
        F = P[tb + ".ff.net_2.kernel"].shape[0]
        pl_ff2 = inf and L.PLANES_OUT and L.planes_pay(P[tb + ".ff.net_2.kernel"], F, B * N)       # GEGLU output stage -> planes -> plane-fed FF2
        pl_out = inf and emit_planes and L.planes_out_ok(P[name + ".proj_out.kernel"], C, B * N, C)
        # sampling: h3 is read by proj_out only — FF2's output stage writes it as planes ONLY where proj_out is faster plane-fed (the 64x64 level:
        # 103 -> 62 us per launch in the model, profiles/r04_timeline_sampling_step.txt), and no fp32 copy exists
        pl_h3 = L.planes_pay(P[name + ".proj_out.kernel"], C, B * N) if inf and L.PLANES_OUT and H3_PLANES else 0
        if pl_h3 and not L.planes_out_ok(P[tb + ".ff.net_2.kernel"], F, B * N, C):
            pl_h3 = 0
        hn, st = L.groupnorm(x.t, B, N, P[name + ".norm.scale"], P[name + ".norm.bias"], cfg.norm_groups, 1e-5, False, return_stats=True,
                             planes=pl_in)
        if cfg.use_linear_projection:
            h0 = L.linear(hn, P[name + ".proj_in.kernel"], P[name + ".proj_in.bias"])
        else:
            h0, _, _ = L.conv2d(hn, P[name + ".proj_in.kernel"], P[name + ".proj_in.bias"], B, x.H, x.W, C, C, 1)
        ln = lambda n, t, pl=False: L.layernorm(t, P[f"{tb}.{n}.scale"], P[f"{tb}.{n}.bias"], 1e-5, planes=pl)
        a1r = None if rec is None else {}
        a2r = None if rec is None else {}
        l1 = ln("norm1", h0, pl_1)
        a1 = self._attention(tb + ".attn1", l1, B, N, C, heads, None, 0, a1r)
        h1 = L.linear(a1, P[tb + ".attn1.to_out_0.kernel"], P[tb + ".attn1.to_out_0.bias"], residual=h0)
        l2 = ln("norm2", h1, pl_2)
        a2 = self._attention(tb + ".attn2", l2, B, N, C, heads, ctx, ctx_len, a2r)
        h2 = L.linear(a2, P[tb + ".attn2.to_out_0.kernel"], P[tb + ".attn2.to_out_0.bias"], residual=h1)
        l3 = ln("norm3", h2, pl_3)
        f = None
        # sampling: GEGLU fused into the GEMM epilogue, written as planes when FF2 can take them
        if tape is None:
            gg = L.linear_geglu(l3, P[tb + ".ff.net_0.proj.kernel"], planes_out=pl_ff2)
        else:                                      # training: the same fused launch also stores the pre-activation the GEGLU backward reads
            r_ = L.linear_geglu(l3, P[tb + ".ff.net_0.proj.kernel"], pre_out=True)
            gg, f = r_ if r_ is not None else (None, None)
        if gg is None:
            f = L.linear(l3, P[tb + ".ff.net_0.proj.kernel"], P[tb + ".ff.net_0.proj.bias"])
            gg = L.geglu(f)
        h3 = L.linear(gg, P[tb + ".ff.net_2.kernel"], P[tb + ".ff.net_2.bias"], residual=h2,
                      **(dict(planes_out="only", planes_fmt=pl_h3) if pl_h3 else {}))
        po = dict(planes_out="both", planes_fmt=emit_planes) if pl_out else {}
        if dest is not None:
            po.update(out=dest, ld_out=int(dest.stride(0)))
        if x.ld is not None:
            po["ld_res"] = x.ld
        if cfg.use_linear_projection:
            out = L.linear(h3, P[name + ".proj_out.kernel"], P[name + ".proj_out.bias"], residual=x.t, **po)
        else:
            out, _, _ = L.conv2d(h3, P[name + ".proj_out.kernel"], P[name + ".proj_out.bias"], B, x.H, x.W, C, C, 1, residual=x.t, **po)
        out, opl = out if pl_out else (out, None)
        if rec is not None:
            rec.update(st=st, hn=hn, h0=h0, l1=l1, a1r=a1r, h1=h1, l2=l2, a2r=a2r, h2=h2, l3=l3, f=f, gg=gg, h3=h3)
            tape.append(("transformer", rec))
        return Act(out, B, x.H, x.W, C, opl)

    def _transformer_backward(self, r, d_out):
        P, G, cfg = self.params, self.grads, self.cfg
        name, x, heads = r["name"], r["x"], r["heads"]
        C, B, N, H, W = x.C, x.B, x.HW, x.H, x.W
        tb = name + ".transformer_blocks_0"
        gw = lambda n: G[n + ".kernel"]
        # out = proj_out(h3) + x
        if cfg.use_linear_projection:
            L.linear_wgrad(r["h3"], d_out, gw(name + ".proj_out"), dbias=G[name + ".proj_out.bias"])
            d_h3 = L.linear_dgrad(d_out, P[name + ".proj_out.kernel"])
        else:
            L.conv2d_wgrad(r["h3"], d_out, gw(name + ".proj_out"), B, H, W, C, C, 1, dbias=G[name + ".proj_out.bias"])
            d_h3 = L.conv2d_dgrad(d_out, P[name + ".proj_out.kernel"], B, H, W, C, C, 1)
        # h3 = ff2(geglu(ff1(LN3(h2)))) + h2
        L.linear_wgrad(r["gg"], d_h3, gw(tb + ".ff.net_2"), dbias=G[tb + ".ff.net_2.bias"])
        d_gg = L.linear_dgrad(d_h3, P[tb + ".ff.net_2.kernel"])
        d_f = L.geglu_bwd(r["f"], d_gg)
        L.linear_wgrad(r["l3"], d_f, gw(tb + ".ff.net_0.proj"), dbias=G[tb + ".ff.net_0.proj.bias"])
        d_l3 = L.linear_dgrad(d_f, P[tb + ".ff.net_0.proj.kernel"])
        d_h2 = L.layernorm_bwd(r["h2"], d_l3, P[tb + ".norm3.scale"], G[tb + ".norm3.scale"], G[tb + ".norm3.bias"], 1e-5, dx_add=d_h3)
        # h2 = to_out(attn2(LN2(h1), ctx)) + h1
        L.linear_wgrad(r["a2r"]["o"], d_h2, gw(tb + ".attn2.to_out_0"), dbias=G[tb + ".attn2.to_out_0.bias"])
        d_a2 = L.linear_dgrad(d_h2, P[tb + ".attn2.to_out_0.kernel"])
        d_l2 = self._attention_backward(tb + ".attn2", r["a2r"], r["l2"], r["ctx"], d_a2, B, N, C, heads, False)
        d_h1 = L.layernorm_bwd(r["h1"], d_l2, P[tb + ".norm2.scale"], G[tb + ".norm2.scale"], G[tb + ".norm2.bias"], 1e-5, dx_add=d_h2)
        # h1 = to_out(attn1(LN1(h0))) + h0
        L.linear_wgrad(r["a1r"]["o"], d_h1, gw(tb + ".attn1.to_out_0"), dbias=G[tb + ".attn1.to_out_0.bias"])
        d_a1 = L.linear_dgrad(d_h1, P[tb + ".attn1.to_out_0.kernel"])
        d_l1 = self._attention_backward(tb + ".attn1", r["a1r"], r["l1"], r["l1"], d_a1, B, N, C, heads, True)
        d_h0 = L.layernorm_bwd(r["h0"], d_l1, P[tb + ".norm1.scale"], G[tb + ".norm1.scale"], G[tb + ".norm1.bias"], 1e-5, dx_add=d_h1)
        # h0 = proj_in(GN(x))
        if cfg.use_linear_projection:
            L.linear_wgrad(r["hn"], d_h0, gw(name + ".proj_in"), dbias=G[name + ".proj_in.bias"])
            d_hn = L.linear_dgrad(d_h0, P[name + ".proj_in.kernel"])
        else:
            L.conv2d_wgrad(r["hn"], d_h0, gw(name + ".proj_in"), B, H, W, C, C, 1, dbias=G[name + ".proj_in.bias"])
            d_hn = L.conv2d_dgrad(d_h0, P[name + ".proj_in.kernel"], B, H, W, C, C, 1)
        return L.groupnorm_bwd(x.t, d_hn, r["st"], P[name + ".norm.scale"], B, N, cfg.norm_groups, False,
                               G[name + ".norm.scale"], G[name + ".norm.bias"], dx_add=d_out)

    # -------------------------------------------------------------------------------- forward
    def forward(self, sample, timesteps, context, tape=None, cfg_dup=False):
        """sample (B,C,H,W) fp32 NCHW; timesteps (B,) int32; context (B,L,D) fp32 -> (B,C_out,H,W).
        With `tape` (a list) every layer records what `backward` needs.
        `cfg_dup`: the caller guarantees that the second half of `sample` / `timesteps` repeats the first (classifier-free
        guidance feeds [x; x]): everything in front of the first cross-attention (conv_in, the first ResBlock) is then
        computed on one half and its rows duplicated — bit-identical, because no kernel depends on batch composition."""
        P, cfg = self.params, self.cfg
        B, Cin, H, W = sample.shape
        boc = cfg.block_out_channels
        nlev = len(boc)
        if H % (1 << (nlev - 1)) or W % (1 << (nlev - 1)) or Cin != cfg.in_channels:
            raise ValueError(f"latent size {H}x{W} must be divisible by {1 << (nlev - 1)} (the U-Net halves it {nlev - 1} times) "
                             f"and carry {cfg.in_channels} channels")
        if context.shape[0] != B or context.shape[2] != cfg.cross_attention_dim or timesteps.shape[0] != B:
            raise ValueError("context must be (B, L, cross_attention_dim) and timesteps (B,)")
        G = cfg.norm_groups
        Lc = context.shape[1]
        ctx = context.reshape(B * Lc, context.shape[2]).contiguous()
        timesteps = timesteps.to(torch.int32)

        shared_t = self._temb_active and tape is None
        if shared_t:
            # sampling: every sample of the batch is at the same timestep and the time path depends on nothing else — its 27
            # launches per step (embedding MLP + one projection per ResBlock) were run once for all steps by precompute_timesteps;
            # select_timestep() put this step's row into the static buffer the views below point at
            temb_act = self._temb["views"]
        else:
            emb = L.timestep_embedding(timesteps, boc[0])
            t1 = L.linear(emb, P["time_embedding.linear_1.kernel"], P["time_embedding.linear_1.bias"])
            s1 = L.silu(t1)
            temb = L.linear(s1, P["time_embedding.linear_2.kernel"], P["time_embedding.linear_2.bias"])
            temb_act = L.silu(temb)          # every ResBlock applies SiLU before its time_emb_proj

        dup = bool(cfg_dup) and tape is None and B % 2 == 0 and cfg.cross_attn_down[0]
        Bh = B // 2 if dup else B
        def twice(a):
            """[a; a] along the rows (the CFG halves of a tensor computed once), by the engine's own copy kernel: no torch op inside the
            captured step."""
            if not dup:
                return a
            n, c = a.shape
            out2 = torch.empty(2 * n, c, dtype=a.dtype, device=a.device)
            L.copy_cols(a, out2[:n], 0, n, c)
            L.copy_cols(a, out2[n:], 0, n, c)
            return out2
        # Skip concatenation without copies (sampling; round 4): every tensor of the down path that is also a skip connection is written by its
        # producer straight into ITS column range of the concat buffer of the up block that consumes it, and so is the up path's running
        # activation (column range 0 .. c0) — the two ddpo_copy_cols launches per up block (1248 per 50-step sampling call, re-reading and
        # re-writing both halves) disappear.  Same kernels, same arithmetic: only row strides change (bit-identical, tests/test_gpu_model.py).
        # The training forward (tape) keeps contiguous tensors: its backward kernels take them.
        inplace = tape is None and SKIP_INPLACE
        plan = self._concat_plan() if inplace else None              # (c0, c1) per up block, in consumption order
        cat_bufs = []                                                  # concat buffers of the pushed skips, parallel to `skips`

        def skip_dest(rows):
            """Destination view (rows, c1) of the NEXT skip to be pushed, inside a fresh concat buffer of its consumer."""
            if not inplace:
                return None
            c0, c1 = plan[len(plan) - 1 - len(cat_bufs)]
            buf = torch.empty(rows, c0 + c1, dtype=torch.float32, device=self.device)
            cat_bufs.append(buf)
            return buf[:, c0:]

        def up_dest(k):
            """Destination view (rows, c0) for the activation that up block number k (consumption order) concatenates in front of its skip."""
            if not inplace or k >= len(plan):
                return None
            return cat_bufs[-1][:, :plan[k][0]]

        x = L.nchw_to_nhwc(sample[:Bh].contiguous())
        d0 = skip_dest(B * H * W)
        if d0 is not None:
            t, _, _ = L.conv2d(x, P["conv_in.kernel"], P["conv_in.bias"], Bh, H, W, Cin, boc[0], 3, out=d0[:Bh * H * W], ld_out=int(d0.stride(0)))
            if dup:                              # second CFG half of the skip: a row-strided column range of its consumer's concat buffer
                n0 = Bh * H * W
                L.copy_cols(d0[:n0], d0[n0:], 0, n0, boc[0], ld_src=int(d0.stride(0)), ld_dst=int(d0.stride(0)))
            t_full = d0
        else:
            t, _, _ = L.conv2d(x, P["conv_in.kernel"], P["conv_in.bias"], Bh, H, W, Cin, boc[0], 3)
            t_full = twice(t)
        if tape is not None:
            tape.append(("head", dict(emb=emb, t1=t1, s1=s1, temb=temb, temb_act=temb_act, x=Act(x, B, H, W, Cin))))
        h_half = Act(t, Bh, H, W, boc[0])
        h = Act(t_full, B, H, W, boc[0])
        skips = [h]
        for i in range(nlev):
            for j in range(cfg.layers_per_block):
                if dup and i == 0 and j == 0:      # still in front of the first cross-attention: half the batch, then duplicate
                    r = resnet_forward(P, "down_blocks_0.resnets_0", h_half, temb_act if shared_t else temb_act[:Bh], G, 1e-5, None)
                    h = Act(twice(r.t), B, r.H, r.W, r.C)
                else:
                    # the level's last block feeds the down-sampler convolution: its output stage also emits planes (sampling)
                    emit = 0
                    if tape is None and i < nlev - 1 and j == cfg.layers_per_block - 1:      # the consumer's planes_pay value (0 / 1 / 2)
                        emit = L.planes_pay(P[f"down_blocks_{i}.downsamplers_0.conv.kernel"], boc[i], h.M)
                    h = resnet_forward(P, f"down_blocks_{i}.resnets_{j}", h, temb_act, G, 1e-5, tape,
                                       emit_planes=0 if cfg.cross_attn_down[i] else emit,
                                       dest=None if cfg.cross_attn_down[i] else skip_dest(h.M))
                if cfg.cross_attn_down[i]:
                    emit_t = 0
                    if tape is None and i < nlev - 1 and j == cfg.layers_per_block - 1:
                        emit_t = L.planes_pay(P[f"down_blocks_{i}.downsamplers_0.conv.kernel"], boc[i], h.M)
                    h = self._transformer(f"down_blocks_{i}.attentions_{j}", h, ctx, Lc, cfg.num_heads[i], tape, emit_planes=emit_t,
                                          dest=skip_dest(h.M))
                skips.append(h)
                if tape is not None:
                    tape.append(("skip_push", None))
            if i < nlev - 1:
                name = f"down_blocks_{i}.downsamplers_0.conv"
                src = h.pl if (h.pl is not None and L.planes_pay(P[name + ".kernel"], h.C, h.M) == h.pl.fmt + 1) else h.t
                skw = {} if (src is not h.t or h.ld is None) else dict(ld_src=h.ld)
                dd = skip_dest(B * (h.H // 2) * (h.W // 2))
                if dd is not None:
                    skw.update(out=dd, ld_out=int(dd.stride(0)))
                t, OH, OW = L.conv2d(src, P[name + ".kernel"], P[name + ".bias"], B, h.H, h.W, h.C, h.C, 3, stride=2, pad=1, **skw)
                if tape is not None:
                    tape.append(("down", dict(name=name, x=h)))
                    tape.append(("skip_push", None))
                h = Act(t, B, OH, OW, h.C)
                skips.append(h)
        h = resnet_forward(P, "mid_block.resnets_0", h, temb_act, G, 1e-5, tape)
        h = self._transformer("mid_block.attentions_0", h, ctx, Lc, cfg.num_heads[-1], tape)
        h = resnet_forward(P, "mid_block.resnets_1", h, temb_act, G, 1e-5, tape, dest=up_dest(0))
        kblk = 0                                                       # up blocks consumed so far
        for i in range(nlev):
            lvl = nlev - 1 - i
            for j in range(cfg.layers_per_block + 1):
                s = skips.pop()
                if inplace:                                            # both halves are already in place (written by their producers)
                    cat = cat_bufs.pop()
                    assert cat.shape == (B * h.HW, h.C + s.C) and h.t.data_ptr() == cat.data_ptr() and s.t.data_ptr() == cat.data_ptr() + 4 * h.C
                else:
                    cat = torch.empty(B * h.HW, h.C + s.C, dtype=torch.float32, device=self.device)
                    L.copy_cols(h.t, cat, 0, B * h.HW, h.C)
                    L.copy_cols(s.t, cat, h.C, B * h.HW, s.C)
                if tape is not None:
                    tape.append(("concat", dict(c0=h.C, c1=s.C)))
                kblk += 1
                emit = 0
                last_of_level = j == cfg.layers_per_block
                if tape is None and i < nlev - 1 and last_of_level:      # feeds the up-sampler convolution: its planes_pay value
                    emit = L.planes_pay(P[f"up_blocks_{i}.upsamplers_0.conv.kernel"], boc[lvl], B * h.HW)
                # where this block's output goes: the next up block's concat buffer — unless an up-sampler convolution (or conv_norm_out) reads it
                nxt = None if last_of_level else up_dest(kblk)
                h = resnet_forward(P, f"up_blocks_{i}.resnets_{j}", Act(cat, B, h.H, h.W, h.C + s.C), temb_act, G, 1e-5, tape,
                                   emit_planes=0 if cfg.cross_attn_down[lvl] else emit, dest=None if cfg.cross_attn_down[lvl] else nxt)
                if cfg.cross_attn_down[lvl]:
                    h = self._transformer(f"up_blocks_{i}.attentions_{j}", h, ctx, Lc, cfg.num_heads[lvl], tape, emit_planes=emit, dest=nxt)
            if i < nlev - 1:
                name = f"up_blocks_{i}.upsamplers_0.conv"
                src = h.pl if (h.pl is not None and L.planes_pay(P[name + ".kernel"], h.C, h.M) == h.pl.fmt + 1) else h.t
                dd = up_dest(kblk)
                t, OH, OW = L.conv2d(src, P[name + ".kernel"], P[name + ".bias"], B, h.H, h.W, h.C, h.C, 3, upsample=True,
                                     **({} if dd is None else dict(out=dd, ld_out=int(dd.stride(0)))))
                if tape is not None:
                    tape.append(("up", dict(name=name, x=h)))
                h = Act(t, B, OH, OW, h.C)
        hn, st = L.groupnorm(h.t, B, h.HW, P["conv_norm_out.scale"], P["conv_norm_out.bias"], G, 1e-5, True, return_stats=True,
                             planes=L.norm_planes(P["conv_out.kernel"], h.C, h.M, tape is not None) if tape is None else 0)
        t, _, _ = L.conv2d(hn, P["conv_out.kernel"], P["conv_out.bias"], B, h.H, h.W, h.C, cfg.out_channels, 3)
        if tape is not None:
            tape.append(("tail", dict(x=h, hn=hn, st=st)))
        return L.nhwc_to_nchw(t, B, cfg.out_channels, h.H, h.W)

    __call__ = forward

    def _concat_plan(self):
        """(c0, c1) = (channels of the up path's running activation, channels of the popped skip) of every up block, in consumption order."""
        cfg = self.cfg
        boc = cfg.block_out_channels
        nlev = len(boc)
        pushed = [boc[0]]
        for i in range(nlev):
            pushed += [boc[i]] * cfg.layers_per_block
            if i < nlev - 1:
                pushed.append(boc[i])
        plan, ch, rev = [], boc[-1], boc[::-1]
        for i in range(nlev):
            for _ in range(cfg.layers_per_block + 1):
                plan.append((ch, pushed.pop()))
                ch = rev[i]
        assert not pushed
        return plan

    # -------------------------------------------------------------------------------- text-context K/V cache (sampling)
    def cross_attention_names(self):
        return [n[:-len(".to_k.kernel")] for n in self.params.views if n.endswith(".attn2.to_k.kernel")]

    def precompute_context(self, context):
        """Project the (constant) text context through every cross-attention to_k / to_v ONCE for a sampling call: the 50
        DDIM steps then skip 32 small GEMMs each.  Results land in persistent buffers (same addresses across calls, so a
        captured HIP graph keeps reading them); bit-identical to projecting per step.  Valid until `release_context()` or
        the next parameter update — the sampler brackets its step loop with these two calls.
        Buffers are keyed by (layer, context rows) and never freed or replaced: a HIP graph captured for one batch geometry
        keeps valid addresses when the same U-Net later samples another geometry and comes back (batch 8 -> 4 -> 8)."""
        B, Lc, D = context.shape
        ctx = context.reshape(B * Lc, D).contiguous()
        for name in self.cross_attention_names():
            wk, wv = self.params[name + ".to_k.kernel"], self.params[name + ".to_v.kernel"]
            ent = self._ctx_kv.get((name, B * Lc))
            heads = self._heads_of(name)
            nb = 0
            if L.current_datapath() != "fp32" and (wk.shape[1] // heads) in (8, 16, 40, 64, 80) and os.environ.get("DDPO_CTX_IMAGES", "1") != "0":
                nb = int(L.load().ddpo_attention_kv_images_bytes(B, heads, Lc, wk.shape[1] // heads))
            if ent is None:
                kbuf = torch.empty(B * Lc, wk.shape[1], dtype=torch.float32, device=self.device)
                vbuf = torch.empty(B * Lc, wv.shape[1], dtype=torch.float32, device=self.device)
                ent = (kbuf, vbuf, torch.empty(nb, dtype=torch.uint8, device=self.device) if nb else None)
                self._ctx_kv[(name, B * Lc)] = ent
            elif ent[2] is None and nb:            # first projected on the fp32 datapath (no image kernels): the image buffer is added now
                ent = (ent[0], ent[1], torch.empty(nb, dtype=torch.uint8, device=self.device))
                self._ctx_kv[(name, B * Lc)] = ent
            L.linear(ctx, wk, out=ent[0])
            L.linear(ctx, wv, out=ent[1])
            if ent[2] is not None:                 # the cross-attention of all T steps streams these images: no per-step split / transpose of K, V
                L.attention_kv_images(ent[0], ent[1], B, heads, Lc, wk.shape[1] // heads, out=ent[2])
        self._ctx_kv_active = True

    def release_context(self):
        self._ctx_kv_active = False
        self._graph_ctx_token = None              # (forward_graphed_cfg: the next sampling call copies its own context)

    def _heads_of(self, attn_name):
        """Number of heads of the attention layer `attn_name` (…down_blocks_i / up_blocks_i / mid_block…): cfg.num_heads per level."""
        cfg = self.cfg
        nlev = len(cfg.block_out_channels)
        if attn_name.startswith("down_blocks_"):
            return cfg.num_heads[int(attn_name.split("_")[2].split(".")[0])]
        if attn_name.startswith("up_blocks_"):
            return cfg.num_heads[nlev - 1 - int(attn_name.split("_")[2].split(".")[0])]
        return cfg.num_heads[-1]

    # -------------------------------------------------------------------------------- time-projection table (sampling)
    def time_proj_names(self):
        return [n[:-len(".time_emb_proj.kernel")] for n in self.params.views if n.endswith(".time_emb_proj.kernel")]

    def precompute_timesteps(self, timesteps):
        """Run the time path — sinusoidal embedding, the 2-layer MLP, SiLU and every ResBlock's time_emb_proj — ONCE for all T
        timesteps of a sampling call (rows = steps) instead of once per step on B identical rows: 27 launch-bound GEMM /
        elementwise launches per U-Net call disappear from the step loop.  Each row is bit-identical to what forward() computes
        for that timestep (a GEMM row does not depend on the other rows; the tile / split-K choice depends on ceil(M / 128),
        which is 1 either way — asserted).  `select_timestep(i)` copies row i into a static (1, sum cout) buffer whose column
        slices are the `rowbias` operands of the ResBlocks' conv1 (stable addresses: captured HIP graphs keep reading them).
        Valid until `release_timesteps()` or the next parameter update, like the text-context cache."""
        P = self.params
        ts = torch.as_tensor(timesteps, dtype=torch.int32, device=self.device).reshape(-1).contiguous()
        T = ts.numel()
        names = self.time_proj_names()
        widths = [P[n + ".time_emb_proj.bias"].numel() for n in names]
        total = sum(widths)
        ent = self._temb
        if ent is None or ent["table"].shape != (T, total):
            row = torch.zeros(1, total, dtype=torch.float32, device=self.device) if ent is None else ent["row"]
            views, off = {}, 0
            for n, w in zip(names, widths):
                views[n] = row[:, off:off + w]
                off += w
            ent = dict(table=torch.empty(T, total, dtype=torch.float32, device=self.device), row=row, views=views, names=names, widths=widths)
            self._temb = ent
        # at most 128 rows per GEMM: one row tile, i.e. the tiling (and so every row's bits) of the per-step time path, for any step count
        for r0 in range(0, T, 128):
            r1 = min(T, r0 + 128)
            emb = L.timestep_embedding(ts[r0:r1].contiguous(), self.cfg.block_out_channels[0])
            t1 = L.linear(emb, P["time_embedding.linear_1.kernel"], P["time_embedding.linear_1.bias"])
            temb = L.linear(L.silu(t1), P["time_embedding.linear_2.kernel"], P["time_embedding.linear_2.bias"])
            act = L.silu(temb)
            off = 0
            for n, w in zip(names, widths):
                L.linear(act, P[n + ".time_emb_proj.kernel"], P[n + ".time_emb_proj.bias"], out=ent["table"][r0:r1, off:off + w], ld_out=total)
                off += w
        self._temb_active = True

    def select_timestep(self, i):
        """Make step i of the table current (one small device copy, stream-ordered with the U-Net launches / graph replay)."""
        self._temb["row"].copy_(self._temb["table"][i:i + 1])

    def release_timesteps(self):
        self._temb_active = False

    def forward_graphed(self, sample, timesteps, context, cfg_dup=False):
        """Same as forward(), but the ~1000 kernel launches of one U-Net pass are captured once into a HIP graph (per
        input geometry) and replayed: the launch-bound host loop disappears from the sampling hot loop.  Inputs are copied
        into the graph's static buffers; the returned tensor is the graph's static output (valid until the next replay).
        Weights are read in place, so optimizer updates / re-packing are seen by later replays."""
        key = (tuple(sample.shape), tuple(context.shape), L.current_datapath(), self._ctx_kv_active, bool(cfg_dup), self._temb_active)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        ent = self._graphs.get(key)
        if ent == "eager":
            return self.forward(sample, timesteps, context, cfg_dup=cfg_dup)
        if ent is None:
            s_in = sample.clone().contiguous()
            t_in = timesteps.to(torch.int32).clone().contiguous()
            c_in = context.clone().contiguous()
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):                      # warm-up: first-call attribute setup, scratch allocation
                    self.forward(s_in, t_in, c_in, cfg_dup=cfg_dup)
            torch.cuda.current_stream(self.device).wait_stream(side)
            try:
                graph = torch.cuda.CUDAGraph()
                # thread_local: other host threads (RCCL watchdog, reward callbacks) may touch the device during capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = self.forward(s_in, t_in, c_in, cfg_dup=cfg_dup)
            except Exception as exc:          # capture is an optimisation only: fall back to eager launches, loudly
                print(f"[ ddpo_amd ] WARNING: HIP-graph capture of the U-Net failed ({type(exc).__name__}: {exc}); launching eagerly")
                torch.cuda.synchronize(self.device)
                self._graphs[key] = "eager"
                return self.forward(sample, timesteps, context, cfg_dup=cfg_dup)
            ent = (graph, s_in, t_in, c_in, out)
            self._graphs[key] = ent
        graph, s_in, t_in, c_in, out = ent
        s_in.copy_(sample)
        t_in.copy_(timesteps)
        c_in.copy_(context)
        graph.replay()
        return out

    def forward_graphed_cfg(self, x, step, timesteps, context, cfg_dup=True):
        """One classifier-free-guidance sampling step from the captured graph: forward_graphed([x; x], ...) with the graph's inputs staged by ONE
        launch of the engine (ddpo_stage_cfg_inputs: both halves of the latents, the step's row of the time-projection table when `step` is not
        None, the timesteps) instead of six stock copy launches per step.  The text context is copied into the graph's
        buffer once per sampling call (precompute_context resets the token).  The first call per geometry captures through forward_graphed."""
        B = x.shape[0]
        key = ((2 * B,) + tuple(x.shape[1:]), tuple(context.shape), L.current_datapath(), self._ctx_kv_active, bool(cfg_dup), self._temb_active)
        ent = getattr(self, "_graphs", {}).get(key)
        if ent is None or ent == "eager":
            if step is not None:
                self.select_timestep(step)
            self._graph_ctx_token = None
            return self.forward_graphed(torch.cat([x, x]), timesteps, context, cfg_dup=cfg_dup)
        graph, s_in, t_in, c_in, out = ent
        tm = self._temb if step is not None else None
        L.stage_cfg_inputs(x if x.is_contiguous() else x.contiguous(), s_in,
                           None if tm is None else tm["table"][step], None if tm is None else tm["row"],
                           timesteps.to(torch.int32).contiguous(), t_in)
        token = (id(context), context._version, key)
        if getattr(self, "_graph_ctx_token", None) != token:
            c_in.copy_(context)
            self._graph_ctx_token = token
        graph.replay()
        return out

    # -------------------------------------------------------------------------------- backward
    class _Progress:
        """Which suffix of the flat gradient buffer is final: blocks report themselves by name prefix, the answer is the offset of the
        first parameter of the longest fully-reported run of parameters at the END of the buffer (independent of how resnets and
        attentions of a block interleave in the layout)."""

        def __init__(self, offsets):
            self.names = list(offsets)
            self.offs = [offsets[n] for n in self.names]
            self.fin = [False] * len(self.names)
            self.ptr = len(self.names)                   # parameters [ptr:] are final

        def report(self, prefix):
            for i, n in enumerate(self.names):
                if n.startswith(prefix):
                    self.fin[i] = True
            while self.ptr > 0 and self.fin[self.ptr - 1]:
                self.ptr -= 1
            return self.offs[self.ptr] if self.ptr < len(self.offs) else None

    def backward(self, tape, d_out, on_ready=None):
        """Accumulates d loss / d params into self.grads given d loss / d output (B,C_out,H,W) and the tape of `forward`.
        (The latents and the text context are not differentiated: DDPO only needs parameter gradients.)
        on_ready(lo): called after the kernels of each block have been queued — every gradient at flat offset >= lo is then final
        on this stream (the backward runs through the parameters from the end of the buffer to its start); the data-parallel
        trainer hangs the bucketed all-reduce on it (training/distributed.GradBucketer)."""
        P, cfg = self.params, self.cfg
        G = self.ensure_grads()
        prog = UNet2DCondition._Progress(self.params.offsets) if on_ready is not None else None

        def done(prefix):
            if prog is not None:
                lo = prog.report(prefix + ".")
                if lo is not None:
                    on_ready(lo)
        kind, head = tape[0]
        assert kind == "head"
        tctx = dict(temb_act=head["temb_act"], d_temb_act=None)
        kind, tail = tape[-1]
        assert kind == "tail"
        x = tail["x"]
        B, H, W = x.B, x.H, x.W
        d = L.nchw_to_nhwc(d_out.contiguous())                                 # (B*H*W, C_out)
        L.conv2d_wgrad(tail["hn"], d, G["conv_out.kernel"], B, H, W, x.C, cfg.out_channels, 3, dbias=G["conv_out.bias"])
        d = L.conv2d_dgrad(d, P["conv_out.kernel"], B, H, W, x.C, cfg.out_channels, 3)
        d = L.groupnorm_bwd(x.t, d, tail["st"], P["conv_norm_out.scale"], B, x.HW, cfg.norm_groups, True,
                            G["conv_norm_out.scale"], G["conv_norm_out.bias"])
        done("conv_out")
        done("conv_norm_out")
        skip_grads = []          # filled by the (reversed) up path: ends up in forward production order, consumed from the end
        for kind, r in reversed(tape[1:-1]):
            if kind == "resnet":
                d = resnet_backward(P, G, r, d, tctx)
                done(r["name"])
            elif kind == "transformer":
                d = self._transformer_backward(r, d)
                done(r["name"])
            elif kind == "concat":
                rows = d.shape[0]
                d_h = torch.empty(rows, r["c0"], dtype=torch.float32, device=self.device)
                d_s = torch.empty(rows, r["c1"], dtype=torch.float32, device=self.device)
                L.copy_cols(d, d_h, 0, rows, r["c0"], ld_src=r["c0"] + r["c1"])
                L.copy_cols(d[:, r["c0"]:], d_s, 0, rows, r["c1"], ld_src=r["c0"] + r["c1"])
                skip_grads.append(d_s)
                d = d_h
            elif kind == "up":
                xx, name = r["x"], r["name"]
                L.conv2d_wgrad(xx.t, d, G[name + ".kernel"], xx.B, xx.H, xx.W, xx.C, xx.C, 3, upsample=True, dbias=G[name + ".bias"])
                d_up = L.conv2d_dgrad(d, P[name + ".kernel"], xx.B, 2 * xx.H, 2 * xx.W, xx.C, xx.C, 3)
                d = L.sumpool2x2(d_up, xx.B, xx.H, xx.W, xx.C)
                done(name)
            elif kind == "down":
                xx, name = r["x"], r["name"]
                L.conv2d_wgrad(xx.t, d, G[name + ".kernel"], xx.B, xx.H, xx.W, xx.C, xx.C, 3, stride=2, pad=1, dbias=G[name + ".bias"])
                d = L.conv2d_dgrad(d, P[name + ".kernel"], xx.B, xx.H, xx.W, xx.C, xx.C, 3, stride=2)
                done(name)
            elif kind == "skip_push":
                # the activation produced just before this marker was also a skip connection: add its up-path grad.
                d = L.add(d, skip_grads.pop())
            else:
                raise RuntimeError(kind)
        # conv_in output was the first skip
        d = L.add(d, skip_grads.pop())
        assert not skip_grads
        xin = head["x"]
        L.conv2d_wgrad(xin.t, d, G["conv_in.kernel"], xin.B, xin.H, xin.W, xin.C, cfg.block_out_channels[0], 3, dbias=G["conv_in.bias"])
        # time embedding MLP
        d_temb = L.silu_bwd(head["temb"], tctx["d_temb_act"])
        L.linear_wgrad(head["s1"], d_temb, G["time_embedding.linear_2.kernel"], dbias=G["time_embedding.linear_2.bias"])
        d_s1 = L.linear_dgrad(d_temb, P["time_embedding.linear_2.kernel"])
        d_t1 = L.silu_bwd(head["t1"], d_s1)
        L.linear_wgrad(head["emb"], d_t1, G["time_embedding.linear_1.kernel"], dbias=G["time_embedding.linear_1.bias"])
