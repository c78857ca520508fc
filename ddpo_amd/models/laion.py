"""LAION aesthetic predictor on the engine's kernels: CLIP ViT-L/14 image embedding (models/clip_vision.py) -> L2 normalisation
(`ddpo_l2_normalize_rows`) -> 5-layer linear MLP 768-1024-128-64-16-1 (five `ddpo_gemm_conv_fwd` launches, exact-fp32 MFMA).

Mirror of /root/reference/ddpo/models/laion.py:7-51 (the dropouts are inert at inference; `set_weights` :38-51 transposes the
published `.pth`'s torch (out,in) matrices into Dense kernels — the engine's (in,out) layout) and of the scoring path of
/root/reference/ddpo/training/callbacks.py:60-95.  Runs on a private HIP stream: the reward callback is evaluated by a worker thread
while the main thread samples the next batch (pipeline/policy_gradient.py), and the two must not share a stream or scratch space.

Weights (nothing can be downloaded here):
  CLIP   `<weights_dir>/clip/` or an HF cache snapshot of `openai/clip-vit-large-patch14` (torch or Flax files)
  MLP    `<weights_dir>/sac+logos+ava1-l14-linearMSE.pth`, else `<repo>/<cache>/sac+logos+ava1-l14-linearMSE.pth` (where the reference
         keeps it, laion.py:22-24)
with `weights_dir` = the argument, else $DDPO_AESTHETIC_WEIGHTS.  Missing weights raise, unless DDPO_ALLOW_SYNTHETIC=1 asks for a
seeded random-init model (benchmarks / tests; `synthetic` is then True and the callback's info says so)."""
import os

import numpy as np
import torch

from .. import lib as L
from .clip_vision import ClipVisionTower, VisionConfig, flax_tree_to_torch_names, preprocess

MLP_FILE = "sac+logos+ava1-l14-linearMSE.pth"
MLP_LAYERS = (0, 2, 4, 6, 7)                  # keys of the published state dict (laion.set_weights :41)
MLP_DIMS = (1024, 128, 64, 16, 1)
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_state_dicts(cfg, in_dim, seed=0):
    """Seeded random-init CLIP-vision + MLP state dicts in the checkpoint naming (transformers torch names / the .pth's names)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    C, v, sd = cfg.hidden, "vision_model.", {}
    sd[v + "embeddings.class_embedding"] = 0.02 * rn(C)
    sd[v + "embeddings.patch_embedding.weight"] = rn(C, 3, cfg.patch, cfg.patch) / (3 * cfg.patch * cfg.patch) ** 0.5
    sd[v + "embeddings.position_embedding.weight"] = 0.02 * rn(cfg.tokens, C)
    for n in ("pre_layrnorm", "post_layernorm"):
        sd[v + n + ".weight"], sd[v + n + ".bias"] = 1 + 0.1 * rn(C), 0.02 * rn(C)
    for i in range(cfg.layers):
        p = f"{v}encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = 1 + 0.1 * rn(C), 0.02 * rn(C)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{n}.weight"], sd[p + f"self_attn.{n}.bias"] = rn(C, C) / C ** 0.5, 0.02 * rn(C)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = rn(cfg.mlp, C) / C ** 0.5, 0.02 * rn(cfg.mlp)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = rn(C, cfg.mlp) / cfg.mlp ** 0.5, 0.02 * rn(C)
    sd["visual_projection.weight"] = rn(cfg.proj, C) / C ** 0.5
    mlp, a = {}, in_dim
    for idx, b in zip(MLP_LAYERS, MLP_DIMS):
        mlp[f"layers.{idx}.weight"], mlp[f"layers.{idx}.bias"] = rn(b, a) / a ** 0.5, 0.02 * rn(b)
        a = b
    return sd, mlp


def _load_clip_state(clip_dir):
    for f in ("model.safetensors", "pytorch_model.bin"):
        p = os.path.join(clip_dir, f)
        if os.path.exists(p):
            if f.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p)
            return torch.load(p, map_location="cpu", weights_only=True)
    p = os.path.join(clip_dir, "flax_model.msgpack")
    if os.path.exists(p):
        from ..utils.flax_msgpack import flatten, from_bytes
        with open(p, "rb") as fh:
            return flax_tree_to_torch_names(flatten(from_bytes(fh.read())))
    return None


def find_weights(weights_dir=None, cache="cache"):
    """(clip_dir or None, mlp_path or None) following the lookup order in the module docstring."""
    from ..utils.serialization import resolve_pretrained
    weights_dir = weights_dir or os.environ.get("DDPO_AESTHETIC_WEIGHTS")
    clip_dir = None
    if weights_dir and os.path.isdir(os.path.join(weights_dir, "clip")):
        clip_dir = os.path.join(weights_dir, "clip")
    if clip_dir is None:
        clip_dir = resolve_pretrained("openai/clip-vit-large-patch14", os.path.join(REPO_ROOT, cache) if not os.path.isabs(cache) else cache)
    mlp = None
    for cand in ([os.path.join(weights_dir, MLP_FILE)] if weights_dir else []) + [os.path.join(REPO_ROOT, cache, MLP_FILE), os.path.join(cache, MLP_FILE)]:
        if os.path.exists(cand):
            mlp = cand
            break
    return clip_dir, mlp


class AestheticScorer:
    def __init__(self, weights_dir=None, cache="cache", seed=0, device="cuda", config="vit-l/14", clip_state=None, mlp_state=None):
        """`clip_state` / `mlp_state`: state dicts handed in directly (tests); otherwise files are looked up (see module docstring)."""
        self.device = torch.device(device)
        self.cfg = VisionConfig.named(config)
        self.synthetic = False
        if clip_state is None or mlp_state is None:
            clip_dir, mlp_path = find_weights(weights_dir, cache)
            loaded = _load_clip_state(clip_dir) if clip_dir else None
            if loaded is None or mlp_path is None:
                from ..utils.serialization import allow_synthetic
                if not allow_synthetic():
                    missing = [n for n, ok in (("the CLIP ViT-L/14 checkpoint (openai/clip-vit-large-patch14)", loaded is not None),
                                               (MLP_FILE, mlp_path is not None)) if not ok]
                    raise FileNotFoundError(
                        f"aesthetic reward: {' and '.join(missing)} not found (looked in weights_dir / $DDPO_AESTHETIC_WEIGHTS, the HF cache and "
                        f"'{os.path.join(REPO_ROOT, cache)}'; nothing is downloaded).  Set DDPO_ALLOW_SYNTHETIC=1 to score with a seeded "
                        f"RANDOM-INIT model (benchmarks / tests only)")
                print("[ models/laion ] WARNING: DDPO_ALLOW_SYNTHETIC=1 and no aesthetic-predictor weights on disk — scoring with a seeded "
                      "random-init CLIP ViT-L/14 + MLP; rewards are meaningless")
                clip_state, mlp_state = synthetic_state_dicts(self.cfg, self.cfg.proj, seed)
                self.synthetic = True
            else:
                clip_state = loaded
                mlp_state = torch.load(mlp_path, map_location="cpu", weights_only=True)
        self.stream = torch.cuda.Stream(self.device)          # (after the weight lookup: a missing-weights refusal needs no GPU)
        with torch.cuda.stream(self.stream):
            self.tower = ClipVisionTower(self.cfg, self.device)
            self.tower.load_state_dict(clip_state)
            # MLP in the engine's (in, out) layout; the 1-wide last layer is zero-padded to 4 columns (vector epilogue of the GEMM)
            self.mlp = []
            for idx in MLP_LAYERS:
                w = torch.as_tensor(mlp_state[f"layers.{idx}.weight"]).float().t().contiguous()
                b = torch.as_tensor(mlp_state[f"layers.{idx}.bias"]).float()
                n4 = (w.shape[1] + 3) // 4 * 4
                wp = torch.zeros(w.shape[0], n4); wp[:, :w.shape[1]] = w
                bp = torch.zeros(n4); bp[:b.numel()] = b
                self.mlp.append((wp.to(self.device), bp.to(self.device), w.shape[1]))
        self.stream.synchronize()

    def features(self, pixel_values):
        return self.tower(pixel_values)

    def __call__(self, images):
        """images: float32 (N,H,W,3) in [0,1] (host) -> (N,) float32 scores (host)."""
        px = preprocess(images, self.cfg.image)                               # host, PIL: byte-identical resize
        with torch.cuda.stream(self.stream), L.fp32_class_datapath():
            x = torch.from_numpy(px).to(self.device)
            f = L.l2_normalize_rows(self.tower(x))
            for wp, bp, n in self.mlp:
                f = L.linear(f[:, :wp.shape[0]].contiguous() if f.shape[1] != wp.shape[0] else f, wp, bp)
            scores = f[:, 0].contiguous().cpu().numpy()
        self.stream.synchronize()
        return scores
