"""LAION aesthetic predictor: CLIP ViT-L/14 image embedding -> 5-layer linear MLP (768-1024-128-64-16-1).

Mirror of /root/reference/ddpo/models/laion.py:7-51 (the dropouts are inert at inference) and of the scoring path of
/root/reference/ddpo/training/callbacks.py:60-95.  Stock PyTorch-ROCm modules on a private stream (this is a
reward model, not part of the hand-written hot path — SURVEY.md §8f-1).
"""
import os

import numpy as np
import torch


class AestheticClassifier(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in ((768, 1024), (1024, 128), (128, 64), (64, 16), (16, 1))])

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class AestheticScorer:
    def __init__(self, weights_dir=None, seed=0, device="cuda"):
        from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.synthetic = True
        clip_dir = os.path.join(weights_dir, "clip") if weights_dir else None
        if clip_dir and os.path.isdir(clip_dir):
            self.clip = CLIPVisionModelWithProjection.from_pretrained(clip_dir)
            self.synthetic = False
        else:
            torch.manual_seed(seed)
            cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                                   image_size=224, patch_size=14, projection_dim=768)
            self.clip = CLIPVisionModelWithProjection(cfg)
        self.head = AestheticClassifier()
        head_path = os.path.join(weights_dir, "sac+logos+ava1-l14-linearMSE.pth") if weights_dir else None
        if head_path and os.path.exists(head_path):
            sd = torch.load(head_path, map_location="cpu")
            keys = sorted({k.rsplit(".", 1)[0] for k in sd}, key=lambda s: int(s.split(".")[-1]))
            for layer, k in zip(self.head.layers, keys):
                layer.weight.data.copy_(sd[k + ".weight"])
                layer.bias.data.copy_(sd[k + ".bias"])
        else:
            self.synthetic = True
        self.clip.to(self.device).eval()
        self.head.to(self.device).eval()

    @torch.no_grad()
    def __call__(self, images):
        """images: float32 (N,H,W,3) in [0,1] -> (N,) float32 scores."""
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else torch.no_grad()
        with ctx:
            x = torch.from_numpy(images).to(self.device).permute(0, 3, 1, 2)
            # CLIPProcessor: bicubic resize of the short side to 224, centre crop 224, normalise
            x = torch.nn.functional.interpolate(x, size=(224, 224), mode="bicubic", align_corners=False, antialias=True)
            mean = torch.tensor(_CLIP_MEAN, device=self.device).view(1, 3, 1, 1)
            std = torch.tensor(_CLIP_STD, device=self.device).view(1, 3, 1, 1)
            feats = self.clip(pixel_values=(x - mean) / std).image_embeds
            feats = feats / feats.norm(dim=-1, keepdim=True)
            scores = self.head(feats).squeeze(-1).float().cpu().numpy()
        if self.stream is not None:
            self.stream.synchronize()
        return scores
