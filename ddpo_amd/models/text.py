"""Tokenizer + CLIP text encoder used to embed prompts (reference: `pipeline.prepare_inputs`
/root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:148-161 and `text_encode`
/root/reference/pipeline/policy_gradient.py:185-199; `make_uncond_text` /root/reference/ddpo/datasets/bucket.py:66-73).

The text tower is NOT a hand-kernel target (~6.5 GFLOP per prompt): it runs as stock PyTorch-ROCm
modules.  With local HF files (`<dir>/tokenizer`, `<dir>/text_encoder`) the real CLIP BPE vocabulary and weights are
used; offline (no vocab / checkpoints on disk) a reversible byte-level stand-in tokenizer and a seeded random-init
CLIPTextModel of the right architecture are used, and `synthetic` is set so callers can say so.
"""
import os

import numpy as np
import torch


class _Encoding:
    def __init__(self, input_ids):
        self.input_ids = input_ids


class ByteTokenizer:
    """Offline stand-in with CLIP's framing: <bos>=49406, ids = 1000 + utf-8 byte, <eos>=49407 used for padding."""
    model_max_length = 77
    bos_token_id = 49406
    eos_token_id = 49407
    synthetic = True

    def __call__(self, text, padding="max_length", max_length=None, truncation=True, return_tensors="np"):
        if isinstance(text, str):
            text = [text]
        L = max_length or self.model_max_length
        out = np.full((len(text), L), self.eos_token_id, dtype=np.int64)
        for i, t in enumerate(text):
            body = [1000 + b for b in t.lower().encode("utf-8")][: L - 2]
            out[i, 0] = self.bos_token_id
            out[i, 1:1 + len(body)] = body
        return _Encoding(out)

    def batch_decode(self, ids, skip_special_tokens=True):
        res = []
        for row in np.asarray(ids):
            res.append(bytes(int(t) - 1000 for t in row if 1000 <= t < 1256).decode("utf-8", errors="ignore"))
        return res


def make_uncond_text(tokenizer, batch_size):
    """Token ids of the empty prompt, padded to max length."""
    return tokenizer([""] * batch_size, padding="max_length", max_length=tokenizer.model_max_length, return_tensors="np").input_ids


class TextEncoder:
    def __init__(self, pretrained_dir=None, hidden=768, device="cuda", seed=0):
        from transformers import CLIPTextConfig, CLIPTextModel
        self.device = torch.device(device)
        sub = os.path.join(pretrained_dir, "text_encoder") if pretrained_dir else None
        if sub and os.path.isdir(sub):
            has_torch = any(os.path.exists(os.path.join(sub, f)) for f in ("model.safetensors", "pytorch_model.bin"))
            fx = os.path.join(sub, "flax_model.msgpack")
            if has_torch:
                self.model = CLIPTextModel.from_pretrained(sub)
            elif os.path.exists(fx):
                # HF Flax repositories (the reference's default `duongna/stable-diffusion-v1-4-flax`) ship only the Flax tree:
                # read it with the in-tree msgpack reader and transpose it into the torch module (no flax / jax needed)
                from ..utils.flax_msgpack import flatten, from_bytes
                from ..utils.serialization import flax_clip_text_to_torch
                self.model = CLIPTextModel(CLIPTextConfig.from_pretrained(sub))
                with open(fx, "rb") as f:
                    sd = flax_clip_text_to_torch(flatten(from_bytes(f.read())))
                want = {k for k in self.model.state_dict() if not k.endswith("position_ids")}
                missing, extra = sorted(want - set(sd)), sorted(set(sd) - want)
                if missing:
                    raise KeyError(f"{fx}: parameters missing for CLIPTextModel: {missing[:4]}{'...' if len(missing) > 4 else ''}")
                self.model.load_state_dict({k: v for k, v in sd.items() if k not in extra}, strict=False)
            else:
                raise FileNotFoundError(f"{sub} holds neither PyTorch (model.safetensors / pytorch_model.bin) nor Flax (flax_model.msgpack) weights")
            self.synthetic = False
        else:
            if hidden == 768:      # CLIP ViT-L/14 text tower (SD-1.x): 123,060,480 parameters
                cfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                     num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=768)
            elif hidden < 512:     # toy tower for the `tiny` test configuration
                cfg = CLIPTextConfig(vocab_size=49408, hidden_size=hidden, intermediate_size=4 * hidden, num_hidden_layers=2,
                                     num_attention_heads=4, max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=hidden)
            else:                  # OpenCLIP ViT-H text tower (SD-2.x), penultimate layer dropped as in the SD-2 checkpoints
                cfg = CLIPTextConfig(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                                     num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu", projection_dim=512)
            torch.manual_seed(seed)
            self.model = CLIPTextModel(cfg)
            self.synthetic = True
        self.model.to(self.device).eval()

    @torch.no_grad()
    def __call__(self, input_ids):
        ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long, device=self.device)
        return self.model(ids)[0].float()          # last hidden state (N, 77, D)


def load_tokenizer(pretrained_dir=None):
    sub = os.path.join(pretrained_dir, "tokenizer") if pretrained_dir else None
    if sub and os.path.isdir(sub):
        from transformers import CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(sub)
        tok.synthetic = False
        return tok
    return ByteTokenizer()
