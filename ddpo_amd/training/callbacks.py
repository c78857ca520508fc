"""`filter_field` plugin surface: reward callbacks.

Behavioural mirror of the registry / calling convention of /root/reference/ddpo/training/callbacks.py:
  callback_fns[name](**factory_kwargs) -> fn(images, prompts, metadata) -> (scores, info)       (:549-564)
  evaluate_callbacks(fns, images, prompts, metadata) -> {name: (scores, info)}                  (:540-546)
images: float32 (N,H,W,3) in [0,1]; scores: (N,) or (N,1) numpy; info: dict of numpy arrays.
Callbacks run in a worker thread of the entrypoint (ThreadPoolExecutor, max_workers=2) next to the sampling of the
following batch, so they must not touch the sampler's HIP stream: the host ones below are pure CPU code and the
on-device one (aesthetic) uses its own stream.

In scope (BASELINE.json configs): jpeg, neg_jpeg, aesthetic, llava_bertscore (+ its sibling llava_vqa wire format).
The other reward ideas of the reference (rotational / mirror symmetry, thumbnail, BLIP-2 vqa, ...) are not part of
any benchmark config; add them as plugins with `register`.
"""
import io
import pickle
import random

import numpy as np
from PIL import Image


def register(name):
    def deco(factory):
        callback_fns[name] = factory
        return factory
    return deco


# ------------------------------------------------------------------------------------------------ jpeg compressibility
def encode_jpeg(x, quality=95):
    """float [0,1] or uint8 HxWx3 -> JPEG bytes as a uint8 array (reference ddpo/utils/hdf5.py:25-37: floats are
    TRUNCATED to uint8 via (x*255).astype(uint8), PIL quality 95)."""
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.floating):
        assert np.abs(x).max() <= 1.0
        x = (x * 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(x).save(buf, "JPEG", quality=quality)
    return np.frombuffer(buf.getvalue(), dtype=np.uint8)


def jpeg_fn(devices=None, jit=False):
    """reward = -(JPEG size in kB): compressibility (reference :143-153).  Returns (N,1) float64."""
    assert not jit

    def _fn(images, prompts, metadata):
        del prompts, metadata
        kb = [len(encode_jpeg(im)) / 1000.0 for im in images]
        return -np.array(kb)[:, None], {}

    return _fn


def neg_jpeg_fn(*a, **kw):
    """reward = +(JPEG size in kB): incompressibility (reference :156-163)."""
    inner = jpeg_fn(*a, **kw)

    def _fn(*args, **kwargs):
        scores, info = inner(*args, **kwargs)
        return -scores, info

    return _fn


# ------------------------------------------------------------------------------------------------ LAION aesthetic
def aesthetic_fn(devices=None, rng=0, cache="cache", jit=True, weights_dir=None):
    """CLIP ViT-L/14 image features -> L2-normalise -> LAION aesthetic MLP (reference :60-95, ddpo/models/laion.py), on the engine's
    own kernels (models/clip_vision.py, models/laion.py), on a private HIP stream.  Weights: `weights_dir` or $DDPO_AESTHETIC_WEIGHTS
    (`clip/` in HF format + `sac+logos+ava1-l14-linearMSE.pth`), else the HF cache and `<repo>/<cache>/` where the reference keeps the
    `.pth`; nothing is downloaded.  Missing weights RAISE — an `a_*` run must not silently optimise a random reward — unless
    DDPO_ALLOW_SYNTHETIC=1, in which case info['synthetic_weights'] is True."""
    del devices, jit
    from ..models.laion import AestheticScorer
    scorer = AestheticScorer(weights_dir=weights_dir, cache=cache, seed=rng)

    def _wrapper(images, prompts, metadata):
        del prompts, metadata
        scores = scorer(np.asarray(images, dtype=np.float32))
        return scores[:, None], {"synthetic_weights": np.array(scorer.synthetic)}

    return _wrapper


# ------------------------------------------------------------------------------------------------ LLaVA over HTTP
def _to_jpeg_bytes(image_u8, quality=80):
    buf = io.BytesIO()
    Image.fromarray(image_u8).save(buf, format="JPEG", quality=quality)
    return buf.getvalue()


def _llava_session():
    import requests
    from requests.adapters import HTTPAdapter, Retry
    sess = requests.Session()
    sess.mount("http://", HTTPAdapter(max_retries=Retry(total=1000, backoff_factor=1, status_forcelist=[500], allowed_methods=False)))
    return sess


def llava_bertscore(devices=None, jit=False, url="http://127.0.0.1:8085", batch_size=16, timeout=120):
    """Alignment reward served by a LLaVA + BERTScore server (reference :465-537).  Wire format: POST of
    pickle.dumps({"images": [jpeg bytes, q=80], "queries": [[str]], "answers": [[str]]}); the reply is a pickled dict
    with "recall" (the reward), "precision", "f1", "outputs".  Batches of 16; 1000 retries on HTTP 500."""
    sess = _llava_session()

    def _fn(images, prompts, metadata):
        del metadata
        images = (np.asarray(images) * 255).astype(np.uint8)
        nb = int(np.ceil(len(images) / batch_size))
        scores, info = [], {"precision": [], "f1": [], "outputs": []}
        for img_b, prm_b in zip(np.array_split(images, nb), np.array_split(np.asarray(prompts), nb)):
            payload = {"images": [_to_jpeg_bytes(im) for im in img_b],
                       "queries": [["Answer concisely: what is going on in this image?"]] * len(img_b),
                       "answers": [[f"The image contains {p}"] for p in prm_b]}
            reply = pickle.loads(sess.post(url, data=pickle.dumps(payload), timeout=timeout).content)
            scores += np.array(reply["recall"]).squeeze().reshape(-1).tolist()
            for k in info:
                info[k] += np.array(reply[k]).squeeze().reshape(-1).tolist()
        return np.array(scores), {k: np.array(v) for k, v in info.items()}

    return _fn


def llava_vqa_satisfaction(devices=None, jit=False, url="http://127.0.0.1:8085", batch_size=4, timeout=120):
    """VQA reward (reference :402-462): request {"images", "queries"} (questions from the prompt metadata), reply
    {"outputs"}; the score of an image is the fraction of answers that contain the expected answer string
    (case-sensitive); info = {"answers": the server's outputs}."""
    sess = _llava_session()

    def _fn(images, prompts, metadata):
        del prompts
        images = (np.asarray(images) * 255).astype(np.uint8)
        nb = int(np.ceil(len(images) / batch_size))
        metadata = list(metadata)
        scores, answers = [], []
        for img_b, meta_b in zip(np.array_split(images, nb), np.array_split(np.arange(len(images)), nb)):
            metas = [metadata[i] for i in meta_b]
            payload = {"images": [_to_jpeg_bytes(im) for im in img_b], "queries": [m["questions"] for m in metas]}
            reply = pickle.loads(sess.post(url, data=pickle.dumps(payload), timeout=timeout).content)
            for m, outs in zip(metas, reply["outputs"]):
                assert len(outs) == len(m["answers"])
                hits = [a in o for a, o in zip(m["answers"], outs)]          # case-sensitive substring test (:357-360)
                scores.append(float(np.mean(np.array(hits, dtype=int))))
            answers += reply["outputs"]
        return np.array(scores), {"answers": np.array(answers)}

    return _fn


# ------------------------------------------------------------------------------------------------ registry
def evaluate_callbacks(fns, images, prompts, metadata):
    if type(prompts[0]) == list:
        prompts = [random.choice(p) for p in prompts]
    images = np.asarray(images).astype(np.float32)
    return {key: fn(images, prompts, metadata) for key, fn in fns.items()}


def vae_fn(devices=None, dtype="float32", jit=True, encoder=None):
    """The `vae` field of the RWR sampler (reference :37-57): images (N,H,W,3) in [0,1] -> VAE posterior moments
    concat([mean, logvar], -1) (N,H/8,W/8,8) from the engine's VAE encoder (ddpo_amd/models/vae.py:VAEEncoder).  The reference
    loads the SD-1.4 Flax VAE itself; here the caller hands the loaded encoder over (`set_vae_encoder`, pipeline/sample.py) —
    nothing is downloaded."""
    import torch
    enc = encoder if encoder is not None else _VAE_ENCODER.get("encoder")
    if enc is None:
        raise RuntimeError("the `vae` callback needs a loaded VAE encoder: call ddpo_amd.training.callbacks.set_vae_encoder(encoder) first")

    def _fn(images, prompts=None, metadata=None):
        with torch.no_grad():
            m = enc.encode(torch.as_tensor(np.asarray(images, dtype=np.float32)))
        return m.cpu().numpy(), {}

    return _fn


_VAE_ENCODER = {}


def set_vae_encoder(encoder):
    _VAE_ENCODER["encoder"] = encoder


callback_fns = {
    "vae": vae_fn,
    "jpeg": jpeg_fn,
    "neg_jpeg": neg_jpeg_fn,
    "aesthetic": aesthetic_fn,
    "llava_bertscore": llava_bertscore,
    "llava_vqa": llava_vqa_satisfaction,
}
