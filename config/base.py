"""Hyper-parameters for `pipeline/policy_gradient.py` — the `pg` experiment of the reference's flag surface.

Same keys, defaults and precedence as /root/reference/config/base.py:61-102 (`base["pg"]`) and its per-dataset
`common` / `pg` overrides (:106-148, :222-314).  Round 3 adds the `sample` / `train` experiments of the RWR baseline
(`pipeline/sample.py`, `pipeline/finetune.py`; reference :4-60 and the per-dataset `sample` / `train` overrides); `sizes` and
`calibrate` (bucket bookkeeping) stay out.  Values are declarative data; the tables are assembled by helpers so that adding a
dataset is one line.
"""
from . import user

_PG_DEFAULTS = (
    # misc
    ("loadpath", ""), ("load_epoch", "latest"), ("modelpath", "models/pg"), ("savepath", "f:models/pg"),
    ("pretrained_model", "duongna/stable-diffusion-v1-4-flax"), ("resolution", 512), ("filter_field", None),
    ("guidance_scale", 5.0), ("dtype", "float32"), ("cache", "cache"), ("verbose", False), ("seed", 0), ("iteration", 0),
    # sampling (batch sizes are per device)
    ("sample_batch_size", 8), ("num_sample_batches_per_epoch", 1), ("n_inference_steps", 50), ("identical_batch", False),
    ("evaluate", False), ("eta", 1.0),
    # training
    ("train_batch_size", 2), ("train_accumulation_steps", 1), ("num_train_epochs", 200), ("num_inner_epochs", 1),
    ("ppo_clip_range", 1e-4), ("train_cfg", True), ("learning_rate", 1e-5), ("beta1", 0.9), ("beta2", 0.999),
    ("weight_decay", 1e-4), ("epsilon", 1e-8), ("max_grad_norm", 1.0), ("save_freq", 10), ("optimizer", "adamw"),
    ("train_timestep_ratio", 1.0), ("prompt_kwargs", {}), ("per_prompt_stats_bufsize", 32), ("per_prompt_stats_min_count", 16),
    # engine-specific addition (not a reference flag): how optax.adamw(mu_dtype=bfloat16) forms `b1 * mu` — True: in bf16, what JAX's
    # weak-type promotion does (scalar x bf16 array stays bf16: b1 -> 0.8984375, product rounded before the f32 term is added);
    # False: decay in f32, one rounding when mu is stored.  optax is not installable here, so the reading is a derivation, not a measurement.
    # None = not given on the command line: the entrypoint then takes DDPO_MU_DECAY_IN_BF16 (default 1 = True).
    ("mu_decay_in_bf16", None),
)

_SAMPLE_DEFAULTS = (
    ("loadpath", "f:models/{iteration}"), ("savepath", "f:samples/{iteration}"), ("load_epoch", "latest"), ("n_samples_per_device", 4),
    ("pretrained_model", "duongna/stable-diffusion-v1-4-flax"), ("prompt_kwargs", {}), ("n_inference_steps", 50), ("eta", 1.0),
    ("resolution", 512), ("max_samples", 50e3), ("max_steps", None), ("local_size", 1600), ("guidance_scale", 5.0),
    ("filter_field", "labels"), ("mask_mode", "streaming_percentile"), ("mask_param", 95), ("identical_batch", False), ("iteration", 0),
    ("evaluate", False), ("cache", "cache"), ("seed", None),
)
_TRAIN_DEFAULTS = (
    ("modelpath", "f:models/{iteration}"), ("loadpath", "f:samples/{iteration}"), ("savepath", "f:models/{iteration+1}"),
    ("pretrained_model", "duongna/stable-diffusion-v1-4-flax"), ("finetuned_model", None), ("load_epoch", "latest"),
    ("max_train_samples", None), ("resolution", 512), ("train_cfg", False), ("guidance_scale", 5.0), ("train_batch_size", 2),
    ("num_train_epochs", 40), ("max_train_steps", None), ("learning_rate", 1e-5), ("beta1", 0.9), ("beta2", 0.999),
    ("weight_decay", 1e-4), ("epsilon", 1e-8), ("max_grad_norm", 1.0), ("iteration", 0), ("weighted_batch", False),
    ("weighted_dataset", False), ("dtype", "float32"), ("cache", "cache"), ("verbose", False), ("save_freq", 100),
    ("per_prompt_weights", False), ("seed", 0),
)

base = {"sample": dict(_SAMPLE_DEFAULTS), "train": dict(_TRAIN_DEFAULTS), "pg": dict(_PG_DEFAULTS)}

# the two `sample` recipes of the reference's datasets: keep the top decile of 1024 identical-prompt batches (filter-finetune),
# or keep all of 10240 samples (reward-weighted regression)
_SAMPLE_TOP10 = {"max_samples": 1024, "mask_mode": "percentile", "mask_param": 90, "identical_batch": True}
_SAMPLE_ALL = {"max_samples": 10240, "mask_mode": "streaming_percentile", "mask_param": 0, "identical_batch": False}


def _train(train_cfg=True, train_batch_size=1, num_train_epochs=50, save_freq=20, **kw):
    return dict(train_cfg=train_cfg, train_batch_size=train_batch_size, num_train_epochs=num_train_epochs, save_freq=save_freq,
                dtype="float32", **kw)


_TRAIN_RWR = dict(num_train_epochs=5, weighted_dataset=True, temperature=1 / 5.0)


def _dataset(logdir, prompt_fn, filter_field, prompt_kwargs=None, sample=None, train=None, **pg):
    common = {"logbase": f"{user.bucket}/logs/{logdir}", "prompt_fn": prompt_fn, "filter_field": filter_field}
    if prompt_kwargs is not None:
        common["prompt_kwargs"] = prompt_kwargs
    out = {"common": common, "pg": pg}
    if sample is not None:
        out["sample"] = dict(sample)
    if train is not None:
        out["train"] = dict(train)
    return out


_ANIMALS = {"loadpath": "assets/common_animals.txt"}
_NOUNS_ACTIVITIES = {"nouns_path": "assets/common_animals.txt", "activities_path": "assets/activities_v0.txt"}

compressed_animals = _dataset("identical-compressed-animals-s1024-p90", "imagenet_animals", "jpeg", sample=_SAMPLE_TOP10,
                              train=_train(train_batch_size=4))
neg_compressed_animals = _dataset("identical-neg-compressed-animals-s1024-p90", "imagenet_animals", "neg_jpeg", sample=_SAMPLE_TOP10,
                                  train=_train())
# reward-weighted regression on all samples (the RWR baseline of the paper)
compressed_animals_rwr = _dataset("rwr-compressed-animals-s10k", "imagenet_animals", "jpeg", sample=_SAMPLE_ALL, train=_train(**_TRAIN_RWR))
neg_compressed_animals_rwr = _dataset("rwr-neg-compressed-animals-s10k", "imagenet_animals", "neg_jpeg", sample=_SAMPLE_ALL,
                                      train=_train(**_TRAIN_RWR))
a_animals_rwr = _dataset("aesthetic_simple_animals_rwr_ppb", "from_file", "aesthetic", {"loadpath": "assets/common_animals.txt"},
                         sample=_SAMPLE_ALL, train=_train(train_batch_size=4, save_freq=10000000, per_prompt_weights=True, **_TRAIN_RWR))
llava_vqa = _dataset("llava-vqa-v2", "vqa_dataset", "llava_vqa", {"loadpath": "assets/vqa_v2.txt"},
                     per_prompt_stats_bufsize=128, per_prompt_stats_min_count=32, num_train_epochs=120)
llava_counting = _dataset("llava-counting-v0-8", "counting", "llava_vqa",
                          {"nouns_path": "assets/very_simple_animals.txt", "number_range": (2, 8)})
llava_bertscore = _dataset("llava-bertscore-2-simple-animals", "nouns_activities", "llava_bertscore", _NOUNS_ACTIVITIES)
a_dog_1 = _dataset("aesthetic_dogs_sweep/one", "manual", "aesthetic", {"prompts": ["a dog"]}, per_prompt_stats_bufsize=None,
                   per_prompt_stats_min_count=None, train_batch_size=1, train_accumulation_steps=2)
a_dog_2 = _dataset("aesthetic_dogs_sweep/imagenet", "imagenet_dogs", "aesthetic", {}, train_batch_size=1,
                   train_accumulation_steps=2)
a_animals = _dataset("aesthetic_simple_animals", "from_file", "aesthetic", _ANIMALS, sample=_SAMPLE_TOP10, train=_train(),
                     train_batch_size=1, train_accumulation_steps=2)

# CFG-free ablations and the VQA-v0 prompt set of the reference (config/base.py of the reference, same names / overrides)
compressed_animals_nocfg = _dataset("nocfg-compressed-animals-s1024-p90", "imagenet_animals", "jpeg", sample=_SAMPLE_TOP10,
                                    train=_train(train_cfg=False, train_batch_size=2))
neg_compressed_animals_nocfg = _dataset("nocfg-neg-compressed-animals-s1024-p90", "imagenet_animals", "neg_jpeg", sample=_SAMPLE_TOP10,
                                        train=_train(train_cfg=False, train_batch_size=2))
vqa_v0 = _dataset("vqa-v0-n2k-s5.0-e50", "vqa_dataset", "vqa", {"loadpath": "assets/vqa_v0.txt"},
                  sample={"max_samples": 2e3, "mask_mode": "threshold", "mask_param": 0.65, "identical_batch": False},
                  train={"train_cfg": True, "train_batch_size": 1, "num_train_epochs": 50, "save_freq": 20})
