#!/usr/bin/env python
"""Generates tests/golden/reference_host_logic.json by EXECUTING the pure-Python parts of the reference in place.

Run in the build container only (`python tests/golden/make_reference_goldens.py`); it reads /root/reference (read-only)
and writes one JSON fixture.  Tests and the GPU box never touch /root/reference: they compare `ddpo_amd` against the
committed fixture (tests/test_reference_goldens.py).

What runs from the reference, unmodified, loaded by file path:
  * ddpo/utils/stat_tracking.py            (numpy only)                      -> PerPromptStatTracker advantages / stats
  * ddpo/utils/imagenet.py                 (literals only)                   -> class / colour lists
  * ddpo/training/prompts.py               (needs `ddpo.utils`, `inflect`)   -> prompt streams under random.seed(s)
  * config/base.py + config/user.py        (literals only)                   -> the `pg` flag surface per dataset
  * encode_jpeg (ddpo/utils/hdf5.py) and jpeg_fn / neg_jpeg_fn (ddpo/training/callbacks.py), lifted with `ast` like the
    loaders below                                                              -> JPEG-size rewards of seeded images
    (byte counts depend on the PIL / libjpeg build: the fixture records PIL's version and the test skips on another)
  * llava_bertscore / llava_vqa_satisfaction (+ single_satisfaction), lifted the same way, with requests.Session.post
    intercepted: the REQUEST each batch would send (pickled dict: JPEG q=80 bytes, queries, answers; np.array_split
    batching) and what the callback returns for a scripted reply                -> the LLaVA wire protocol
The reference's `ddpo.utils` package cannot be imported here (jax / flax / gcsfs / h5py missing), so prompts.py gets a
stand-in `ddpo.utils` whose `load_lines` / `load_general_prompts` are the reference's OWN function bodies, lifted out of
ddpo/utils/serialization.py with `ast` and exec'd unchanged.  `inflect` is not installable offline: a three-call
stand-in (a / number_to_words / plural) with plain English rules is injected — the only part of the fixture that is not
reference code; it affects the article of `nouns_activities` and the words of `counting`.
"""
import ast
import importlib.util
import json
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_host_logic.json")


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def lift_functions(path, names, extra=None):
    """exec only the named top-level functions of a reference file (its module-level imports are not executed)."""
    tree = ast.parse(open(path).read())
    ns = {"__builtins__": __builtins__}
    ns.update(extra or {})
    import functools as _functools
    import re as _re
    ns.update(re=_re, functools=_functools)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


class _Inflect:                       # stand-in, see module docstring
    _irregular = {"mouse": "mice", "goose": "geese", "sheep": "sheep", "deer": "deer", "fish": "fish", "wolf": "wolves", "fox": "foxes"}
    _words = "zero one two three four five six seven eight nine ten eleven twelve".split()

    def a(self, noun):
        return ("an " if noun[:1].lower() in "aeiou" else "a ") + noun

    def number_to_words(self, n):
        return self._words[n] if 0 <= n < len(self._words) else str(n)

    def plural(self, noun):
        if noun in self._irregular:
            return self._irregular[noun]
        if noun.endswith(("s", "x", "z", "ch", "sh")):
            return noun + "es"
        if noun.endswith("y") and noun[-2:-1] not in "aeiou":
            return noun[:-1] + "ies"
        return noun + "s"


def jpeg_test_images(seed, n, hw):
    """float32 (n, hw, hw, 3) in [0,1]: smooth gradients + seeded noise (shared recipe with tests/test_reference_goldens.py)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, hw), np.linspace(0, 1, hw), indexing="ij")
    imgs = []
    for i in range(n):
        base = np.stack([yy, xx, 0.5 + 0.5 * np.sin(6.0 * (xx + yy) + i)], axis=-1)
        img = np.clip(base + rng.randn(hw, hw, 3) * 0.05 * (i + 1), 0.0, 1.0)
        imgs.append(img.astype(np.float32))
    return np.stack(imgs)


def jsonable(x):
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    if isinstance(x, np.ndarray):
        return x.tolist()
    return x


def main():
    os.chdir(REF)                                           # the reference resolves "assets/..." relative to its root
    out = {"generated_by": "tests/golden/make_reference_goldens.py", "reference": "jannerm/ddpo @ /root/reference"}

    # ---------------------------------------------------------------- stat tracker
    st = load_by_path("ref_stat_tracking", os.path.join(REF, "ddpo/utils/stat_tracking.py"))
    cases = []
    for (buf, mc, n_prompts, n_batches, bs, seed) in [(32, 16, 3, 8, 16, 0), (4, 2, 2, 6, 6, 1), (8, 8, 5, 5, 8, 2)]:
        rng = np.random.RandomState(seed)
        tr = st.PerPromptStatTracker(buf, mc)
        names = np.array([f"prompt {i}" for i in range(n_prompts)])
        steps = []
        for _ in range(n_batches):
            prompts = names[rng.randint(0, n_prompts, size=bs)]
            rewards = rng.randn(bs) * 3.0 + rng.randint(-2, 3)
            adv = tr.update(prompts, rewards.copy())
            steps.append({"prompts": prompts.tolist(), "rewards": rewards.tolist(), "advantages": np.asarray(adv).tolist()})
        cases.append({"buffer_size": buf, "min_count": mc, "steps": steps, "stats": jsonable(tr.get_stats())})
    out["stat_tracker"] = cases

    # ---------------------------------------------------------------- prompts
    imagenet = load_by_path("ref_imagenet", os.path.join(REF, "ddpo/utils/imagenet.py"))
    lifted = lift_functions(os.path.join(REF, "ddpo/utils/serialization.py"), ["load_lines", "load_general_prompts"])
    fake_utils = types.ModuleType("ddpo.utils")
    fake_utils.load_lines = lifted["load_lines"]
    fake_utils.load_general_prompts = lifted["load_general_prompts"]
    fake_utils.imagenet = imagenet
    fake_pkg = types.ModuleType("ddpo")
    fake_pkg.utils = fake_utils
    fake_pkg.__path__ = []
    sys.modules.update({"ddpo": fake_pkg, "ddpo.utils": fake_utils, "ddpo.utils.imagenet": imagenet})
    fake_inflect = types.ModuleType("inflect")
    fake_inflect.engine = _Inflect
    sys.modules["inflect"] = fake_inflect
    P = load_by_path("ref_prompts", os.path.join(REF, "ddpo/training/prompts.py"))

    cls = imagenet.classes                                   # dict index -> label in the reference
    out["imagenet"] = {"n_classes": len(cls), "classes": {str(i): cls[i] for i in sorted(cls)},
                       "n_colors": len(imagenet.colors), "colors": list(imagenet.colors)}
    specs = [
        ("imagenet_animals", {}), ("imagenet_dogs", {}), ("imagenet_aesthetic", {}), ("imagenet_single", {}), ("imagenet_simple", {"idx": 7}),
        ("simple_dogs", {}), ("animal_debug", {}), ("person_pet", {}), ("consistent_animals", {}), ("n_fingers", {}),
        ("consistent_imagenet_animals", {"colors": True}), ("consistent_imagenet_animals_3", {"colors": False}),
        ("from_file", {"loadpath": "assets/common_animals.txt"}), ("manual", {"prompts": ["a dog", "a cat", "a bird"]}),
        ("nouns_activities", {"nouns_path": "assets/common_animals.txt", "activities_path": "assets/activities_v0.txt"}),
        ("counting", {"nouns_path": "assets/very_simple_animals.txt", "number_range": [2, 5]}),
        ("vqa_dataset", {"loadpath": "assets/vqa_v0.txt"}), ("vqa_dataset", {"loadpath": "assets/vqa_v2.txt"}),
    ]
    streams = []
    for fn, kwargs in specs:
        if not hasattr(P, fn):
            continue
        for seed, identical in [(0, False), (42, False), (7, True)]:
            random.seed(seed)
            kw = dict(kwargs)
            if fn not in ("consistent_imagenet_animals", "consistent_imagenet_animals_3"):
                kw["evaluate"] = False
            inf, train, meta = P.make_prompts(fn, 6, identical, **kw)
            after = random.random()                          # pins how many draws the call consumed
            streams.append({"fn": fn, "kwargs": jsonable(kwargs), "seed": seed, "identical_batch": identical, "batch_size": 6,
                            "inference": list(inf), "training": jsonable(train), "metadata": jsonable(meta), "next_random": after})
    out["prompt_streams"] = streams

    # ---------------------------------------------------------------- config
    cfg_pkg = types.ModuleType("ref_config")
    cfg_pkg.__path__ = [os.path.join(REF, "config")]
    sys.modules["ref_config"] = cfg_pkg
    load_by_path("ref_config.user", os.path.join(REF, "config/user.py"))
    base_mod = load_by_path("ref_config.base", os.path.join(REF, "config/base.py"))
    cfg = {"base_pg": jsonable(base_mod.base["pg"]), "base_sample": jsonable(base_mod.base["sample"]), "base_train": jsonable(base_mod.base["train"]),
           "datasets": {}}
    for name in dir(base_mod):
        val = getattr(base_mod, name)
        if name.startswith("_") or name == "base" or not isinstance(val, dict):
            continue
        if "pg" in val or "common" in val:
            cfg["datasets"][name] = {"common": jsonable(val.get("common", {})), "pg": jsonable(val.get("pg", {})),
                                     "sample": jsonable(val.get("sample")), "train": jsonable(val.get("train"))}
    out["config"] = cfg

    # ---------------------------------------------------------------- jpeg rewards
    import io
    import PIL
    from PIL import Image
    enc = lift_functions(os.path.join(REF, "ddpo/utils/hdf5.py"), ["encode_jpeg"], {"np": np, "io": io, "Image": Image})
    ref_utils = types.SimpleNamespace(encode_jpeg=enc["encode_jpeg"])
    cb = lift_functions(os.path.join(REF, "ddpo/training/callbacks.py"), ["jpeg_fn", "neg_jpeg_fn"],
                        {"np": np, "utils": ref_utils, "DEVICES": None})
    jcases = []
    for seed, n, hw in [(0, 3, 64), (1, 2, 96), (2, 1, 512)]:
        images = jpeg_test_images(seed, n, hw)
        s_jpeg, _ = cb["jpeg_fn"]()(images, None, None)
        s_neg, _ = cb["neg_jpeg_fn"]()(images, None, None)
        jcases.append({"seed": seed, "n": n, "hw": hw, "jpeg": np.asarray(s_jpeg).tolist(), "neg_jpeg": np.asarray(s_neg).tolist(),
                       "dtype": str(np.asarray(s_jpeg).dtype), "shape": list(np.asarray(s_jpeg).shape)})
    out["jpeg_rewards"] = {"pil_version": PIL.__version__, "cases": jcases}

    # ---------------------------------------------------------------- LLaVA reward wire protocol
    import hashlib
    import pickle
    import requests
    cbl = lift_functions(os.path.join(REF, "ddpo/training/callbacks.py"), ["single_satisfaction", "llava_vqa_satisfaction", "llava_bertscore"],
                         {"np": np, "Image": Image, "DEVICES": None})
    cbl["llava_vqa_satisfaction"].__globals__["single_satisfaction"] = cbl["single_satisfaction"]
    captured = []

    def fake_post(self, url, data=None, timeout=None, **kw):
        req = pickle.loads(data)
        n = len(req["images"])
        i0 = sum(len(c["images_sha256"]) for c in captured)
        captured.append({"url": url, "timeout": timeout, "keys": sorted(req), "queries": req["queries"], "answers": req.get("answers"),
                         "images_sha256": [hashlib.sha256(b).hexdigest() for b in req["images"]], "images_len": [len(b) for b in req["images"]]})
        if "answers" in req:          # bertscore server
            rep = {"recall": [[0.05 * (i0 + i) + 0.1] for i in range(n)], "precision": [[0.9 - 0.01 * (i0 + i)] for i in range(n)],
                   "f1": [[0.5 + 0.002 * (i0 + i)] for i in range(n)], "outputs": [[f"a picture of thing {i0 + i}"] for i in range(n)]}
        else:                         # vqa server: one answer string per question
            rep = {"outputs": [[("It is a Cat." if (i0 + i + j) % 3 == 0 else "riding a bike") for j in range(len(req["queries"][i]))] for i in range(n)]}
        return types.SimpleNamespace(content=pickle.dumps(rep), status_code=200)

    real_post = requests.Session.post
    requests.Session.post = fake_post
    try:
        imgs = jpeg_test_images(5, 20, 32)
        prompts = np.array([f"a {w} riding a bike" for w in ("cat dog horse monkey rabbit spider bird sheep cow lion tiger bear raccoon fox wolf ant fish "
                                                              "squirrel turtle frog").split()])
        captured.clear()
        sc, info = cbl["llava_bertscore"]()(imgs.copy(), prompts, None)
        out["llava_bertscore"] = {"n": 20, "hw": 32, "seed": 5, "prompts": prompts.tolist(), "requests": list(captured),
                                  "scores": np.asarray(sc).tolist(), "info": {k: np.asarray(v).tolist() for k, v in info.items()},
                                  "info_dtypes": {k: str(np.asarray(v).dtype.kind) for k, v in info.items()}}
        meta = [{"questions": ["what animal is this?", "what is it doing?"], "answers": ["Cat", "bike"], "prompt": str(p)} for p in prompts[:10]]
        captured.clear()
        sc, info = cbl["llava_vqa_satisfaction"]()(imgs[:10].copy(), None, meta)
        out["llava_vqa"] = {"n": 10, "hw": 32, "seed": 5, "metadata": meta, "requests": list(captured), "scores": np.asarray(sc).tolist(),
                            "info": {k: np.asarray(v).tolist() for k, v in info.items()},
                            "info_shapes": {k: list(np.asarray(v).shape) for k, v in info.items()}}
    finally:
        requests.Session.post = real_post

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(streams)} prompt streams, {len(cases)} tracker cases, {len(cfg['datasets'])} dataset configs")


if __name__ == "__main__":
    main()
