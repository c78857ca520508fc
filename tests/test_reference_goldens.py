"""Host logic of the drop-in surfaces against outputs of the REFERENCE'S OWN CODE.

tests/golden/reference_host_logic.json was produced by tests/golden/make_reference_goldens.py, which executes the
reference's stat tracker, ImageNet tables, prompt functions (under seeded `random`) and config module in place.
These tests never read /root/reference; they compare `ddpo_amd` / `config` with the committed fixture:
prompt streams and consumed-draw counts bit-exact (strings / ints), advantages to 1e-12 (float64 numpy on both sides)."""
import json
import os
import random

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_host_logic.json")))


def test_stat_tracker_matches_reference_runs():
    from ddpo_amd.utils.stat_tracking import PerPromptStatTracker
    for case in GOLD["stat_tracker"]:
        tr = PerPromptStatTracker(case["buffer_size"], case["min_count"])
        for step in case["steps"]:
            adv = tr.update(np.array(step["prompts"]), np.array(step["rewards"], dtype=np.float64))
            np.testing.assert_allclose(adv, np.array(step["advantages"]), rtol=1e-12, atol=1e-12)
        stats = tr.get_stats()
        assert set(stats) == set(case["stats"])
        for k, v in case["stats"].items():
            assert stats[k]["count"] == v["count"]
            assert stats[k]["mean"] == pytest.approx(v["mean"], rel=1e-12, abs=1e-12)
            assert stats[k]["std"] == pytest.approx(v["std"], rel=1e-12, abs=1e-12)


def test_imagenet_tables_match_reference():
    from ddpo_amd.training import prompts as P
    classes = P.imagenet.classes
    assert len(classes) == GOLD["imagenet"]["n_classes"] == 1000
    for i, label in GOLD["imagenet"]["classes"].items():
        assert classes[int(i)] == label
    assert list(P.imagenet.colors) == GOLD["imagenet"]["colors"]


@pytest.mark.parametrize("idx", range(len(GOLD["prompt_streams"])))
def test_prompt_stream_matches_reference(idx):
    """Same strings, same metadata AND the same number of draws from Python's global `random` (the next random()
    after the call is pinned), for every prompt_fn a reference config names plus the other generators of the module."""
    from ddpo_amd.training import prompts as P
    g = GOLD["prompt_streams"][idx]
    kw = dict(g["kwargs"])
    if g["fn"] not in ("consistent_imagenet_animals", "consistent_imagenet_animals_3"):
        kw["evaluate"] = False
    random.seed(g["seed"])
    inf, train, meta = P.make_prompts(g["fn"], g["batch_size"], g["identical_batch"], **kw)
    nxt = random.random()
    assert list(inf) == g["inference"]
    assert [list(t) for t in train] == g["training"]
    assert [dict(m) for m in meta] == g["metadata"]
    assert nxt == g["next_random"]


def _norm(v):
    return json.loads(json.dumps(v))          # tuples -> lists, like the fixture


def test_config_flag_surface_matches_reference():
    """Every `pg` key and default of the reference's config/base.py, and every dataset's overrides, are present here."""
    from config import base as C
    ref = GOLD["config"]
    mine = _norm(C.base["pg"])
    for k, v in ref["base_pg"].items():
        assert k in mine, f"missing pg flag {k}"
        if k in ("logbase",):                  # site-specific (the reference points at a GCS bucket)
            continue
        assert mine[k] == v, f"default of {k}: {mine[k]!r} != reference {v!r}"
    # the `sample` / `train` experiments of the RWR baseline (pipeline/sample.py, pipeline/finetune.py): defaults and per-dataset overrides equal
    for sect in ("sample", "train"):
        assert _norm(C.base[sect]) == ref["base_" + sect], sect
    for name, ds in ref["datasets"].items():
        assert hasattr(C, name), f"missing dataset config {name}"
        got = getattr(C, name)
        for sect in ("sample", "train"):
            assert _norm(got.get(sect)) == ds[sect], f"{name}.{sect}"
        for sect in ("common", "pg"):
            for k, v in ds[sect].items():
                mine_v = _norm(got.get(sect, {}).get(k))
                if k == "logbase":             # site-specific root (GCS bucket vs local directory); the run sub-path must agree
                    assert mine_v.split("logs/", 1)[1] == v.split("logs/", 1)[1]
                    continue
                assert mine_v == v, f"{name}.{sect}.{k}: {mine_v!r} != {v!r}"


def _jpeg_test_images(seed, n, hw):
    """Same recipe as tests/golden/make_reference_goldens.py:jpeg_test_images."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, hw), np.linspace(0, 1, hw), indexing="ij")
    imgs = []
    for i in range(n):
        base = np.stack([yy, xx, 0.5 + 0.5 * np.sin(6.0 * (xx + yy) + i)], axis=-1)
        img = np.clip(base + rng.randn(hw, hw, 3) * 0.05 * (i + 1), 0.0, 1.0)
        imgs.append(img.astype(np.float32))
    return np.stack(imgs)


def test_jpeg_rewards_match_reference_functions():
    """jpeg / neg_jpeg rewards: integer byte counts / 1000 — bit-exact against the reference's own encode_jpeg + jpeg_fn
    (executed by the fixture generator) for identical pixels and the same PIL / libjpeg build."""
    import PIL
    from ddpo_amd.training import callbacks as CB
    ref = GOLD["jpeg_rewards"]
    if PIL.__version__ != ref["pil_version"]:
        pytest.skip(f"fixture was produced with PIL {ref['pil_version']}, this is {PIL.__version__}")
    for case in ref["cases"]:
        images = _jpeg_test_images(case["seed"], case["n"], case["hw"])
        for name in ("jpeg", "neg_jpeg"):
            scores, info = CB.callback_fns[name]()(images, ["p"] * case["n"], ({},) * case["n"])
            scores = np.asarray(scores)
            assert list(scores.shape) == case["shape"] and str(scores.dtype) == case["dtype"] and info == {}
            assert scores.tolist() == case[name]


def _patched_post(captured, kind):
    import hashlib
    import pickle
    import types

    def fake_post(self, url, data=None, timeout=None, **kw):
        req = pickle.loads(data)
        n = len(req["images"])
        i0 = sum(len(c["images_sha256"]) for c in captured)
        captured.append({"url": url, "timeout": timeout, "keys": sorted(req), "queries": req["queries"], "answers": req.get("answers"),
                         "images_sha256": [hashlib.sha256(b).hexdigest() for b in req["images"]], "images_len": [len(b) for b in req["images"]]})
        if kind == "bertscore":
            rep = {"recall": [[0.05 * (i0 + i) + 0.1] for i in range(n)], "precision": [[0.9 - 0.01 * (i0 + i)] for i in range(n)],
                   "f1": [[0.5 + 0.002 * (i0 + i)] for i in range(n)], "outputs": [[f"a picture of thing {i0 + i}"] for i in range(n)]}
        else:
            rep = {"outputs": [[("It is a Cat." if (i0 + i + j) % 3 == 0 else "riding a bike") for j in range(len(req["queries"][i]))] for i in range(n)]}
        return types.SimpleNamespace(content=pickle.dumps(rep), status_code=200)
    return fake_post


@pytest.mark.parametrize("kind", ["bertscore", "vqa"])
def test_llava_wire_protocol_matches_reference_functions(kind, monkeypatch):
    """The LLaVA reward callbacks against the reference's own functions (lifted + run by the fixture generator with
    requests.Session.post intercepted): identical requests per batch (keys, queries, answers, JPEG q=80 bytes by sha256,
    np.array_split batching, URL, timeout) and identical (scores, info) for the same scripted server replies."""
    import PIL
    import requests
    from ddpo_amd.training import callbacks as CB
    ref = GOLD["llava_bertscore" if kind == "bertscore" else "llava_vqa"]
    if PIL.__version__ != GOLD["jpeg_rewards"]["pil_version"]:
        pytest.skip("JPEG bytes depend on the PIL build the fixture was produced with")
    captured = []
    monkeypatch.setattr(requests.Session, "post", _patched_post(captured, kind))
    images = _jpeg_test_images(ref["seed"], 20, ref["hw"])[:ref["n"]]
    if kind == "bertscore":
        scores, info = CB.callback_fns["llava_bertscore"]()(images, np.array(ref["prompts"]), None)
    else:
        scores, info = CB.callback_fns["llava_vqa"]()(images, None, ref["metadata"])
    assert json.loads(json.dumps(captured)) == ref["requests"]
    assert np.asarray(scores).tolist() == ref["scores"]
    assert sorted(info) == sorted(ref["info"])
    for k, v in ref["info"].items():
        assert np.asarray(info[k]).tolist() == v
