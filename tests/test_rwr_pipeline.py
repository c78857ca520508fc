"""RWR baseline plumbing: the local sample store, maskers and dataset weights (CPU), the VAE encoder and the two
entrypoints pipeline/sample.py -> pipeline/finetune.py end to end on the tiny architecture (GPU)."""
import ast
import json
import os

import numpy as np
import pytest

from ddpo_amd.utils import bucket

REF = "/root/reference"


def _rows(n, seed=0):
    r = np.random.RandomState(seed)
    return {"inference_prompts": [f"a {w}" for w in r.choice(["cat", "dog", "fox"], size=n)],
            "training_prompts": [f"a {w}" for w in r.choice(["cat", "dog", "fox"], size=n)],
            "images": r.rand(n, 16, 16, 3).astype(np.float32), "jpeg": -r.rand(n, 1) * 5, "vae": r.randn(n, 2, 2, 8).astype(np.float32)}


def test_local_writer_reader_round_trip_and_masking(tmp_path):
    from ddpo_amd.training.callbacks import encode_jpeg
    w = bucket.LocalWriter(str(tmp_path), split_size=5, run_id="r")
    w.configure("images", encode_fn=encode_jpeg, decode_fn=bucket.decode_jpeg)
    b1, b2 = _rows(6, 1), _rows(4, 2)
    mask = np.array([1, 0, 1, 1, 0, 1], dtype=bool)
    assert w.add_batch(b1, mask=mask[:, None]) == 4 and w.add_batch(b2) == 4 and len(w) == 8
    w.close(metadata={"guidance_scale": 5.0})
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")) == ["0_r_00000.npz", "0_r_00001.npz"]
    assert json.load(open(tmp_path / "metadata.json"))["guidance_scale"] == 5.0
    rd = bucket.LocalReader(str(tmp_path))
    assert len(rd) == 8
    kept = [i for i in range(6) if mask[i]]
    for j, i in enumerate(kept):
        row = rd[j]
        assert row["inference_prompts"] == b1["inference_prompts"][i] and np.array_equal(row["vae"], b1["vae"][i])
        assert np.array_equal(row["images"], encode_jpeg(b1["images"][i]))            # stored JPEG-encoded, like the reference's `images` field
        assert bucket.decode_jpeg(row["images"]).shape == (16, 16, 3)
    assert np.array_equal(rd[4]["vae"], b2["vae"][0])


def test_rerun_into_the_same_directory_does_not_mix_stale_shards(tmp_path):
    """a second sampling run into the same samples/<iteration> directory with fewer shards / fewer ranks must not leave the first
    run's shards in the RWR training set.  A writer removes ITS rank's old shards; the reader reads what the manifests list, and only the
    ranks of the run that wrote last (manifest `world`)."""
    for rank in (0, 1):                                   # first run: two ranks, three shards each
        w = bucket.LocalWriter(str(tmp_path), split_size=2, rank=rank, run_id="run-a")
        w.add_batch(_rows(6, 10 + rank))
        w.close(metadata={"guidance_scale": 5.0}, world=2)
    assert len(bucket.LocalReader(str(tmp_path))) == 12
    w = bucket.LocalWriter(str(tmp_path), split_size=4, rank=0, run_id="run-b")      # second run: one rank, one shard
    b = _rows(3, 99)
    w.add_batch(b)
    # nothing of the previous dataset is touched before close() — a run that crashes here (or was started by mistake) leaves it readable
    assert len(bucket.LocalReader(str(tmp_path))) == 12
    w.close(metadata={"guidance_scale": 5.0}, world=1)
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("0_")) == ["0_run-b_00000.npz"]    # rank 0's old shards are gone, after close()
    # files of ranks >= the new world are nobody's in this run — rank 0's close() removes them instead of letting them pile up
    assert not [f for f in os.listdir(tmp_path) if f.startswith("1_") or f == "manifest_1.json"]
    rd = bucket.LocalReader(str(tmp_path))
    assert len(rd) == 3 and [rd[i]["inference_prompts"] for i in range(3)] == list(b["inference_prompts"])
    os.remove(tmp_path / "0_run-b_00000.npz")
    with pytest.raises(FileNotFoundError):
        bucket.LocalReader(str(tmp_path))
    # a run whose rank-1 manifest is missing is refused rather than silently halved
    w1 = bucket.LocalWriter(str(tmp_path), split_size=2, rank=1, run_id="run-a")      # (a rank of an earlier two-rank run that closed late)
    w1.add_batch(_rows(2, 11))
    w1.close(world=2)
    w = bucket.LocalWriter(str(tmp_path), split_size=4, rank=0, run_id="run-c")
    w.add_batch(b)
    w.close(world=2)
    with pytest.raises(FileNotFoundError, match="different sampling runs"):      # rank 1's manifest is run-a's: a mixture of two runs is refused
        bucket.LocalReader(str(tmp_path))
    os.remove(tmp_path / "manifest_1.json")
    with pytest.raises(FileNotFoundError):
        bucket.LocalReader(str(tmp_path))
    # writers without an explicit id get distinct ones (timestamp + random suffix), so two independent single-rank runs never collide
    assert bucket.LocalWriter(str(tmp_path)).run_id != bucket.LocalWriter(str(tmp_path)).run_id


def test_maskers_and_dataset_weights(tmp_path):
    xs = np.random.RandomState(0).randn(40, 1)
    m = bucket.make_masker("percentile", 90)
    assert m(xs).sum() == 4 and m.p == np.percentile(xs.squeeze(-1), 90)
    sp = bucket.make_masker("streaming_percentile", 50)
    a, b = sp(xs[:20]), sp(xs[20:])
    assert sp.size == 40 and sp.p == np.percentile(xs.squeeze(-1), 50) and a.shape == (20,) and b.sum() == (xs[20:, 0] >= sp.p).sum()
    assert bucket.make_masker("streaming_percentile", 0)(xs).all()                          # mask_param 0: save all samples (the RWR configs)
    th = bucket.make_masker("threshold", 0.65)
    assert np.array_equal(th(xs), xs >= 0.65)
    avg = bucket.StreamingAverage()
    for i, x in enumerate(xs[:, 0]):
        avg(x)
        assert np.isclose(avg.avg, xs[: i + 1].mean())
    # dataset weights: softmax(reward * temperature) * N over the dataset, or within each prompt's group (hdf5.py:437-451)
    w = bucket.LocalWriter(str(tmp_path), split_size=100)
    batch = _rows(12, 3)
    w.add_batch(batch)
    w.close()
    rd = bucket.LocalReader(str(tmp_path))
    rd.make_weights("jpeg", 0.2, False)
    lab = batch["jpeg"].squeeze()
    ref = np.exp(lab * 0.2 - (lab * 0.2).max())
    ref = ref / ref.sum() * 12
    assert np.allclose(rd.weights, ref) and np.isclose(rd.weights.sum(), 12.0) and "weights" in rd[0]
    rd.make_weights("jpeg", 0.2, True)
    prompts = np.asarray(batch["inference_prompts"])
    for p in np.unique(prompts):
        assert np.isclose(rd.weights[prompts == p].sum(), (prompts == p).sum())             # expected weight 1 per item inside every prompt group


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is only present in the build container")
def test_maskers_equal_the_reference_classes():
    tree = ast.parse(open(os.path.join(REF, "ddpo/utils/logger.py")).read())
    ns = {"np": np}
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in ("Masker", "StreamingAverage", "StreamingPercentile", "Percentile", "Threshold", "make_masker"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), "logger.py", "exec"), ns)
    r = np.random.RandomState(4)
    for mode, param in (("percentile", 90), ("streaming_percentile", 95), ("streaming_percentile", 0), ("threshold", 0.1)):
        a, b = ns["make_masker"](mode, param), bucket.make_masker(mode, param)
        for _ in range(3):
            xs = r.randn(16, 1)
            assert np.array_equal(np.asarray(a(xs)).reshape(-1), np.asarray(b(xs)).reshape(-1)) and repr(a) == repr(b)


def test_parser_evaluates_fstring_expressions_and_the_train_experiment(tmp_path, monkeypatch):
    from ddpo_amd.utils.parser import Parser
    monkeypatch.chdir(tmp_path)

    class P(Parser):
        config = "config.base"
        dataset = "compressed_animals_rwr"
    args = P(["--iteration", "2"]).parse_args("train")
    assert args.savepath.endswith("models/3") and args.loadpath.endswith("samples/2") and args.modelpath.endswith("models/2")     # "f:models/{iteration+1}"
    assert args.weighted_dataset is True and args.temperature == pytest.approx(0.2) and args.num_train_epochs == 5 and args.filter_field == "jpeg"
    s = P([]).parse_args("sample")
    assert s.mask_mode == "streaming_percentile" and s.mask_param == 0 and s.max_samples == 10240 and s.n_samples_per_device == 4


@pytest.mark.gpu
def test_vae_encoder_matches_oracle_tiny_and_sd_shapes():
    import torch
    from ddpo_amd import lib as L
    from ddpo_amd.models.vae import VAEEncoder, VAEConfig
    from oracle import unet as OU
    old = L.DATAPATH
    try:
        for dp, name, ocfg, hw, tol in (("fp32", "tiny", OU.VAE_TINY, 64, 2e-4), ("bf16x3", "tiny", OU.VAE_TINY, 48, 2e-4), ("bf16x3", "sd", OU.VAE_SD, 64, 3e-4)):
            L.DATAPATH = dp
            L.PACKED.clear()
            op = OU.init_params(OU.vae_encoder_param_shapes(ocfg), seed=6)
            enc = VAEEncoder(VAEConfig.named(name), "cuda")
            if name == "sd":
                assert enc.params.n_params == 34163664
            enc.params.load_dict(op)
            if dp != "fp32":
                enc.params.pack_bf16(bwd=False)
            img = torch.rand(2, hw, hw, 3, generator=torch.Generator().manual_seed(3))
            with torch.no_grad():
                ref = OU.vae_encode(op, ocfg, img).numpy()
            got = enc.encode(img.to("cuda")).cpu().numpy()
            assert got.shape == ref.shape == (2, hw // 8, hw // 8, 8)
            err = float(np.abs(got - ref).max() / np.abs(ref).max())
            print(f"\n[vae encoder] {name} {dp} {hw}^2: max rel err {err:.2e}")
            assert err < tol
    finally:
        L.DATAPATH = old
        L.PACKED.clear()


@pytest.mark.gpu
def test_sample_then_finetune_end_to_end_tiny(tmp_path, monkeypatch):
    """pipeline/sample.py writes a reward-labelled dataset (images, prompts, jpeg rewards, VAE moments), pipeline/finetune.py reads it back,
    builds dataset weights (softmax over rewards, temperature 0.2) and runs RWR steps: the loss is finite, decreases over a few epochs on a
    fixed tiny dataset, and a checkpoint in both formats lands under the run directory."""
    import torch
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DDPO_ALLOW_SYNTHETIC", "1")
    monkeypatch.setenv("DDPO_MODEL_CONFIG", "tiny")
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sample = importlib.import_module("pipeline.sample")
    finetune = importlib.import_module("pipeline.finetune")
    common = ["--dataset", "compressed_animals_rwr", "--logbase", str(tmp_path / "logs"), "--resolution", "64", "--seed", "3"]
    savepath = sample.main(common + ["--n_inference_steps", "4", "--n_samples_per_device", "4", "--max_samples", "12", "--local_size", "8"])
    rd = bucket.LocalReader(savepath)
    assert len(rd) == 12 and set(rd[0]) >= {"images", "inference_prompts", "training_prompts", "jpeg", "vae"}
    assert np.asarray(rd[0]["vae"]).shape == (8, 8, 8) and json.load(open(os.path.join(savepath, "metadata.json")))["guidance_scale"] == 5.0
    hist = finetune.main(common + ["--train_batch_size", "4", "--num_train_epochs", "6", "--learning_rate", "3e-4", "--save_freq", "100"])
    print(f"\n[rwr end to end, tiny] epoch losses {['%.4f' % h for h in hist]}")
    assert len(hist) == 6 and all(np.isfinite(hist)) and hist[-1] < hist[0]
    ck = tmp_path / "logs" / "models" / "1" / "checkpoints"
    assert any(f.startswith("checkpoint_") for f in os.listdir(ck))
