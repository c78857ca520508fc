"""CPU tests of the host-side mirror of the reference's plugin / flag / advantage logic (no GPU, no oracle needed)."""
import json
import os
import pickle
import random
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ prompts
def test_prompt_fns_follow_reference_rng_call_order():
    from ddpo_amd.training import prompts as P
    labels = [l.strip() for l in open(os.path.join(ROOT, "assets", "imagenet_labels.txt"))]
    assert len(labels) == 1000 and labels[0] == "tench, Tinca tinca" and labels[397] == "puffer, pufferfish, blowfish, globefish"
    random.seed(123)
    got, training, meta = P.make_prompts("imagenet_animals", 5, False, evaluate=False)
    random.seed(123)
    want = []
    for _ in range(5):                       # reference: randint(0, 397), then random.choice of the 1-element list
        c = labels[random.randint(0, 397)]
        want.append(random.choice([c]))
    assert got == want and all(t == [g] for t, g in zip(training, got)) and meta == ({},) * 5
    nouns = [l.strip() for l in open(os.path.join(ROOT, "assets", "common_animals.txt"))]
    acts = [l.strip() for l in open(os.path.join(ROOT, "assets", "activities_v0.txt"))]
    assert len(nouns) == 45 and acts == ["washing the dishes", "riding a bike", "playing chess"]
    random.seed(7)
    got = P.make_prompts("nouns_activities", 4, False, nouns_path="assets/common_animals.txt",
                         activities_path="assets/activities_v0.txt", evaluate=False)[0]
    random.seed(7)
    want = []
    for _ in range(4):
        n = random.choice(nouns); a = random.choice(acts)
        want.append(("an " if n[0] in "aeiou" else "a ") + n + " " + a)
    assert got == want
    random.seed(7)
    same = P.make_prompts("from_file", 3, True, loadpath="assets/common_animals.txt", evaluate=False)
    assert len(set(same[0])) == 1 and same[0][0] in nouns      # identical_batch: ONE draw replicated
    with pytest.raises(KeyError):
        P.make_prompts("no_such_prompt_fn", 1)

    @P.register
    def my_plugin(evaluate=False, word="x"):
        return f"a {word}", [f"a {word}"], {"k": 1}
    assert P.make_prompts("my_plugin", 2, False, word="cat", evaluate=False)[0] == ["a cat", "a cat"]


# ------------------------------------------------------------------------------------------------ parser / config
def test_parser_precedence_and_casting(tmp_path):
    from ddpo_amd.utils.parser import Parser
    a = Parser(["--dataset", "a-animals", "--logbase", str(tmp_path), "--sample_batch_size", "4", "--train_cfg", "False",
                "--per_prompt_stats_bufsize", "None", "--eta", "0.25", "--seed", "5"]).parse_args("pg", process_index=2)
    assert a.prompt_fn == "from_file" and a.filter_field == "aesthetic" and a.prompt_kwargs == {"loadpath": "assets/common_animals.txt"}
    assert a.train_batch_size == 1 and a.train_accumulation_steps == 2          # dataset["pg"] over base["pg"]
    assert a.sample_batch_size == 4 and a.train_cfg is False and a.per_prompt_stats_bufsize is None and a.eta == 0.25
    assert a.seed == 7                                                            # seed + process index
    assert a.savepath == os.path.join(str(tmp_path), "models/pg") and os.path.isdir(a.savepath)
    assert a._dict["learning_rate"] == 1e-5 and a._dict["ppo_clip_range"] == 1e-4 and a._dict["n_inference_steps"] == 50
    json.dumps(a._dict, default=str)
    with pytest.raises(AssertionError):
        Parser(["--dataset", "compressed_animals", "--not_a_flag", "1"]).parse_args("pg")
    b = Parser(["--dataset", "compressed-animals", "--logbase", str(tmp_path)]).parse_args("pg")
    assert b.filter_field == "jpeg" and b.prompt_fn == "imagenet_animals" and b.per_prompt_stats_bufsize == 32


# ------------------------------------------------------------------------------------------------ advantages
def test_per_prompt_stat_tracker_semantics():
    from ddpo_amd.utils.stat_tracking import PerPromptStatTracker
    tr = PerPromptStatTracker(buffer_size=4, min_count=3)
    prompts = np.array(["a", "b", "a", "b"])
    r = np.array([[1.0], [2.0], [3.0], [6.0]])
    adv = tr.update(prompts, r)                       # fewer than min_count entries: batch-global statistics
    np.testing.assert_allclose(adv, (r - r.mean()) / (r.std() + 1e-6))
    adv2 = tr.update(prompts, r)                      # now 4 >= 3 per prompt: per-prompt buffers
    for p in ("a", "b"):
        buf = np.concatenate([r[prompts == p], r[prompts == p]])
        np.testing.assert_allclose(adv2[prompts == p], (r[prompts == p] - buf.mean()) / (buf.std() + 1e-6))
    tr.update(prompts, r)
    assert tr.get_stats()["a"]["count"] == 4          # ring buffer capped at buffer_size
    tr2 = PerPromptStatTracker(4, 3)
    tr2.load_state_dict(tr.state_dict())
    np.testing.assert_allclose(tr2.update(prompts, r), tr.update(prompts, r))


def test_jpeg_reward_is_integer_exact_and_signed():
    from ddpo_amd.training import callback_fns, evaluate_callbacks
    from ddpo_amd.training.callbacks import encode_jpeg
    import io
    from PIL import Image
    rng = np.random.default_rng(0)
    imgs = rng.random((3, 32, 32, 3)).astype(np.float32)
    out = evaluate_callbacks({"jpeg": callback_fns["jpeg"](), "neg_jpeg": callback_fns["neg_jpeg"]()}, imgs, ["p"] * 3, ({},) * 3)
    s, sneg = out["jpeg"][0], out["neg_jpeg"][0]
    assert s.shape == (3, 1) and s.dtype == np.float64 and np.array_equal(s, -sneg)
    for im, sc in zip(imgs, s[:, 0]):
        buf = io.BytesIO()
        Image.fromarray((im * 255).astype(np.uint8)).save(buf, "JPEG", quality=95)     # truncation, q=95
        assert sc == -len(buf.getvalue()) / 1000.0 and len(encode_jpeg(im)) == len(buf.getvalue())
    flat = np.full((1, 32, 32, 3), 0.5, dtype=np.float32)
    assert callback_fns["jpeg"]()(flat, None, None)[0][0, 0] > s.max()                  # flat image compresses better


def test_llava_bertscore_wire_format_against_stub():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import llava_stub_server
    from ddpo_amd.training.callbacks import llava_bertscore, llava_vqa_satisfaction
    srv = llava_stub_server.serve(0)
    port = srv.server_address[1]
    th = threading.Thread(target=srv.serve_forever, daemon=True)
    th.start()
    try:
        rng = np.random.default_rng(1)
        imgs = rng.random((20, 16, 16, 3)).astype(np.float32)          # 2 chunks of <=16
        fn = llava_bertscore(url=f"http://127.0.0.1:{port}")
        scores, info = fn(imgs, [f"a cat {i}" for i in range(20)], None)
        assert scores.shape == (20,) and set(info) == {"precision", "f1", "outputs"} and info["precision"].shape == (20,)
        np.testing.assert_allclose(info["precision"], scores / 2)
        vq = llava_vqa_satisfaction(url=f"http://127.0.0.1:{port}")
        meta = tuple({"questions": ["is it?", "what?"], "answers": ["yes", "cat"]} for _ in range(5))
        s2, i2 = vq(imgs[:5], None, meta)
        np.testing.assert_allclose(s2, 0.5)
    finally:
        srv.shutdown()


def test_global_advantage_normalisation_and_local_slice():
    from ddpo_amd.training import distributed as D
    r = np.arange(8, dtype=np.float64)[:, None]
    adv = (r - r.mean()) / r.std()
    assert np.array_equal(D.local_slice(adv, 1, 4), adv.reshape(4, -1)[1])
    assert np.array_equal(D.allgather_array(r), r) and D.allgather_strings(["a"]) == ["a"]


def test_byte_tokenizer_framing():
    from ddpo_amd.models.text import ByteTokenizer, make_uncond_text
    t = ByteTokenizer()
    ids = t(["a cat", ""], padding="max_length", max_length=77, truncation=True, return_tensors="np").input_ids
    assert ids.shape == (2, 77) and ids[0, 0] == 49406 and (ids[1, 1:] == 49407).all()
    assert t.batch_decode(ids) == ["a cat", ""] and np.array_equal(make_uncond_text(t, 1)[0], ids[1])


@pytest.mark.parametrize("BM,BN,NW", [(128, 320, 8), (128, 128, 4), (128, 64, 4), (256, 320, 8)])
def test_lds_dma_piece_map_reproduces_the_swizzled_lds_image(BM, BN, NW):
    """Index algebra of the plane-fed GEMM's operand fill (csrc/gemm_bf16.hip, APL path), modelled lane by lane: every
    16-byte slot of every tile row is written exactly once per plane, by the lane whose SOURCE chunk is the one swz_off()
    places there (LDS-DMA writes lane-linearly: 64 lanes x 16 B = 16 rows x 64 B per wave instruction)."""
    PAIRS = NW // 2
    for rows, base in ((BM, 0), (BN, 2 * BM * 64)):              # A planes, then W planes: [hi | lo] each, 64 B per row
        G = rows // 16
        assert G % PAIRS == 0
        pieces = G // PAIRS
        plane_bytes = rows * 64
        seen = {}
        for w in range(NW):
            plane, pr = w & 1, w >> 1
            for i in range(pieces):
                lds_piece = base + plane * plane_bytes + pr * 1024 + i * PAIRS * 1024
                for lane in range(64):
                    src_row = 16 * (pr + PAIRS * i) + (lane >> 2)
                    src_chunk = (lane & 3) ^ ((lane >> 4) & 3)
                    addr = lds_piece + lane * 16                                  # lane-linear destination
                    off = addr - base - plane * plane_bytes
                    row, slot = off // 64, (off % 64) // 16
                    assert row == src_row
                    assert slot == src_chunk ^ ((row >> 2) & 3)                   # == swz_off(row, chunk)
                    assert (plane, row, slot) not in seen
                    seen[(plane, row, slot)] = (w, lane)
        assert len(seen) == 2 * rows * 4


def test_train_fuse_default_is_capped_by_the_validated_activation_footprint(monkeypatch):
    from ddpo_amd.training.policy_gradient import train_fuse_default
    monkeypatch.delenv("DDPO_TRAIN_FUSE", raising=False)
    assert train_fuse_default() == 16
    assert train_fuse_default(4, 64 * 64) == 16          # defaults: 2 samples x CFG at 64x64 latents -> U-Net batch 64 per launch (4 whole rounds of 256x320 tiles)
    assert train_fuse_default(4, 96 * 96) == 7           # SD-2.1 at 768^2: no more latent pixels per launch than that
    assert train_fuse_default(16, 96 * 96) == 1
    assert train_fuse_default(4, 8 * 8) == 16
    monkeypatch.setenv("DDPO_TRAIN_FUSE", "25")          # an explicit request is taken as is
    assert train_fuse_default(4, 96 * 96) == 25
    monkeypatch.setenv("DDPO_TRAIN_FUSE", "0")
    assert train_fuse_default() == 1


def test_aesthetic_reward_refuses_to_score_without_weights(monkeypatch, tmp_path):
    """every `a_*` run used to optimise a random-init reward model silently.  Without weights the factory now raises (no GPU
    needed to find that out) unless DDPO_ALLOW_SYNTHETIC=1."""
    import pytest
    from ddpo_amd.models import laion
    monkeypatch.delenv("DDPO_ALLOW_SYNTHETIC", raising=False)
    monkeypatch.delenv("DDPO_AESTHETIC_WEIGHTS", raising=False)
    monkeypatch.setenv("HF_HOME", str(tmp_path))
    with pytest.raises(FileNotFoundError, match="DDPO_ALLOW_SYNTHETIC"):
        laion.AestheticScorer(cache=str(tmp_path / "cache"))
    # the reference's own location for the MLP file is honoured: <cache>/sac+logos+ava1-l14-linearMSE.pth
    (tmp_path / "cache").mkdir()
    (tmp_path / "cache" / laion.MLP_FILE).write_bytes(b"x")
    assert laion.find_weights(None, str(tmp_path / "cache"))[1] == str(tmp_path / "cache" / laion.MLP_FILE)
    (tmp_path / "w" / "clip").mkdir(parents=True)
    assert laion.find_weights(str(tmp_path / "w"), str(tmp_path / "cache"))[0] == str(tmp_path / "w" / "clip")


def test_planes_pay_rule_and_bench_train_fuse_default(monkeypatch):
    """Host-side speed rules of round 2: a layer is plane-fed only where the plane-fed kernel is the faster one in the model (long
    reductions, or >= 32768 rows), and bench.py's --train-fuse follows the entrypoint's geometry rule unless given."""
    import torch
    from ddpo_amd import lib as L
    monkeypatch.setattr(L, "PLANES", True)
    monkeypatch.setattr(L, "PLANES_ALL", False)
    monkeypatch.setattr(L, "DATAPATH", "bf16x3")
    mk = lambda K, N: torch.zeros(1)
    entries = {}
    def reg(K, N):
        w = torch.zeros(1)
        entries[w.data_ptr()] = dict(fwd=(None, None, K), bwd=None, K=K, N=N)
        return w
    monkeypatch.setattr(L, "PACKED", entries)
    conv = reg(2880, 320)          # 3x3 conv 320 -> 320: long reduction
    lin0 = reg(320, 320)           # 64x64-level projection
    lin2 = reg(1280, 1280)         # 16x16-level projection
    ff2 = reg(5120, 1280)          # FF2 at the 16x16 level
    assert L.planes_pay(conv, 320, 4096) and L.planes_pay(ff2, 5120, 4096)
    assert L.planes_pay(lin0, 320, 65536) and not L.planes_pay(lin0, 320, 16384)
    assert not L.planes_pay(lin2, 1280, 4096)
    monkeypatch.setattr(L, "PLANES_ALL", True)
    assert L.planes_pay(lin2, 1280, 4096)                     # DDPO_PLANES_ALL=1: every eligible layer
    assert not L.planes_pay(reg(40, 64), 40, 4096)            # never where the plane-fed kernel cannot run (K % 32 != 0)
    monkeypatch.delenv("DDPO_TRAIN_FUSE", raising=False)
    import bench
    assert bench.parse([]).train_fuse == 16 and bench.parse(["--model", "sd21"]).train_fuse == 7
    assert bench.parse(["--train-fuse", "10"]).train_fuse == 10


def test_f16mx_routing_is_a_property_of_the_layer(monkeypatch):
    """Host-side routing of the opt-in f16mx datapath (lib.DATAPATHS): a layer is an f16mx layer iff f16mx weight planes are registered for it
    (pack_weights does that for K >= MX_MIN_K) — whatever the row count; planes_pay() answers with the FORMAT the producer must emit (0 fp32 /
    1 bf16 hi-lo / 2 f16mx), norm_planes() keeps fp32 in front of an f16mx layer on the training forward (its weight gradient reads fp32), and the
    reward towers' fp32-class override maps f16mx to bf16x3."""
    import torch
    from ddpo_amd import lib as L
    monkeypatch.setattr(L, "PLANES", True)
    monkeypatch.setattr(L, "PLANES_ALL", False)
    monkeypatch.setattr(L, "TRAIN_PLANES", True)
    monkeypatch.setattr(L, "DATAPATH", "f16mx")
    entries = {}
    def reg(K, N, mx):
        w = torch.zeros(1)
        entries[w.data_ptr()] = dict(fwd=(None, None, K), bwd=None, K=K, N=N, **({"mx": {}} if mx else {}))
        return w
    monkeypatch.setattr(L, "PACKED", entries)
    conv = reg(2880, 320, True)            # 3x3 conv: long reduction, f16mx planes registered
    lin = reg(320, 320, False)             # 64x64-level projection: stays a bf16x3 layer
    assert L.mx_layer(conv) and not L.mx_layer(lin)
    assert L.planes_pay(conv, 320, 64) == 2 and L.planes_pay(conv, 320, 1 << 20) == 2          # never a function of the rows
    assert L.planes_pay(lin, 320, 65536) == 1 and L.planes_pay(lin, 320, 4096) == 0            # the bf16x3 speed rule (bit-identical either way)
    assert L.norm_planes(conv, 320, 4096, training=False) == 2 and L.norm_planes(conv, 320, 4096, training=True) == 0
    assert L.norm_planes(lin, 320, 65536, training=True) == 1
    monkeypatch.setattr(L, "TRAIN_PLANES", False)
    assert L.norm_planes(lin, 320, 65536, training=True) == 0 and L.norm_planes(lin, 320, 65536, training=False) == 1
    with L.fp32_class_datapath():
        assert L.current_datapath() == "bf16x3" and not L.mx_layer(conv) and L.planes_pay(conv, 320, 64) == 1
    monkeypatch.setattr(L, "DATAPATH", "bf16x3")
    assert not L.mx_layer(conv) and L.planes_pay(conv, 320, 64) == 1                            # registered planes alone do not switch the arithmetic
    with pytest.raises(ValueError):
        L.datapath("fp16")
    with L.datapath("f16mx"):
        assert L.mx_layer(conv)
    import bench
    assert bench.parse(["--datapath", "f16mx"]).datapath == "f16mx"


def test_plane_handover_routing_predicates(monkeypatch):
    """Host-side conditions of round 4's plane hand-over (models/unet.py `_attention` / `_transformer`): the attention hands planes to to_out, and
    FF2 hands planes to proj_out, only where the plane-fed consumer is the faster one (planes_pay == 1: the 64x64 level), only on the 16-bit MFMA
    attention kernels, and FF2 can only emit planes from a buffer-addressed kernel (planes_out_ok)."""
    import torch
    from ddpo_amd import lib as L
    monkeypatch.setattr(L, "PLANES", True)
    monkeypatch.setattr(L, "PLANES_OUT", True)
    monkeypatch.setattr(L, "PLANES_ALL", False)
    entries = {}

    def reg(K, N):
        w = torch.zeros(1)
        entries[w.data_ptr()] = dict(fwd=(None, None, K), bwd=None, K=K, N=N)
        return w
    monkeypatch.setattr(L, "PACKED", entries)
    to_out_64, to_out_32, ff2 = reg(320, 320), reg(640, 640), reg(1280, 320)
    for dp, ok in (("fp32", False), ("bf16", False), ("bf16x3", True), ("f16mx", True)):
        monkeypatch.setattr(L, "DATAPATH", dp)
        assert L.attention_planes_ok(40) == ok and L.attention_planes_ok(80) == ok
        assert not L.attention_planes_ok(160)                                   # d = 160 (SD-2.1's lowest level) stays on the exact-fp32 kernel
    monkeypatch.setattr(L, "DATAPATH", "f16mx")
    assert L.planes_pay(to_out_64, 320, 65536) == 1 and L.planes_pay(to_out_32, 640, 16384) == 0      # 64x64 level only
    assert L.planes_out_ok(ff2, 1280, 65536, 320) and not L.planes_out_ok(ff2, 1280, 65536, 322)      # N % 4
    assert not L.planes_out_ok(torch.zeros(1), 1280, 65536, 320)                                      # weight planes not registered
    monkeypatch.setattr(L, "PLANES_OUT", False)
    assert not L.planes_out_ok(ff2, 1280, 65536, 320)
    # the switches of the model default to on and are plain module attributes (the GPU test toggles them)
    from ddpo_amd.models import unet as U
    assert isinstance(U.ATTN_PLANES, bool) and isinstance(U.H3_PLANES, bool)
