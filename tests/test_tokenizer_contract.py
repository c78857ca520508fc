"""Token-id contract of the text front end on a REAL `transformers.CLIPTokenizer`, given a `tokenizer/` directory.

No CLIP vocabulary is reachable offline, so the directory is synthesised here: CLIP's byte-level BPE alphabet (every byte symbol, bare
and with the end-of-word marker), a handful of merges, and the two special tokens.  What is pinned is everything that does not depend
on WHICH merges the real vocabulary holds: `prepare_inputs` (reference ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:148-161)
pads to `model_max_length` = 77 with the end-of-text id, truncates long prompts to 77 keeping <bos> ... <eos>, returns numpy ids, and
`make_uncond_text` (reference ddpo/datasets/bucket.py:66-73) is the padded empty prompt.  Where /root/reference exists (this container,
not the GPU box) the reference's own `prepare_inputs` body is lifted with `ast` and run on the same tokenizer: ids must be equal."""
import ast
import json
import os

import numpy as np
import pytest

from ddpo_amd.diffusers_patch.pipeline_stable_diffusion import StableDiffusionPipeline
from ddpo_amd.models.text import load_tokenizer, make_uncond_text

REF = "/root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py"


@pytest.fixture(scope="module")
def tok_dir(tmp_path_factory):
    root = tmp_path_factory.mktemp("sd")
    d = root / "tokenizer"
    d.mkdir()
    # byte-level BPE alphabet: printable bytes map to themselves, the other bytes to code points from 256 upwards
    keep = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    chars = [chr(b) for b in keep] + [chr(256 + i) for i in range(256 - len(keep))]
    vocab = chars + [c + "</w>" for c in chars]
    merges = [("a", "t</w>"), ("c", "at</w>"), ("d", "o"), ("do", "g</w>"), ("r", "i"), ("ri", "d"), ("rid", "i"), ("ridi", "n"),
              ("ridin", "g</w>"), ("b", "i"), ("bi", "k"), ("bik", "e</w>")]
    vocab += ["".join(m) for m in merges]
    vocab += ["<|startoftext|>", "<|endoftext|>"]
    (d / "vocab.json").write_text(json.dumps({t: i for i, t in enumerate(vocab)}))
    (d / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    (d / "tokenizer_config.json").write_text(json.dumps({"model_max_length": 77, "tokenizer_class": "CLIPTokenizer"}))
    (d / "special_tokens_map.json").write_text(json.dumps({"bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>",
                                                           "unk_token": "<|endoftext|>", "pad_token": "<|endoftext|>"}))
    return str(root)


PROMPTS = ["a cat riding a bike", "", "dog", "a " + "cat " * 100, "Ünïcode dog!"]


def test_prepare_inputs_pads_and_truncates_like_the_reference(tok_dir):
    tok = load_tokenizer(tok_dir)
    assert type(tok).__name__ == "CLIPTokenizer" and tok.synthetic is False and tok.model_max_length == 77
    pipe = StableDiffusionPipeline(None, None, None, tokenizer=tok)
    ids = pipe.prepare_inputs(PROMPTS)
    assert isinstance(ids, np.ndarray) and ids.shape == (len(PROMPTS), 77) and np.issubdtype(ids.dtype, np.integer)
    bos, eos = tok.bos_token_id, tok.eos_token_id
    assert (ids[:, 0] == bos).all()
    # the empty prompt: <bos> <eos> then padding with the pad id (= <eos> in CLIP's tokenizer config)
    assert ids[1, 1] == eos and (ids[1, 2:] == tok.pad_token_id).all() and tok.pad_token_id == eos
    # a short prompt: its tokens, one <eos>, padding; "cat" is ONE token through the merges, "a" the merged "a</w>"
    body = [t for t in ids[0, 1:] if t != eos]
    assert len(body) == 5 and tok.convert_ids_to_tokens(body) == ["a</w>", "cat</w>", "riding</w>", "a</w>", "bike</w>"]
    assert ids[0, 1 + len(body)] == eos
    # a 101-word prompt is truncated to 77 ids and still ends with <eos>
    assert ids[3, 76] == eos and (ids[3, 1:76] != eos).all()
    # single string -> (1, 77); a non-str / list prompt is refused with the reference's message
    assert pipe.prepare_inputs("dog").shape == (1, 77) and np.array_equal(pipe.prepare_inputs("dog")[0], ids[2])
    with pytest.raises(ValueError, match="has to be of type `str` or `list`"):
        pipe.prepare_inputs(("dog",))
    un = make_uncond_text(tok, 3)
    assert un.shape == (3, 77) and all(np.array_equal(r, ids[1]) for r in un)


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is only present in the build container")
def test_prepare_inputs_equals_the_reference_method_on_the_same_tokenizer(tok_dir):
    tok = load_tokenizer(tok_dir)
    tree = ast.parse(open(REF).read())
    fn = next(n for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "FlaxStableDiffusionPipeline"
              for n in c.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_inputs")
    ns = {"Union": __import__("typing").Union, "List": __import__("typing").List}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)

    class Self:
        tokenizer = tok
    ref_ids = ns["prepare_inputs"](Self(), PROMPTS)
    ours = StableDiffusionPipeline(None, None, None, tokenizer=tok).prepare_inputs(PROMPTS)
    assert ref_ids.dtype == ours.dtype and np.array_equal(ref_ids, ours)


def test_mu_decay_env_is_honoured_when_the_flag_is_not_given(monkeypatch):
    """config/base.py used to default `mu_decay_in_bf16` to True, so DDPO_MU_DECAY_IN_BF16=0 never took effect."""
    import importlib
    base = importlib.import_module("config.base")
    assert base.base["pg"]["mu_decay_in_bf16"] is None
    from pipeline.policy_gradient import _flag
    for env, want in (("0", False), ("1", True), (None, True)):
        if env is None:
            monkeypatch.delenv("DDPO_MU_DECAY_IN_BF16", raising=False)
        else:
            monkeypatch.setenv("DDPO_MU_DECAY_IN_BF16", env)
        assert _flag(None, os.environ.get("DDPO_MU_DECAY_IN_BF16", "1") != "0") is want
    assert _flag("False", True) is False and _flag(True, False) is True
