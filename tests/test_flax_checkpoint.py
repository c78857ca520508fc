"""The reference's checkpoint file format (flax msgpack of the U-Net param tree): writer / reader round trip, the byte-level
encoding of a leaf (ext type 1 = packb((shape, dtype name, C-order bytes))), chunked leaves, and the name mapping between
this engine's flat parameter names and the Flax module tree.  (flax itself is not installable here: the format is restated
from flax 0.6.9 `serialization.msgpack_serialize`, see ddpo_amd/utils/flax_msgpack.py.)"""
import os

import msgpack
import pytest
import numpy as np

from ddpo_amd.utils import flax_msgpack as FM
from ddpo_amd.models.unet import UNetConfig, unet_param_shapes


def _tiny_params(seed=0):
    rng = np.random.RandomState(seed)
    return {n: rng.randn(*shp).astype(np.float32) for n, shp in unet_param_shapes(UNetConfig.named("tiny")).items()}


def test_round_trip_and_tree_structure(tmp_path):
    flat = _tiny_params()
    path = FM.save_flax_checkpoint(str(tmp_path), flat, step=7)
    assert os.path.basename(path) == "checkpoint_7"
    back = FM.load_flax_checkpoint(path)
    assert set(back) == set(flat)
    assert all(back[n].dtype == np.float32 and np.array_equal(back[n], flat[n]) for n in flat)
    assert FM.load_flax_checkpoint(str(tmp_path)).keys() == flat.keys()          # directory -> latest step
    tree = FM.nest(flat)
    assert {"conv_in", "time_embedding", "down_blocks_0", "mid_block", "up_blocks_3", "conv_norm_out", "conv_out"} <= set(tree)
    assert set(tree["down_blocks_0"]["resnets_0"]["conv1"]) == {"kernel", "bias"}
    assert tree["down_blocks_0"]["attentions_0"]["transformer_blocks_0"]["attn1"]["to_q"]["kernel"].shape == (32, 32)   # Dense (in, out), no bias
    assert tree["conv_in"]["kernel"].shape == (3, 3, 4, 32)                                                            # conv HWIO


def test_leaf_encoding_is_flax_ext_type_1():
    arr = np.arange(12, dtype=np.float32).reshape(3, 4)
    raw = msgpack.unpackb(FM.to_bytes({"a": {"kernel": arr}, "step": np.int32(5)}), raw=False, strict_map_key=False)
    leaf = raw["a"]["kernel"]
    assert isinstance(leaf, msgpack.ExtType) and leaf.code == 1
    shape, dtype_name, buf = msgpack.unpackb(leaf.data, raw=True)
    assert list(shape) == [3, 4] and dtype_name == b"float32" and buf == arr.tobytes("C")
    assert raw["step"].code == 3                                                   # numpy scalar
    back = FM.from_bytes(FM.to_bytes({"a": {"kernel": arr}, "step": np.int32(5)}))
    assert back["step"] == 5 and np.array_equal(back["a"]["kernel"], arr)


def test_chunked_leaf_is_reassembled(monkeypatch):
    monkeypatch.setattr(FM, "_MAX_CHUNK_BYTES", 64)
    arr = np.arange(100, dtype=np.float32).reshape(10, 10)
    data = FM.to_bytes({"w": arr})
    raw = msgpack.unpackb(data, ext_hook=FM._ext_unpack, raw=False, strict_map_key=False)
    assert raw["w"]["__msgpack_chunked_array__"] is True and len(raw["w"]["chunks"]) == 7
    assert np.array_equal(FM.from_bytes(data)["w"], arr)


def test_save_checkpoint_formats_and_load_unet_paths(tmp_path, monkeypatch):
    """save_checkpoint writes the reference's `checkpoint_<step>` (flax msgpack) by default next to the safetensors copy;
    with DDPO_CKPT_FORMATS=flax only that file; load_unet restores from either (host-side logic only: device="cpu")."""
    import torch
    from ddpo_amd.models.unet import ParamStore
    from ddpo_amd.utils.serialization import load_unet, save_checkpoint
    monkeypatch.setenv("DDPO_MODEL_CONFIG", "tiny")
    store = ParamStore(unet_param_shapes(UNetConfig.named("tiny")), "cpu")
    g = torch.Generator().manual_seed(3)
    store.flat.copy_(torch.randn(store.flat.shape, generator=g))
    d1 = str(tmp_path / "both")
    save_checkpoint(d1, store, 4)
    assert sorted(os.listdir(d1)) == ["checkpoint_4", "checkpoint_4.safetensors"]
    monkeypatch.setenv("DDPO_CKPT_FORMATS", "flax")
    d2 = str(tmp_path / "flax_only")
    assert save_checkpoint(d2, store, 9).endswith("checkpoint_9") and os.listdir(d2) == ["checkpoint_9"]
    for loadpath in (d1, d2, "flax:" + d2, "flax:" + os.path.join(d2, "checkpoint_9")):
        _, params = load_unet(loadpath, pretrained_model="none", device="cpu")
        assert all(torch.equal(params["unet"][n], store[n]) for n in store.views), loadpath


def test_resume_bundle_round_trip(tmp_path):
    """The resumable part of a checkpoint (an addition: the reference never reloads a policy-gradient run): rank 0's shared bundle
    + one small file per rank with that rank's host RNG streams and sampling key; `load_resume` picks the latest epoch, finds the
    parameter file (safetensors or flax) and restores RNG streams that continue exactly where they were saved."""
    import random
    import numpy as np
    import torch
    from ddpo_amd.utils.serialization import load_resume, save_rank_resume
    ck = str(tmp_path / "checkpoints")
    os.makedirs(ck)
    random.seed(5); np.random.seed(6)
    random.random(); np.random.permutation(7)
    for epoch in (0, 3):
        torch.save({"epoch": epoch, "opt_count": 4 * (epoch + 1), "mu": torch.zeros(3, dtype=torch.bfloat16), "nu": torch.ones(3),
                    "sample_rng": np.asarray([1, 2], dtype=np.uint32), "tracker": {"a dog": [[0.5, 1.0]]},
                    "mean_rewards": [0.1] * (epoch + 1), "std_rewards": [0.2] * (epoch + 1), "wall": [10.0 * (epoch + 1)]},
                   os.path.join(ck, f"resume_{epoch}.pt"))
    open(os.path.join(ck, "checkpoint_3"), "wb").close()                    # flax-format file only for epoch 3
    open(os.path.join(ck, "checkpoint_0.safetensors"), "wb").close()
    for rank in (0, 1):
        save_rank_resume(ck, 3, rank, {"sample_rng": np.asarray([7 + rank, 9], dtype=np.uint32), "py_random": random.getstate(),
                                       "np_random": np.random.get_state()})
    expect = (random.random(), np.random.permutation(5).tolist())
    random.seed(0); np.random.seed(0)
    rs = load_resume(ck, rank=1)
    assert rs["epoch"] == 3 and rs["opt_count"] == 16 and rs["params_path"].endswith("checkpoint_3")
    assert rs["sample_rng"].tolist() == [8, 9] and rs["mu"].dtype == torch.bfloat16 and rs["wall"] == [40.0]
    random.setstate(rs["py_random"]); np.random.set_state(rs["np_random"])
    assert (random.random(), np.random.permutation(5).tolist()) == expect
    rs0 = load_resume(ck, rank=0, epoch="0")
    assert rs0["epoch"] == 0 and rs0["params_path"].endswith("checkpoint_0.safetensors") and "py_random" not in rs0
    assert rs0["sample_rng"].tolist() == [1, 2]                             # no rank file for epoch 0: the shared bundle's key
    with pytest.raises(FileNotFoundError):
        load_resume(str(tmp_path / "nothing"))
