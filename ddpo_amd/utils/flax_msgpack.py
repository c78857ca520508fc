"""Read / write the reference's checkpoint file format: `checkpoints/checkpoint_<epoch>` as produced by
`flax.training.checkpoints.save_checkpoint(..., target=unreplicate(state.params))`
(/root/reference/pipeline/policy_gradient.py:457-464) and consumed by `restore_checkpoint`
(/root/reference/ddpo/utils/serialization.py:357-362, the `flax:` load path).

flax is not installable here, so this is a restatement of `flax.serialization.msgpack_serialize / msgpack_restore`
(flax 0.6.9) on top of the plain `msgpack` package:
  * the param pytree is a nested dict with string keys; leaves are numpy arrays;
  * an array leaf is `msgpack.ExtType(1, msgpack.packb((shape, dtype.name, raw C-order bytes), use_bin_type=True))`;
    numpy scalars use ext code 3 with the same payload, native complex numbers ext code 2;
  * leaves above 2**30 bytes are stored as {"__msgpack_chunked_array__": True, "shape": ..., "chunks": {"0": ..., ...}}
    (never reached by a U-Net parameter; reading it is supported);
  * the outer object is `msgpack.packb(tree, default=<ext packer>, strict_types=True)`.
Parameter names map one-to-one: this engine's flat name "down_blocks_0.resnets_1.conv1.kernel" is the Flax path
("down_blocks_0", "resnets_1", "conv1", "kernel"), layouts are Flax's (conv HWIO, dense (in, out)).
"""
import os

import msgpack
import numpy as np

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_MAX_CHUNK_BYTES = 2 ** 30


def _ndarray_payload(arr):
    arr = np.asarray(arr)
    if arr.dtype.hasobject:
        raise ValueError("object arrays cannot be serialised")
    return msgpack.packb((arr.shape, arr.dtype.name, arr.tobytes("C")), use_bin_type=True)


def _ext_pack(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_payload(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _ndarray_payload(np.asarray(x)))
    if isinstance(x, complex):
        return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((x.real, x.imag)))
    raise TypeError(f"cannot serialise {type(x)}")


def _ext_unpack(code, data):
    if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
        shape, dtype_name, buf = msgpack.unpackb(data, raw=True)
        dtype_name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
        if dtype_name == "bfloat16":
            # numpy has no bfloat16 (flax relies on ml_dtypes): a bf16 value is the upper half of the fp32 with the same bits, so the
            # leaf is widened exactly to float32 (what every consumer here wants: parameters live in fp32 buffers)
            arr = (np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(shape)
        else:
            arr = np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()      # own, writable memory
        return arr[()] if code == _EXT_NPSCALAR else arr
    if code == _EXT_COMPLEX:
        re, im = msgpack.unpackb(data)
        return complex(re, im)
    return msgpack.ExtType(code, data)


def _chunk(arr):
    flat = arr.reshape(-1)
    per = max(1, _MAX_CHUNK_BYTES // arr.dtype.itemsize)
    # flax writes both tuples through `_tuple_to_dict`: shape = {"0": n, "1": m}, chunks = {"0": ..., "1": ...}
    return {"__msgpack_chunked_array__": True, "shape": {str(i): int(n) for i, n in enumerate(arr.shape)},
            "chunks": {str(i): flat[o:o + per] for i, o in enumerate(range(0, flat.size, per))}}


def _prepare(tree):
    if isinstance(tree, dict):
        return {str(k): _prepare(v) for k, v in tree.items()}
    if isinstance(tree, (np.generic, complex)):
        return tree                                    # numpy scalars / native complex keep their own ext codes
    arr = np.asarray(tree)
    if not arr.flags["C_CONTIGUOUS"]:
        arr = np.array(arr, order="C")                 # (np.ascontiguousarray would promote 0-d to 1-d)
    return _chunk(arr) if arr.nbytes > _MAX_CHUNK_BYTES else arr


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get("__msgpack_chunked_array__"):
            chunks, shape = tree["chunks"], tree["shape"]
            if isinstance(shape, dict):               # flax's form; a plain list (files written before round 2) is accepted too
                shape = tuple(int(shape[str(i)]) for i in range(len(shape)))
            return np.concatenate([chunks[str(i)] for i in range(len(chunks))]).reshape(tuple(shape))
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def to_bytes(tree):
    """flax.serialization.to_bytes of a nested dict of arrays."""
    return msgpack.packb(_prepare(tree), default=_ext_pack, strict_types=True)


def from_bytes(data):
    """flax.serialization.msgpack_restore: nested dict of numpy arrays."""
    return _unchunk(msgpack.unpackb(data, ext_hook=_ext_unpack, raw=False, strict_map_key=False))


def nest(flat, sep="."):
    tree = {}
    for name, v in flat.items():
        node = tree
        parts = name.split(sep)
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return tree


def flatten(tree, sep=".", prefix=""):
    out = {}
    for k, v in tree.items():
        name = f"{prefix}{sep}{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten(v, sep, name))
        else:
            out[name] = v
    return out


def save_flax_checkpoint(ckpt_dir, flat_params, step, prefix="checkpoint_"):
    """Write `<ckpt_dir>/<prefix><step>` the way flax's save_checkpoint names and encodes it (atomic rename of a tmp file)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(to_bytes(nest({n: np.asarray(v) for n, v in flat_params.items()})))
    os.replace(tmp, path)
    return path


def load_flax_checkpoint(path):
    """Return {flat_name: numpy array} from a flax msgpack checkpoint file (or the latest `checkpoint_<n>` in a directory)."""
    if os.path.isdir(path):
        steps = sorted(int(f[len("checkpoint_"):]) for f in os.listdir(path) if f.startswith("checkpoint_") and f[len("checkpoint_"):].isdigit())
        if not steps:
            raise FileNotFoundError(f"no checkpoint_<step> file in {path}")
        path = os.path.join(path, f"checkpoint_{steps[-1]}")
    with open(path, "rb") as f:
        return flatten(from_bytes(f.read()))
