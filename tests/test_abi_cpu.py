"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/ddpo_hip.h declares,
struct mirrors agree, the host-side Threefry matches the oracle, and the product's parameter inventory equals the
oracle's (names, Flax layouts, counts).  No device compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

from ddpo_amd import lib as L
from oracle import prng as OP, unet as OU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ddpo_hip.h")).read()
    declared = set(re.findall(r"\b(ddpo_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ddpo_hip.h but not exported"
    assert declared == set(L.EXPORTED_SYMBOLS)
    assert L.load().ddpo_abi_version() == L.ABI_VERSION == 14


def test_struct_mirrors():
    lib = L.load()
    assert lib.ddpo_sizeof_gemm_desc() == ctypes.sizeof(L.GemmDesc)
    assert lib.ddpo_sizeof_ddim_consts() == ctypes.sizeof(L.DdimConsts)


@pytest.mark.parametrize("n", [0, 1, 2, 5, 16, 101])
def test_host_threefry_matches_oracle(n):
    key = OP.PRNGKey(99)
    assert np.array_equal(L.threefry_bits_host(key, n), OP.random_bits(key, n))


def test_host_key_tree_matches_oracle():
    from ddpo_amd.utils import prng
    rng = prng.PRNGKey(3)
    _, sample_rng = prng.split(rng)
    sample_rng, seed = prng.split(sample_rng)
    seeds = prng.split(seed, 8)
    assert np.array_equal(seeds, OP.sample_key_tree(3, 8, 1)[0])


def test_bad_host_arguments_return_einval():
    assert L.load().ddpo_threefry_bits_host(0, 0, 4, None) == -1
    assert L.load().ddpo_gemm_conv_fwd(None, None) == -1


def test_param_inventory_matches_oracle():
    from ddpo_amd.models.unet import UNetConfig, unet_param_shapes
    from ddpo_amd.models.vae import VAEConfig, vae_decoder_param_shapes
    for name, ocfg in (("sd15", OU.SD15), ("sd21", OU.SD21), ("tiny", OU.TINY)):
        mine = unet_param_shapes(UNetConfig.named(name))
        ref = OU.unet_param_shapes(ocfg)
        assert dict(mine) == dict(ref), name
    assert dict(vae_decoder_param_shapes(VAEConfig.named("sd"))) == dict(OU.vae_decoder_param_shapes(OU.VAE_SD))
    assert dict(vae_decoder_param_shapes(VAEConfig.named("tiny"))) == dict(OU.vae_decoder_param_shapes(OU.VAE_TINY))
    assert sum(int(np.prod(s)) for s in unet_param_shapes(UNetConfig.named("sd15")).values()) == 859520964


def test_gemm_desc_field_order_matches_header():
    """The ctypes mirror must list the fields of ddpo_gemm_desc in the header's order (equal sizes alone would not catch
    two swapped ints)."""
    hdr = open(os.path.join(ROOT, "include", "ddpo_hip.h")).read()
    end = hdr.index("} ddpo_gemm_desc;")
    body = hdr[hdr.rindex("typedef struct", 0, end):end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        decl = re.sub(r"^(const\s+)?(float|int|int32_t|int64_t|size_t|void|uint16_t|uint8_t)\s*\*?\s*", "", stmt)
        names += [n.strip().lstrip("*") for n in decl.split(",")]
    assert names == [f[0] for f in L.GemmDesc._fields_]


def test_planes_host_logic(monkeypatch):
    """Eligibility rule of the plane-fed GEMM (mirror of buf_path_ok in csrc/gemm_bf16.hip) and the Planes container; no launch."""
    import torch
    monkeypatch.setattr(L, "PLANES", True)
    monkeypatch.setattr(L, "DATAPATH", "bf16x3")
    w = torch.zeros(64, 64)
    monkeypatch.setitem(L.PACKED, w.data_ptr(), dict(K=64, N=64, fwd=(None, None, 64), bwd=None))
    assert L.planes_ok(w, 64, 1000)
    assert not L.planes_ok(w, 64, 1 << 24)            # 2^24 rows x 64 channels x 4 B >= 2^31: stays on the pointer-addressed kernel
    assert not L.planes_ok(w, 40, 10)                 # k-tiles of 32 must not straddle a tap
    assert not L.planes_ok(torch.zeros(8, 8), 32, 10)     # weight planes not registered
    monkeypatch.setattr(L, "DATAPATH", "fp32")
    assert not L.planes_ok(w, 64, 1000)
    monkeypatch.setattr(L, "DATAPATH", "bf16x3")
    monkeypatch.setattr(L, "PLANES", False)
    assert not L.planes_ok(w, 64, 1000)
    pl = L.Planes(4, 8, "cpu")
    pl.hi.fill_(0x3F80)                                # bf16 1.0
    pl.lo.fill_(0x3B80)                                # bf16 2^-8
    assert pl.shape == (4, 8) and torch.equal(pl.float(), torch.full((4, 8), 1.0 + 2.0 ** -8))
