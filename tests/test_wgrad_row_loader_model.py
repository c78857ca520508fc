"""CPU model of the weight-gradient row loader's index arithmetic (ddpo_amd/csrc/gemm_bf16.hip, `ROWL`).

The kernel keeps the image row / column of a k-tile's first pixel in scalars, gives every thread constant byte offsets relative to the tile
and advances one scalar offset per tile; a tap outside the image becomes an out-of-range buffer offset.  This test restates exactly that
arithmetic (eligibility rule, guard in front of the tensor, per-element (cy, cx), the scalar (oy, ox) update) in Python and checks it against
the plain per-pixel formula — for every U-Net latent geometry incl. the ones the GPU tests do not reach (SD-2.1's 96 / 48 / 24 / 12),
every tap of a 3x3 / 1x1 stride-1 convolution, every split start.  It guards the logic, not the hardware: the GPU suite compares results."""
import itertools

import pytest

BK = 32


def eligible(B, H, W, ks):
    M = B * H * W
    if M % 32:
        return False
    if ks == 0:
        return True
    return (W % 32 == 0 or 32 % W == 0) and (H * W) % 32 == 0


def row_loader_elements(B, H, W, ks, ld, m_begin, n_tiles, dky, dkx, ci):
    """(tile, eoff) -> (fetched?, element offset) as the kernel computes them."""
    conv = ks > 0
    tap_off = (dky * W + dkx) * ld + ci if conv else ci
    guard = (W + 1) * ld if conv else 0
    rem = m_begin % (H * W) if conv else 0
    oy, ox = (rem // W, rem % W) if conv else (0, 0)
    out = {}
    for kt in range(n_tiles):
        so = kt * BK * ld
        for eoff in range(BK):
            dy_e, x_e = (eoff // W, eoff % W) if (conv and W < BK) else (0, eoff)
            va = (m_begin + eoff) * ld + tap_off + guard
            assert va >= 0                                              # the guard makes every tile-0 offset non-negative
            ok = True
            if conv:
                ok = 0 <= oy + dy_e + dky < H and 0 <= ox + x_e + dkx < W
            out[(kt, eoff)] = (ok, va + so - guard)
        if conv:
            if W >= BK:
                ox += BK
                if ox >= W:
                    ox = 0
                    oy = 0 if oy + 1 >= H else oy + 1
            else:
                oy += BK // W
                if oy >= H:
                    oy -= H
    return out


def per_pixel_elements(B, H, W, ks, ld, m_begin, n_tiles, dky, dkx, ci):
    out = {}
    for kt in range(n_tiles):
        for eoff in range(BK):
            m = m_begin + kt * BK + eoff
            if ks == 0:
                out[(kt, eoff)] = (True, m * ld + ci)
                continue
            y, x = (m // W) % H, m % W
            ok = 0 <= y + dky < H and 0 <= x + dkx < W
            out[(kt, eoff)] = (ok, m * ld + (dky * W + dkx) * ld + ci)
    return out


@pytest.mark.parametrize("hw", [64, 32, 16, 8, 96, 128])
@pytest.mark.parametrize("ks", [3, 1])
def test_row_loader_matches_per_pixel_addressing(hw, ks):
    B, ld = 3, 64
    assert eligible(B, hw, hw, ks)
    M = B * hw * hw
    pad = ks // 2
    for dky, dkx in itertools.product(range(-pad, pad + 1), repeat=2):
        for m_begin in sorted({0, 32, (M // 64) * 32, M - 64}):
            n_tiles = min(6, (M - m_begin) // BK)
            a = row_loader_elements(B, hw, hw, ks, ld, m_begin, n_tiles, dky, dkx, ci=8)
            b = per_pixel_elements(B, hw, hw, ks, ld, m_begin, n_tiles, dky, dkx, ci=8)
            for key in b:
                assert a[key][0] == b[key][0], (hw, ks, dky, dkx, m_begin, key)
                if b[key][0]:
                    assert a[key][1] == b[key][1], (hw, ks, dky, dkx, m_begin, key)


def test_row_loader_eligibility_rule():
    assert eligible(16, 64, 64, 3) and eligible(64, 8, 8, 3) and eligible(4, 96, 96, 3) and eligible(777 * 32, 1, 1, 0)
    assert not eligible(4, 48, 48, 3) and not eligible(4, 24, 24, 3) and not eligible(3, 12, 20, 1)      # fall back to the per-pixel loader
    assert not eligible(1, 777, 1, 0)                                                                     # dense, M % 32 != 0
