"""The level plan of the banded shift solve (csrc/shiftsolve.hip plan_level, through mispec_symshift_level_plan — host arithmetic, no
GPU): which path a matrix takes, that the chain of Schur levels (half-bandwidth 2b - 1 per level) always reaches its dense last
level before the band outgrows the chunk kernels (64), and that BASELINE's C5 keeps the plan its measurements were taken with."""
import ctypes as C

import pytest

import spectra_amd as sa
from spectra_amd import _capi


def plan(n, b):
    arrays = [(C.c_int64 * 16)() for _ in range(4)]
    k = sa.lib().mispec_symshift_level_plan(n, b, 16, *arrays)
    return k, [tuple(int(a[i]) for a in arrays) for i in range(max(k, 0))]  # (rows, half-bandwidth, chunk rows, chunks)


def test_config5_plan_is_unchanged():
    k, levels = plan(2_000_000, 3)
    assert k == 3 and levels == [(2_000_000, 3, 128, 15625), (46872, 5, 128, 366), (1825, 9, 1825, 1)]


@pytest.mark.parametrize("b", [1, 2, 3, 5, 8, 9, 12, 16, 24, 31, 32, 33, 40, 48, 63, 64])
@pytest.mark.parametrize("n", [4097, 5000, 20_000, 123_457, 1_000_000, 10_000_000, 100_000_000])
def test_every_supported_band_reaches_a_dense_last_level(n, b):
    k, levels = plan(n, b)
    assert 1 <= k <= 8, (n, b, k)
    assert levels[0][:2] == (n, b)
    for (N, bw, L, P), nxt in zip(levels, levels[1:] + [None]):
        assert P >= 1 and L >= 1 and P * L <= N + L
        if nxt is None:
            assert P == 1                      # the dense level: any width, bounded size
            assert N <= max(2048, 8 * bw) or k == 1
        else:
            assert P > 1 and bw <= 64          # a partitioned level: within the chunk kernels
            assert nxt[0] == (P - 1) * bw and nxt[1] == min(2 * bw - 1, nxt[0] - 1)
            assert L >= 4 * bw
    assert levels[-1][0] <= 2048 or k == 1


def test_paths():
    assert plan(4096, 20)[0] == 0                       # band wider than 8 inside the dense limit: dense inverse
    assert plan(3000, 8)[0] >= 1                        # narrow band: banded path at any n
    assert plan(5000, 65)[0] == _capi.MISPEC_EINVAL     # wider than 64 beyond the dense limit: unsupported
    assert plan(100, 99)[0] == 0
    assert plan(1, 0)[0] == 1
