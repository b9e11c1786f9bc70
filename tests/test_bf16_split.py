"""The backward's three-channel contraction (gsr_blend_bwd.hip, GSR_BWD_BF16) feeds the bf16 matrix instruction with
hi / mid / lo parts of f32 values and claims the split is EXACT: truncate to the upper 16 bits, subtract, twice; the
second remainder has at most eight significant bits.  This restates the device arithmetic in numpy and checks the claim."""
import numpy as np


def upper(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split(x):
    hi = upper(x)
    y = x - hi          # exact: both share the exponent, y has <= 16 significant bits
    mid = upper(y)
    z = y - mid         # exact, <= 8 significant bits
    lo = upper(z)
    return hi, mid, lo, z


def test_three_way_split_is_exact():
    rng = np.random.default_rng(0)
    mant = rng.integers(0, 1 << 23, 400_000, dtype=np.uint32)
    expo = rng.integers(30, 220, 400_000, dtype=np.uint32)      # normal numbers with room for the remainders
    sign = rng.integers(0, 2, 400_000, dtype=np.uint32)
    x = ((sign << 31) | (expo << 23) | mant).view(np.float32)
    x = np.concatenate([x, np.array([0.0, -0.0, 1.0, -1.0, 0.99, 1 / 255, 3.5, 12.25, 1e-30, 3e38], np.float32)])
    hi, mid, lo, z = split(x)
    assert np.array_equal(lo, z), "the second remainder must fit the upper 16 bits"
    s = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))
    for part in (hi, mid, lo):   # each part is a bf16 number: low half of the word is zero
        assert not np.any(part.view(np.uint32) & np.uint32(0xFFFF))


def test_monomials_are_bf16_numbers():
    c = np.arange(8, dtype=np.float32) - 3.5
    xr, yr = np.meshgrid(c, c)
    for m in (np.ones_like(xr), xr, yr, xr * xr, xr * yr, yr * yr):
        assert np.array_equal(upper(m.astype(np.float32)), m.astype(np.float32))
