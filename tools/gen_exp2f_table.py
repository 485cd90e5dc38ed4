#!/usr/bin/env python3
"""Regenerates S2R_EXP2F_TABLE_INIT (synth2_amd/csrc/s2r_math.h): entry i = bits(RN53(2^(i/32))) - (i << 47),
the table of glibc's expf/exp2f/powf (sysdeps/ieee754/flt-32/math_config.h, EXP2F_TABLE_BITS = 5).

2^(i/32) is irrational for 0 < i < 32, so its correctly rounded binary64 value is decided with integers only:
the 53-bit significand m of x = 2^(i/32) in [1, 2) is the m with  (m - 1/2)^32 < 2^(52*32 + i) <= (m + 1/2)^32,
found by bisection on exact big integers.  `python tools/gen_exp2f_table.py` prints the initializer;
tests/test_transcendentals_pinned.py::test_exp2f_table_is_the_correctly_rounded_one diffs it against the header."""
N = 32


def entry(i):
    target = 1 << (52 * N + i)               # (2^52 * 2^(i/32))^32
    lo, hi = 1 << 52, 1 << 53                # significand m with (2m - 1)^32 < 2^32 * target <= (2m + 1)^32
    t2 = target << N                         # compare (2m +- 1)^32 against 2^32 * target
    while lo < hi:
        mid = (lo + hi) // 2
        if (2 * mid + 1) ** N < t2:          # m + 1/2 still below the true value: m is too small
            lo = mid + 1
        else:
            hi = mid
    m = lo
    assert (2 * m - 1) ** N < t2 <= (2 * m + 1) ** N or i == 0
    bits = (1023 << 52) | (m - (1 << 52))    # x in [1, 2): exponent 0
    return (bits - (i << 47)) & 0xFFFFFFFFFFFFFFFF


def table():
    return [entry(i) for i in range(N)]


if __name__ == "__main__":
    t = table()
    print("#define S2R_EXP2F_TABLE_INIT { \\")
    for r in range(0, N, 4):
        print("    " + ", ".join("0x%016xull" % v for v in t[r:r + 4]) + (", \\" if r + 4 < N else " }"))
