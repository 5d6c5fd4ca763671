"""CPU tier: the carry-free field (consensus_amd/csrc/p256_fe29.h: 9 signed 29-bit limbs, R = 2^261) and the XYZZ
point layer (p256_pt29.h), compiled by g++ into tests/emul and diffed against Python big integers — including the
worst cases the documented limb / value bounds allow (the i64 column accumulators must not overflow there) and the
exceptional cases of the addition (P == Q, P == -Q, infinity)."""
import ctypes
import random

import pytest

import p256_py as ec
from test_emul_device_algo import emul  # noqa: F401  (fixture: builds tests/emul/libsbv_emul.so)

P = ec.P
R = 1 << 261
RINV = pow(R, -1, P)
M29 = (1 << 29) - 1
I32x9 = ctypes.c_int32 * 9


def L(limbs):
    return I32x9(*limbs)


def val(arr):
    return sum(int(v) << (29 * i) for i, v in enumerate(arr))


def tight(x):
    """canonical limbs of 0 <= x < 2^261"""
    return [(x >> (29 * i)) & M29 for i in range(9)]


def words(x):
    return (ctypes.c_uint32 * 8)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def wval(arr):
    return sum(int(v) << (32 * i) for i, v in enumerate(arr))


def random_loose(rng, spread):
    """a value given by limbs uniformly in (-spread, spread); the value itself is what the limbs say"""
    return [rng.randrange(-spread + 1, spread) for _ in range(8)] + [rng.randrange(-(1 << 24), 1 << 25)]


def check_tight(out, lo, hi):
    v = val(out)
    assert all(0 <= int(out[i]) < (1 << 29) for i in range(8)), list(out)
    assert lo < v < hi, hex(v)
    return v


def test_f29_mul_sqr_match_bigint_within_contract(emul):
    rng = random.Random(29)
    out = I32x9()
    cases = []
    edge = [0, 1, P - 1, P, 2 * P, (1 << 256) - 1, R % P, (1 << 232) - 1, M29, M29 << 29]
    for a in edge:
        for b in edge:
            cases.append((tight(a), tight(b)))
    for _ in range(300):
        cases.append((tight(rng.randrange(2 * P)), tight(rng.randrange(2 * P))))
    # differences / sums of two tight values, the loosest operands the point formulas multiply directly
    for _ in range(300):
        cases.append((random_loose(rng, 1 << 29), random_loose(rng, 1 << 29)))
    # worst-case limb magnitudes of the contract: every limb at +-(2^29 + 2^25) (value-reduced coordinates)
    big = (1 << 29) + (1 << 25)
    for sa in (1, -1):
        for sb in (1, -1):
            cases.append(([sa * big] * 8 + [sa * (1 << 24)], [sb * big] * 8 + [sb * (1 << 24)]))
            cases.append(([sa * big] * 8 + [sa * (1 << 24)], [sb * (1 << 29)] * 8 + [-sb * (1 << 24)]))
    for a, b in cases:
        A, B = val(a), val(b)
        if abs(A) * abs(B) > 16 * P * P:
            continue
        emul.sbve_f29_mul(L(a), L(b), out)
        v = check_tight(out, -P // 2 - 1, 3 * P // 2 + 1)
        assert (v - A * B * RINV) % P == 0, (a, b)
        assert A * B <= v * R < A * B + P * R or (A * B // R) <= v <= (A * B // R) + P + 1
        if abs(A) * abs(A) <= 16 * P * P:
            emul.sbve_f29_sqr(L(a), out)
            v = check_tight(out, -1, 3 * P // 2 + 1)
            assert (v - A * A * RINV) % P == 0, a


def test_f29_hot_path_forms_match_bigint(emul):
    """f29_mulx / f29_sqrx (32-bit-multiplier reduction: value within +-(4.01 p + |AB|/R), exact 29-bit limbs) and the two
    fused reductions of the mixed addition: a*b - c*d and a^2 - v, value-reduced by f29_red_q."""
    rng = random.Random(34)
    out = I32x9()
    big = (1 << 29) + (1 << 26)

    def operand(kind):
        if kind == 0:
            return tight(rng.randrange(5 * P))
        if kind == 1:
            return [-v for v in tight(rng.randrange(5 * P))]
        if kind == 2:
            return random_loose(rng, 1 << 29)
        s_ = rng.choice((1, -1))
        return [s_ * big] * 8 + [s_ * rng.randrange(1 << 26)]

    for trial in range(400):
        a, b, c, d = (operand(rng.randrange(4)) for _ in range(4))
        A, B, C, D = val(a), val(b), val(c), val(d)
        emul.sbve_f29_mulx(L(a), L(b), out)
        v = val(out)
        assert all(0 <= int(out[i]) < (1 << 29) for i in range(8))
        assert (v * R - A * B) % P == 0 and abs(v * R - A * B) <= 4.01 * P * R
        if all(abs(x) < (1 << 30) for x in a):
            emul.sbve_f29_sqrx(L(a), out)
            v = val(out)
            assert (v * R - A * A) % P == 0 and abs(v * R - A * A) <= 4.01 * P * R
            w = [x + y for x, y in zip(tight(rng.randrange(5 * P)), tight(rng.randrange(9 * P)))]       # PPP + 2 Q, loose
            emul.sbve_f29_sqr_sub_val(L(a), L(w), out)
            v = val(out)
            assert (v * R - (A * A - val(w) * R)) % P == 0 and -(1 << 231) < v < (1 << 256) + (1 << 231)
        emul.sbve_f29_mul_sub_mul(L(a), L(b), L(c), L(d), out)
        v = val(out)
        assert (v * R - (A * B - C * D)) % P == 0 and -(1 << 231) < v < (1 << 256) + (1 << 231)
        assert all(-(1 << 27) < int(out[i]) < (1 << 29) + (1 << 27) for i in range(8))


def test_f29_canon_norm_pack_roundtrip(emul):
    rng = random.Random(30)
    out = I32x9()
    w = (ctypes.c_uint32 * 8)()
    for _ in range(500):
        a = random_loose(rng, 1 << 31)
        a[8] = rng.randrange(-(1 << 27), 1 << 27)
        A = val(a)
        assert abs(A) < 16 * P
        emul.sbve_f29_canon(L(a), out)
        assert list(out) == tight(A % P), a
        emul.sbve_f29_pack(out, w)
        assert wval(w) == A % P
        emul.sbve_f29_unpack(w, out)
        assert list(out) == tight(A % P)
        emul.sbve_f29_norm(L(a), out)
        assert val(out) == A and all(-8 < int(out[i]) < (1 << 29) + 8 for i in range(8))
        emul.sbve_f29_norm_red(L(a), out)
        v = val(out)
        assert (v - A) % P == 0 and -(1 << 229) < v < (1 << 256) + (1 << 229)
        assert all(-(1 << 25) - 8 < int(out[i]) < (1 << 29) + (1 << 25) + 8 for i in range(8))
    for x in (0, 1, P - 1, (1 << 256) - 1, 0xFFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000):
        emul.sbve_f29_unpack(words(x), out)
        assert val(out) == x and list(out) == tight(x)


def test_f29_zero_tests_see_every_multiple_of_p(emul):
    rng = random.Random(31)
    for k in range(-16, 17):
        for _ in range(4):
            # k * p spread over loose limbs: add a random limb-wise "zero" (borrow between neighbours)
            limbs = tight(k * P) if k >= 0 else [-v for v in tight(-k * P)]
            for i in range(8):
                d = rng.randrange(-2, 3)                     # keeps every limb inside int32
                limbs[i] += d << 29
                limbs[i + 1] -= d
            assert val(limbs) == k * P
            assert emul.sbve_f29_maybe_zero(L(limbs)) == 1
            assert emul.sbve_f29_is_zero(L(limbs)) == 1
            limbs[rng.randrange(8)] += 1 + rng.randrange(5)
            assert emul.sbve_f29_is_zero(L(limbs)) == 0
    hits = 0
    for _ in range(2000):
        hits += emul.sbve_f29_maybe_zero(L(tight(rng.randrange(P))))
    assert hits == 0          # the filter passes 2^-24 of random values


def test_f29_domain_changes(emul):
    rng = random.Random(32)
    out = I32x9()
    w = (ctypes.c_uint32 * 8)()
    for x in [0, 1, P - 1] + [rng.randrange(P) for _ in range(100)]:
        X32 = x * (1 << 256) % P                      # the 8 x 32 Montgomery form of p256_fe.h
        emul.sbve_f29_from_fe(words(X32), out)
        assert (val(out) - x * R) % P == 0
        emul.sbve_f29_to_fe(out, w)
        assert wval(w) == X32
        emul.sbve_f29_from_plain(words(x), out)
        assert (val(out) - x * R) % P == 0


def entry(pt):
    x, y = pt
    e = (ctypes.c_uint32 * 16)()
    for i in range(8):
        e[i] = ((x * R % P) >> (32 * i)) & 0xFFFFFFFF
        e[8 + i] = ((y * R % P) >> (32 * i)) & 0xFFFFFFFF
    return list(e)


def run_sum(emul, pts, negs):
    n = len(pts)
    ent = (ctypes.c_uint32 * (16 * n))(*[w for p in pts for w in entry(p)])
    ng = (ctypes.c_uint8 * n)(*[1 if v else 0 for v in negs])
    out = (ctypes.c_int32 * 36)()
    inf = emul.sbve_pt29_sum(ent, ng, n, out)
    X, Y, ZZ, ZZZ = (val(out[9 * k:9 * k + 9]) for k in range(4))
    if inf:
        return None, out, inf
    assert ZZ % P != 0
    assert pow(ZZ, 3, P) == pow(ZZZ, 2, P) * R % P    # ZZ = zz R, ZZZ = zzz R with zz^3 == zzz^2
    x = X * pow(ZZ, -1, P) % P
    y = Y * pow(ZZZ, -1, P) % P
    return (x, y), out, inf


def test_pt29_madd_chain_matches_affine_sum_including_exceptional_cases(emul):
    rng = random.Random(33)
    G = (ec.GX, ec.GY)
    base = [ec.pt_mul(rng.randrange(1, ec.N), G) for _ in range(12)]
    for trial in range(40):
        k = rng.randrange(1, 9)
        pts = [rng.choice(base) for _ in range(k)]
        negs = [rng.random() < 0.4 for _ in range(k)]
        shape = trial % 5
        if shape == 1 and k >= 2:
            pts[1], negs[1] = pts[0], negs[0]                     # P + P: the doubling branch
        if shape == 2 and k >= 2:
            pts[1], negs[1] = pts[0], not negs[0]                 # P - P: infinity, then the chain restarts from it
        if shape == 3 and k >= 3:
            s = ec.pt_add(pts[0] if not negs[0] else ec.pt_neg(pts[0]), pts[1] if not negs[1] else ec.pt_neg(pts[1]))
            if s is not None:
                pts[2], negs[2] = s, True                         # (A + B) - (A + B): infinity through a projective accumulator
        if shape == 4 and k >= 3:
            s = ec.pt_add(pts[0] if not negs[0] else ec.pt_neg(pts[0]), pts[1] if not negs[1] else ec.pt_neg(pts[1]))
            if s is not None:
                pts[2], negs[2] = s, False                        # (A + B) + (A + B): doubling with ZZ != 1
        want = None
        for p_, n_ in zip(pts, negs):
            want = ec.pt_add(want, ec.pt_neg(p_) if n_ else p_)
        got, out, inf = run_sum(emul, pts, negs)
        assert got == want, (trial, shape)
        if want is not None:
            r = want[0] % ec.N
            assert emul.sbve_pt29_rx_matches(out, inf, words(r)) == 1
            assert emul.sbve_pt29_rx_matches(out, inf, words((r + 1) % ec.N)) == 0
        else:
            assert emul.sbve_pt29_rx_matches(out, inf, words(1)) == 0


def test_quad_cooperative_doubling_chain_matches_bigint(emul):
    """p256_keytab29.h: the table builder's doubling chain — four lanes per key, modified Jacobian coordinates, one product
    level per lane (keychain29_dbl) — against 2^n * P from Python big integers; the emulated lanes must agree with one
    another and the carried T must stay -3 Z^4.  Runs under SBV_F29_CHECK: every product asserts its operand bounds."""
    rng = random.Random(77)
    G = (ec.GX, ec.GY)
    emul.sbve_keychain_dbl.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    pts = [G, ec.pt_mul(ec.N - 1, G), ec.pt_mul(2, G)] + [ec.pt_mul(rng.randrange(1, ec.N), G) for _ in range(10)]
    ox, oy = (ctypes.c_uint32 * 8)(), (ctypes.c_uint32 * 8)()
    for i, pt in enumerate(pts):
        for n in (0, 1, 2, 3, 4, 8, 33, 256) if i < 4 else (rng.randrange(1, 64), 128):
            assert emul.sbve_keychain_dbl(words(pt[0]), words(pt[1]), n, ox, oy) == 1, (i, n)
            assert (wval(ox), wval(oy)) == ec.pt_mul(pow(2, n, ec.N), pt), (i, n)


def test_key_table_of_the_key_that_wrapped_a_limb(emul):
    """Regression (round 3, found by the GPU tier): for this key the giants lane of window 8 doubles a point whose x^2 has a
    limb within 2^19 of 2^29 while the matching limb of a4 = -3 Z^4 sits above 2^29 — 3 x^2 + a4 formed without an
    intermediate f29_norm wrapped the i32 and 97 of the key's 4097 table entries came out wrong.  The whole table
    (k_keytab29_chain -> rows -> fill as the grouped step runs them) against k * 2^(8j) * Q from Python big integers."""
    emul.sbve_keytab_build.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
    key = bytes.fromhex("682f6f1f118445cb0a09ef3dc8e5c202fb3d60460e82f513d07bd5fece567f52"
                        "cec0e52c50735a5480607398a6880d494a83013fa269b1442a36466fe2fcfee1")
    q = (int.from_bytes(key[:32], "big"), int.from_bytes(key[32:], "big"))
    tab = (ctypes.c_uint32 * (33 * 128 * 16))()
    # rows (babies 1..8, giants 16..128) + the symmetric fill (both sides of every giant from one inverse), in 1..4 chunks of windows
    for chunks in (1, 2, 3, 4):
        for k in range(len(tab)):
            tab[k] = 0xA5A5A5A5
        assert emul.sbve_keytab_build(key, chunks, tab) == 1
        for j in (0, 7, 8, 9, 31, 32):
            base = ec.pt_mul(pow(2, 8 * j, ec.N), q)
            every = j == 8 and chunks == 2                     # one window in full: every one of its 128 entries
            for m_ in ((1,) if j == 32 else (range(1, 129) if every else (1, 2, 3, 7, 11, 13, 14, 15, 16, 17, 31, 32, 33, 48, 80, 100, 112, 113, 127, 128))):
                e = tab[(j * 128 + m_ - 1) * 16:(j * 128 + m_) * 16]
                want = ec.pt_mul(m_, base)
                assert (wval(e[:8]), wval(e[8:])) == (want[0] * R % P, want[1] * R % P), (chunks, j, m_)
    off = bytearray(key)
    off[63] ^= 1
    assert emul.sbve_keytab_build(bytes(off), 2, tab) == 0          # pointFromAffine refuses it (key29_load)


def test_s29_scalar_field_matches_bigint(emul):
    """p256_sc29.h: Montgomery multiplication mod the group order N with R = 2^261, canonical form, inversion."""
    N = ec.N
    rng = random.Random(35)
    out = I32x9()
    rinv = pow(R, -1, N)
    vals = [0, 1, N - 1, N, (1 << 256) - 1, R % N, (R * R) % N] + [rng.randrange(1 << 256) for _ in range(200)]
    for i, a in enumerate(vals):
        for b in (vals[(7 * i + 3) % len(vals)], vals[(13 * i + 5) % len(vals)]):
            emul.sbve_s29_mul(L(tight(a)), L(tight(b)), out)
            v = val(out)
            assert all(0 <= int(out[k]) < (1 << 29) for k in range(8))
            assert (v - a * b * rinv) % N == 0 and a * b // R <= v <= a * b // R + N + 1
            emul.sbve_s29_canon(out, out)
            assert val(out) == a * b * rinv % N
    for a in [1, 2, N - 1] + [rng.randrange(1, N) for _ in range(40)]:
        aM = a * R % N
        emul.sbve_s29_inv(L(tight(aM)), out)
        assert (val(out) - pow(a, -1, N) * R) % N == 0
    emul.sbve_s29_inv(L(tight(0)), out)
    assert val(out) % N == 0
