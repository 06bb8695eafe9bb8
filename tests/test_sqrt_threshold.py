"""The claim behind the square-root-free sweeps of csrc/sc2pcr.hip (``sqrt_lt_threshold``, round 6): for a correctly rounded fp32 square
root, ``sqrtf(x) < R`` and ``x < T(R)`` are the same predicate over all x >= 0 (and NaN), with ``T(R) = min {x : sqrtf(x) >= R}`` found
by stepping ulps from ``fl(R * R)``.  numpy's float32 sqrt is correctly rounded like the device's (clang's default for HIP), so the
algorithm is restated here line by line and checked around every threshold - the GPU side is covered by the A/B tests of
tests/test_gpu_sc2pcr.py (same poses and seed-wise fitness with ``sqrtf`` in the loops)."""
import numpy as np

f32 = np.float32


def ulp_step(t, d):
    """the float whose bit pattern is ``d`` above that of the non-negative float ``t`` (inf - 1 = FLT_MAX, FLT_MAX + 1 = inf)"""
    return np.array(np.array(t, f32).view(np.uint32).astype(np.int64) + d, np.uint32).view(f32)[()]


def threshold(R):
    R = f32(R)
    if not (R > 0):
        return R if np.isnan(R) else f32(0)
    if R > f32(1.9e19):
        return f32(np.inf)
    with np.errstate(over="ignore", under="ignore"):
        t = f32(R * R)
        while t > 0 and np.sqrt(t, dtype=f32) >= R:
            t = ulp_step(t, -1)
        while np.sqrt(t, dtype=f32) < R:
            t = ulp_step(t, +1)
    return f32(t)


def test_threshold_is_the_same_predicate_as_the_square_root():
    rng = np.random.default_rng(5)
    Rs = np.concatenate([
        np.array([0.0, -1.0, 0.1, 0.6, 1.2, 0.3, 1.0, 2.0, 1e-30, 1e-23, 1.1e-19, 3e-20, 1e19, 1.84e19, 1.8446e19, 1.9e19, 3e38, np.inf, np.nan], np.float32),
        rng.uniform(0.01, 3.0, 300).astype(np.float32),
        np.exp(rng.uniform(np.log(1e-25), np.log(1.8e19), 300)).astype(np.float32)])
    for R in Rs:
        T = threshold(R)
        # candidates: a window of ulps around T and around R*R, plus special values
        xs = [f32(0), f32(np.inf), f32(np.nan), f32(np.finfo(np.float32).max), f32(np.finfo(np.float32).tiny), f32(1e-45)]
        with np.errstate(over="ignore", under="ignore", invalid="ignore"):
            for c in (T, f32(R * R)):
                if np.isfinite(c) and c >= 0:
                    u = np.array(c, f32).view(np.uint32).astype(np.int64)
                    for d in range(-6, 7):
                        v = u + d
                        if 0 <= v <= 0x7F800000:
                            xs.append(np.array(v, np.uint32).view(f32)[()])
            xs += list(rng.uniform(0, 1, 8).astype(np.float32) * (T if np.isfinite(T) else f32(1e30)) * f32(2))
            for x in xs:
                want = bool(np.sqrt(f32(x), dtype=f32) < R)
                got = bool(f32(x) < T)
                assert want == got, (float(R), float(T), float(x))


def test_cross_length_band_covers_a_one_ulp_square_root():
    """``cross_len_fast`` (the mask kernel's pre-test): with each root replaced by ANY float within one ulp of the correctly rounded one
    (what ``v_sqrt_f32`` promises), a comparison of ``|sa' - sb'|`` with d that clears the band ``2^-20 max(sa', sb') + 1e-18`` gives the
    decision of the exact expression ``|sqrtf(a) - sqrtf(b)| < d`` - checked on lengths from millimetres to 10^4 m, thresholds next to the
    cross length itself, equal and near-equal lengths."""
    rng = np.random.default_rng(9)
    n = 200000
    la = np.exp(rng.uniform(np.log(1e-3), np.log(1e4), n)).astype(np.float32)
    lb = (la * (1 + rng.normal(0, 1, n) * np.exp(rng.uniform(np.log(1e-7), np.log(1.0), n)))).astype(np.float32)
    lb = np.abs(lb)
    lb[::7] = la[::7]
    a, b = (la * la).astype(np.float32), (lb * lb).astype(np.float32)
    sa, sb = np.sqrt(a, dtype=f32), np.sqrt(b, dtype=f32)
    c = np.abs(sa - sb).astype(np.float32)
    # thresholds: the cross length itself moved by a few ulps / a few 1e-6 relative, and unrelated ones
    d = np.where(rng.random(n) < 0.5, c * (1 + rng.integers(-8, 9, n) * f32(2.0 ** -22)).astype(np.float32), rng.choice(np.array([0.1, 0.6, 0.05, 1.2], np.float32), n)).astype(np.float32)
    d = np.maximum(d, f32(1e-6))
    want = c < d
    checked = 0
    for da in (-1, 0, 1):
        for db in (-1, 0, 1):
            sa1 = (sa.view(np.uint32).astype(np.int64) + da).clip(0).astype(np.uint32).view(f32)
            sb1 = (sb.view(np.uint32).astype(np.int64) + db).clip(0).astype(np.uint32).view(f32)
            c1 = np.abs(sa1 - sb1).astype(np.float32)
            band = (np.maximum(sa1, sb1).astype(np.float64) * 2.0 ** -20 + 1e-18).astype(np.float32)      # one fma in the kernel
            e = (c1 - d).astype(np.float32)
            sure_in, sure_out = e < -band, e > band
            assert not (sure_in & ~want).any() and not (sure_out & want).any()
            checked += int(sure_in.sum() + sure_out.sum())
    assert checked > 4.5 * n                              # (half of the thresholds sit within a few ulps of the cross length on purpose; the unrelated ones are all decided)
