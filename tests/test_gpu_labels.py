"""GPU parity: label-generation kernels (eyoc_knn2, eyoc_lowe_topk, eyoc_pair_filter) and their Python mirrors of
lib/trainer.py:1025-1151,1195-1218 vs the CPU oracle (oracle/labels.py; parity unpinned, see its header).
Index outputs and distances are bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def unit(rng, n, c=32):
    f = rng.normal(size=(n, c)).astype(np.float32)
    return f / np.linalg.norm(f, axis=1, keepdims=True).astype(np.float32)


@pytest.mark.parametrize("na,nb,c", [(700, 900, 32), (65, 1, 32), (1, 64, 16), (300, 2, 4), (1000, 1033, 4), (129, 257, 64)])
def test_knn2_matches_oracle_bit_for_bit(na, nb, c):
    import eyoc_amd
    from oracle import labels as ol
    rng = np.random.default_rng(na * 7 + nb)
    A, B = unit(rng, na, c), unit(rng, nb, c)
    B[nb // 2] = B[0]                                   # an exact duplicate target: tie on the nearest distance
    if na > 10:
        A[3] = B[0]                                     # distance exactly zero, runner-up also zero
    idx, d1, d2 = eyoc_amd.knn2_segmented(torch.from_numpy(A), torch.from_numpy(B), [0, na], [0, nb])
    ri, r1, r2 = ol.knn2(A, B)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(d1.cpu().numpy(), r1)
    np.testing.assert_array_equal(d2.cpu().numpy(), r2)


def test_knn2_segments_and_knn1_agree():
    import eyoc_amd
    from eyoc_amd.eval import knn1_segmented
    from oracle import labels as ol
    rng = np.random.default_rng(5)
    sizes_a, sizes_b = [300, 1, 777, 64], [500, 90, 3, 640]
    A = unit(rng, sum(sizes_a)); B = unit(rng, sum(sizes_b))
    sa, sb = np.r_[0, np.cumsum(sizes_a)], np.r_[0, np.cumsum(sizes_b)]
    idx, d1, d2 = eyoc_amd.knn2_segmented(torch.from_numpy(A), torch.from_numpy(B), sa, sb)
    i1, e1 = knn1_segmented(torch.from_numpy(A), torch.from_numpy(B), sa, sb)
    np.testing.assert_array_equal(idx.cpu().numpy(), i1.cpu().numpy())
    np.testing.assert_array_equal(d1.cpu().numpy(), e1.cpu().numpy())
    for s in range(4):
        ri, r1, r2 = ol.knn2(A[sa[s]:sa[s + 1]], B[sb[s]:sb[s + 1]])
        np.testing.assert_array_equal(idx.cpu().numpy()[sa[s]:sa[s + 1]], ri)
        np.testing.assert_array_equal(d2.cpu().numpy()[sa[s]:sa[s + 1]], r2)


def test_lowe_topk_weights_and_order():
    import eyoc_amd
    from oracle import labels as ol
    rng = np.random.default_rng(9)
    d1 = rng.uniform(0, 1.5, 6000).astype(np.float32)
    d2 = (d1 + rng.uniform(0, 0.5, 6000).astype(np.float32)).astype(np.float32)
    d1[:40] = 0.0                                        # clamp branch
    d2[10:20] = d1[10:20]                                # ratio exactly 1 -> weight 0, ties in query order
    for k in (5000, 6000, 1):
        idx, w = eyoc_amd.lowe_topk(torch.from_numpy(d1), torch.from_numpy(d2), k)
        ws = ol.lowe_weights(d1, d2)
        ri, _, rw = ol.topk_matches(ws, np.arange(6000), k)
        np.testing.assert_array_equal(idx.cpu().numpy(), ri)
        np.testing.assert_array_equal(w.cpu().numpy(), rw)


def make_pair(rng, n0, n1):
    """Two clouds around two sensors with a shared set of distinctive features."""
    C0 = rng.uniform(-60, 60, (n0, 3)).astype(np.float32)
    C1 = rng.uniform(-60, 60, (n1, 3)).astype(np.float32)
    F0, F1 = unit(rng, n0), unit(rng, n1)
    m = min(n0, n1) // 2
    F1[:m] = (F0[:m] + 0.05 * rng.normal(size=(m, 32))).astype(np.float32)
    F1[:m] /= np.linalg.norm(F1[:m], axis=1, keepdims=True)
    return C0, F0, C1, F1


@pytest.mark.parametrize("feature_filter,spatial_filter", [("Lowe", "Spherical"), ("Lowe", "None"), ("None", "Spherical")])
def test_match_and_filter_corr_vs_oracle(feature_filter, spatial_filter):
    import eyoc_amd
    from oracle import labels as ol
    rng = np.random.default_rng(21)
    pairs = [make_pair(rng, 1500, 1300), make_pair(rng, 800, 2000), make_pair(rng, 1000, 1000)]
    C0 = [torch.from_numpy(p[0]) for p in pairs]; F0 = [torch.from_numpy(p[1]) for p in pairs]
    C1 = [torch.from_numpy(p[2]) for p in pairs]; F1 = [torch.from_numpy(p[3]) for p in pairs]
    matches, unc = eyoc_amd.match_and_filter_corr(C0, F0, C1, F1, radius=20, feature_filter=feature_filter,
                                                  spatial_filter=spatial_filter, num_corres=700)
    rm, ru = ol.match_and_filter_corr([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs],
                                      [p[3] for p in pairs], 20, feature_filter, spatial_filter, num_corres=700)
    np.testing.assert_array_equal(matches.numpy(), rm)
    assert len(unc) == len(ru)
    for a, b in zip(unc, ru):
        np.testing.assert_array_equal(a.cpu().numpy(), b)
    if spatial_filter == "Spherical":
        assert 0 < sum(len(u) for u in ru) < len(rm)


def synthetic_dist_sim_map(seed=0):
    """A table of the reference's shape family (config/dist_sim_plot/*.npz: six float64 slices [gap cells, distance cells],
    similarity decaying with both coordinates) - data, like a checkpoint; the real tables stay with the reference."""
    rng = np.random.default_rng(seed)
    out = {}
    for i, shape in enumerate([(12, 16), (18, 16), (20, 18), (20, 18), (20, 18), (20, 18)]):
        gy, gx = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
        out[i] = np.clip(0.85 * np.exp(-0.12 * gy - 0.05 * gx) + 0.05 * rng.normal(size=shape), 0.0, 1.0)
    return out


def test_similarity_filter_vs_oracle():
    """The "Similarity" spatial filter (lib/trainer.py:1118-1149): table lookup at (centre-distance gap, smaller centre
    distance) + threshold; kept pairs and their order bit-exact against oracle/labels.similarity_mask, including the
    clamped cells (distances beyond the table) and every frame-distance slice."""
    import eyoc_amd
    from oracle import labels as ol
    rng = np.random.default_rng(44)
    table = synthetic_dist_sim_map()
    C0 = (rng.uniform(-1, 1, (3000, 3)) * rng.choice([5, 40, 150], (3000, 1))).astype(np.float32)
    C1 = (rng.uniform(-1, 1, (2500, 3)) * rng.choice([5, 40, 150], (2500, 1))).astype(np.float32)
    a = rng.integers(0, 3000, 4000)
    b = rng.integers(0, 2500, 4000)
    for frame_distance in (0, 4, 7, 12, 19, 26, 100):
        for thresh in (0.4, 0.1):
            got = eyoc_amd.similarity_filter(torch.from_numpy(C0), torch.from_numpy(C1), torch.from_numpy(a), torch.from_numpy(b),
                                             table, frame_distance, thresh).cpu().numpy()
            mask = ol.similarity_mask(C0, C1, a, b, table, frame_distance, thresh)
            np.testing.assert_array_equal(got, np.stack([a[mask], b[mask]], 1))
            assert 0 < mask.sum() < len(mask)
    # through match_and_filter_corr
    pairs = [make_pair(rng, 1500, 1300), make_pair(rng, 800, 2000)]
    args = lambda f: ([f(p[0]) for p in pairs], [f(p[1]) for p in pairs], [f(p[2]) for p in pairs], [f(p[3]) for p in pairs])
    matches, unc = eyoc_amd.match_and_filter_corr(*args(torch.from_numpy), feature_filter="Lowe", spatial_filter="Similarity",
                                                  frame_distance=[3, 17], num_corres=600, dist_sim_map=table, similarity_thresh=0.3)
    rm, ru = ol.match_and_filter_corr(*args(lambda x: x), 20, "Lowe", "Similarity", num_corres=600, frame_distance=[3, 17],
                                      dist_sim_map=table, similarity_thresh=0.3)
    np.testing.assert_array_equal(matches.numpy(), rm)
    for x, y in zip(unc, ru):
        np.testing.assert_array_equal(x.cpu().numpy(), y)
    with pytest.raises(ValueError):
        eyoc_amd.match_and_filter_corr(*args(torch.from_numpy), spatial_filter="Similarity")


def test_correspondences_under_pose_vs_oracle():
    import eyoc_amd
    from oracle import labels as ol
    rng = np.random.default_rng(33)
    n0, n1 = 4000, 4500
    ang = 0.3
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    T[:3, 3] = [4.0, -2.0, 0.3]
    P0 = rng.uniform(-40, 40, (n0, 3)).astype(np.float32)
    P1 = np.concatenate([ol.apply_pose(T, P0[:3000]) + rng.normal(0, 0.3, (3000, 3)).astype(np.float32),
                         rng.uniform(-40, 40, (n1 - 3000, 3)).astype(np.float32)]).astype(np.float32)
    sel = rng.permutation(n0)[:2500]
    got = eyoc_amd.correspondences_under_pose(torch.from_numpy(P0), torch.from_numpy(P1), T, pos_sel=sel, max_dist=2.0)
    ref = ol.correspondences_under_pose(P0, P1, T, sel, 2.0)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    assert 1500 < len(ref) < 2500
    # default draw: at most n_sample pairs, every one within the bound
    out = eyoc_amd.correspondences_under_pose(torch.from_numpy(P0), torch.from_numpy(P1), T, n_sample=1000)
    assert 0 < len(out) <= 1000
