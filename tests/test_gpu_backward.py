"""GPU parity of the training-side kernels: gradients of a sparse convolution layer against autograd
through the oracle's gather -> matmul -> index_add (per layer kind, on a full 31k-voxel cloud), and the
hardest-contrastive loss of lib/trainer.py:935-991 (value and feature gradients) against its torch-CPU restatement.
Tolerance: 1e-4 of the largest gradient entry (north_star's floating-point bar)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL = 1e-4


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def cloud():
    import eyoc_amd
    from eyoc_amd import synthetic as syn
    from oracle import coords as oc
    p = syn.make_pair(1)
    coords = syn.batch_coords([p["coords0"]])
    cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
    return cm, oc.build_maps(coords)


def _oracle_grads(nbr, x, W, dy):
    from oracle import resunet as orr
    xt = torch.from_numpy(x).requires_grad_(True)
    Wt = torch.from_numpy(W).requires_grad_(True)
    out = orr.sparse_conv(xt, nbr, Wt)
    out.backward(torch.from_numpy(dy))
    return out.detach().numpy(), xt.grad.numpy(), Wt.grad.numpy()


@pytest.mark.parametrize("kind,level,cin,cout", [("s1", 0, 32, 32), ("s1", 0, 64, 64), ("s1", 1, 64, 64), ("s1", 2, 128, 128),
                                                 ("down", 0, 32, 64), ("down", 1, 64, 128), ("up", 0, 128, 64), ("up", 1, 256, 64),
                                                 ("down", 2, 128, 256), ("up", 2, 256, 128), ("s1", 3, 256, 256), ("s1", 2, 128, 128),
                                                 # concatenated decoder inputs of ResUNetBN2B / FatBN: the input gradient in column blocks
                                                 ("up", 1, 192, 64), ("up", 2, 384, 128)])
def test_layer_gradients_vs_oracle_autograd(cloud, kind, level, cin, cout):
    """grad-input = the forward kernel over the transposed rulebook with W[k]^T; grad-weight = eyoc_spconv_grad_weight."""
    from eyoc_amd.autograd import sparse_conv
    cm, maps = cloud
    KIND = {"s1": 0, "down": 1, "up": 2}
    table = cm.table(KIND[kind], level)
    table_t = None if kind == "s1" else cm.table(KIND["up" if kind == "down" else "down"], level)
    nbr = maps[kind][level]
    np.testing.assert_array_equal(table.cpu().numpy(), nbr)
    n_out = nbr.shape[1]
    n_in = len(maps["cm"][level + 1]) if kind == "up" else len(maps["cm"][level])
    rng = np.random.default_rng(cin + 3 * cout + level)
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    dy = rng.normal(size=(n_out, cout)).astype(np.float32)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    Wd = torch.from_numpy(W).cuda().requires_grad_(True)
    out = sparse_conv(xd, Wd, table, table_t)
    out.backward(torch.from_numpy(dy).cuda())
    ref_out, ref_dx, ref_dw = _oracle_grads(nbr, x, W, dy)
    e = (rel_err(out.detach().cpu().numpy(), ref_out), rel_err(xd.grad.cpu().numpy(), ref_dx), rel_err(Wd.grad.cpu().numpy(), ref_dw))
    print(f"{kind} level {level} {cin}->{cout}: forward {e[0]:.2e}  grad-input {e[1]:.2e}  grad-weight {e[2]:.2e}")
    assert max(e) < REL
    # deterministic: a second backward gives the same bits
    first_dw, first_dx = Wd.grad.cpu().numpy().copy(), xd.grad.cpu().numpy().copy()
    xd.grad = None; Wd.grad = None
    sparse_conv(xd, Wd, table, table_t).backward(torch.from_numpy(dy).cuda())
    np.testing.assert_array_equal(Wd.grad.cpu().numpy(), first_dw)
    np.testing.assert_array_equal(xd.grad.cpu().numpy(), first_dx)


def test_identity_map_and_chain_of_two_layers(cloud):
    """1x1 convolution (identity map) and a conv -> relu -> conv chain: autograd composes the two Functions."""
    from eyoc_amd.autograd import sparse_conv
    from oracle import resunet as orr
    cm, maps = cloud
    table = cm.table(0, 1)
    nbr = maps["s1"][1]
    n = nbr.shape[1]
    rng = np.random.default_rng(5)
    x = rng.normal(size=(n, 64)).astype(np.float32)
    W1 = (rng.normal(size=(27, 64, 64)) / 24).astype(np.float32)
    W2 = (rng.normal(size=(1, 64, 32)) / 8).astype(np.float32)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    W1d, W2d = torch.from_numpy(W1).cuda().requires_grad_(True), torch.from_numpy(W2).cuda().requires_grad_(True)
    y = sparse_conv(torch.relu(sparse_conv(xd, W1d, table)), W2d, None)
    y.pow(2).sum().backward()
    xt = torch.from_numpy(x).requires_grad_(True)
    W1t, W2t = torch.from_numpy(W1).requires_grad_(True), torch.from_numpy(W2).requires_grad_(True)
    ident = np.arange(n, dtype=np.int32)[None]
    yt = orr.sparse_conv(torch.relu(orr.sparse_conv(xt, nbr, W1t)), ident, W2t)
    yt.pow(2).sum().backward()
    assert rel_err(xd.grad.cpu().numpy(), xt.grad.numpy()) < REL
    assert rel_err(W1d.grad.cpu().numpy(), W1t.grad.numpy()) < REL
    assert rel_err(W2d.grad.cpu().numpy(), W2t.grad.numpy()) < REL


def test_hardest_contrastive_loss_and_gradients():
    """lib/trainer.py:935-991: same draws, same hardest negatives, same masked means; gradients w.r.t. both feature
    matrices (what `loss.backward()` at lib/trainer.py:1667 sends into the network)."""
    from eyoc_amd.autograd import contrastive_hardest_negative_loss as gpu_loss
    from oracle.loss import contrastive_hardest_negative_loss as ref_loss
    rng = np.random.default_rng(0)
    N0, N1 = 6000, 5500
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    F0 = unit(rng.normal(size=(N0, 32)))
    F1 = unit(rng.normal(size=(N1, 32)))
    pos = np.stack([rng.permutation(N0)[:3000], rng.permutation(N1)[:3000]], 1).astype(np.int64)
    F1[pos[:1500, 1]] = unit(F0[pos[:1500, 0]] + 0.2 * rng.normal(size=(1500, 32)))      # half of the positives really match
    for num_pos, num_hn in ((1024, 256), (5192, 2048)):
        F0d = torch.from_numpy(F0).cuda().requires_grad_(True)
        F1d = torch.from_numpy(F1).cuda().requires_grad_(True)
        pl, nl = gpu_loss(F0d, F1d, torch.from_numpy(pos), num_pos, num_hn, rng=np.random.RandomState(7))
        (pl + nl).backward()
        F0t = torch.from_numpy(F0).requires_grad_(True)
        F1t = torch.from_numpy(F1).requires_grad_(True)
        rpl, rnl = ref_loss(F0t, F1t, pos, num_pos, num_hn, rng=np.random.RandomState(7))
        (rpl + rnl).backward()
        pl, nl, rpl, rnl = (float(v.detach()) for v in (pl, nl, rpl, rnl))
        print(f"num_pos {num_pos}: pos_loss {pl:.6f} (ref {rpl:.6f})  neg_loss {nl:.6f} (ref {rnl:.6f})")
        assert abs(pl - rpl) < 1e-5 * max(1.0, abs(rpl))
        assert abs(nl - rnl) < 1e-5 * max(1.0, abs(rnl))
        assert rel_err(F0d.grad.cpu().numpy(), F0t.grad.numpy()) < REL
        assert rel_err(F1d.grad.cpu().numpy(), F1t.grad.numpy()) < REL
