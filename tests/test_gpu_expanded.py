"""``ResUNetExpanded`` / ``ResUNetExpBN2C`` (model/resunet.py:254-490; named in scripts/train_kitti.sh) in eval mode on the packed plan
(``eyoc_model_desc.expanded``, EYOC_VERSION 111): every stage runs ``block<i> -> norm<i>_2 -> block<i>_2`` - the second norm is one
elementwise layer (csrc/model.hip k_affine: it cannot be folded into a neighbour), the second block two more stride-1 convolutions on
the staged kernels.  Checked: both arithmetics against the CPU oracle at the forward's usual 1e-4 bar, the layer-by-layer path
(``eyoc_amd.train.forward_layers``, what eval mode ran before) next to it, the tail still riding in the last staged layer's epilogue,
and the split16 range guard on a stand-alone norm's output."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL = 1e-4


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _model(sd):
    import eyoc_amd
    m = eyoc_amd.load_model("ResUNetExpBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.cuda().eval()


def _x(coords, feats):
    import eyoc_amd
    return eyoc_amd.SparseTensor(torch.from_numpy(feats).cuda(), coordinates=torch.from_numpy(coords).cuda())


@pytest.fixture(scope="module")
def batch():
    """Two 12k-voxel clouds: >= 8192 rows, i.e. automatic arithmetic = split16 on Z-ordered rows (the bench's path)."""
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    p = syn.make_pair(7, beams=32, azimuths=1000, band=None)
    coords = syn.batch_coords([p["coords0"], p["coords1"]])
    feats = np.ones((len(coords), 1), np.float32)
    sd = syn.make_weights(seed=33, expanded=True)
    assert len(coords) >= 8192
    return dict(coords=coords, feats=feats, sd=sd, want=orr.resunet_forward(sd, coords, feats).numpy())


def test_expanded_eval_on_the_packed_plan_matches_the_oracle_in_both_arithmetics(batch):
    from eyoc_amd import _lib as L
    from eyoc_amd.train import forward_layers
    lib = L.load()
    model = _model(batch["sd"])
    x = _x(batch["coords"], batch["feats"])
    errs = {}
    for mode in ("auto", "split16", "fp32"):
        model.spconv_math = mode
        got = model(x).F.cpu().numpy()
        assert model.last_spconv_math == ("fp32" if mode == "fp32" else "split16")
        errs[mode] = rel_err(got, batch["want"])
        cos = (got * batch["want"]).sum(1)
        assert errs[mode] < REL and cos.min() > 1 - 1e-6, (mode, errs[mode], float(cos.min()))
    # 23 layers of ResUNet2 + 7 stages x (norm + two convolutions)
    assert lib.eyoc_model_num_layers(model._handle) == 23 + 7 * 3
    work = model.layer_work(x)
    names = [w["name"] for w in work]
    assert work[3]["pairs"] == 0 and work[4]["pairs"] == work[1]["pairs"] > 0
    assert names[1:8] == ["block1.conv1", "block1.conv2", "norm1_2", "block1_2.conv1", "block1_2.conv2", "conv2", "block2.conv1"]
    assert names[-3:] == ["block2_tr_2.conv2", "conv1_tr", "final"]
    with torch.no_grad():
        ref = forward_layers(model, x).F.cpu().numpy()                       # layer by layer from the parameters (fp32 kernels)
    e_layers = rel_err(ref, batch["want"])
    print(f"ResUNetExpBN2C eval on {len(batch['coords'])} voxels: packed plan {errs}, layer by layer {e_layers:.2e}")
    assert e_layers < REL
    # permuting the caller's rows permutes the output (Z-ordered maps inside)
    model.spconv_math = "auto"
    a = model(x).F.cpu().numpy()
    perm = np.random.default_rng(1).permutation(len(batch["coords"]))
    b = model(_x(batch["coords"][perm], batch["feats"][perm])).F.cpu().numpy()
    assert rel_err(b, a[perm]) < 1e-5


def test_expanded_tail_rides_in_the_last_staged_epilogue_bit_for_bit(batch):
    """eyoc_model_fuse_tail 1 / 2 (the 1x1 pair as one kernel / in block2_tr_2.conv2's epilogue) give the same bits; 0 (two launches)
    sums in another order: a few fp32 ulps, as for ResUNetBN2C (test_gpu_split16.py)."""
    from eyoc_amd import _lib as L
    model = _model(batch["sd"])
    x = _x(batch["coords"], batch["feats"])
    outs = []
    prev = L.knob("eyoc_model_fuse_tail", 2)
    try:
        for mode in (0, 1, 2):
            L.knob("eyoc_model_fuse_tail", mode)
            outs.append(model(x).F.clone())
    finally:
        L.knob("eyoc_model_fuse_tail", prev)
    assert torch.equal(outs[1], outs[2])
    assert float((outs[0] - outs[1]).abs().max() / outs[1].abs().max()) < 2e-6


def test_expanded_small_cloud_and_state_changes(batch):
    """< 8192 rows: fp32 rows in the caller's order; an in-place parameter edit (the EMA labeler sync, lib/trainer.py:1509-1513)
    repacks the stand-alone norms too."""
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    p = syn.make_pair(6, beams=16, azimuths=500, band=None)
    coords = syn.batch_coords([p["coords0"]])
    feats = np.ones((len(coords), 1), np.float32)
    sd = copy.deepcopy(batch["sd"])
    model = _model(sd)
    x = _x(coords, feats)
    got = model(x).F.cpu().numpy()
    assert model.last_spconv_math == "fp32"
    assert rel_err(got, orr.resunet_forward(sd, coords, feats).numpy()) < REL
    with torch.no_grad():
        model.norm3_2.bn.weight.mul_(0.5)
        model.norm3_2.bn.bias.add_(0.25)
    sd["norm3_2.bn.weight"] = sd["norm3_2.bn.weight"] * np.float32(0.5)
    sd["norm3_2.bn.bias"] = sd["norm3_2.bn.bias"] + np.float32(0.25)
    got2 = model(x).F.cpu().numpy()
    assert rel_err(got2, got) > 1e-3
    assert rel_err(got2, orr.resunet_forward(sd, coords, feats).numpy()) < REL


def test_expanded_range_guard_sees_a_stand_alone_norm(batch):
    """A stand-alone norm whose output leaves the fp16 range (weights x 1e5): explicit split16 raises EYOC_ERR_RANGE, automatic mode
    answers in fp32 and still matches the oracle."""
    from eyoc_amd import _lib as L
    from oracle import resunet as orr
    sd = copy.deepcopy(batch["sd"])
    s = np.float32(1.0e5)
    sd["norm2_tr_2.bn.weight"] = sd["norm2_tr_2.bn.weight"] * s
    sd["norm2_tr_2.bn.bias"] = sd["norm2_tr_2.bn.bias"] * s
    for k in ("block2_tr_2.conv1.kernel", ):                                   # keep the rest of the network O(1)
        sd[k] = sd[k] / s
    sd["block2_tr_2.norm2.bn.weight"] = sd["block2_tr_2.norm2.bn.weight"] * np.float32(0.0)   # the residual (1e5 x) would swamp conv2: drop conv2's share so the oracle comparison stays meaningful
    want = orr.resunet_forward(sd, batch["coords"], batch["feats"]).numpy()
    model = _model(sd)
    x = _x(batch["coords"], batch["feats"])
    model.spconv_math = "split16"
    with pytest.raises(L.EyocError) as ei:
        model(x)
    assert ei.value.code == L.ERR_RANGE
    model.spconv_math = "auto"
    got = model(x).F.cpu().numpy()
    assert model.last_spconv_math == "fp32"
    assert np.isfinite(got).all() and rel_err(got, want) < REL
