"""World-size-2 checks of the multi-GPU plumbing on CPU (gloo): round-robin sharding, the one-off
weight-blob broadcast and the result gather.  The data path itself has no collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eyoc_amd import dist as edist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = edist.init(backend="gloo")
    assert (r, w) == (rank, world)
    # weight blob: rank 0 owns the packed parameters, everyone ends up with the same bytes
    blob = torch.arange(1000, dtype=torch.float32) * 0.5 if rank == 0 else torch.zeros(1000)
    edist.broadcast_blob(blob, src=0)
    assert torch.equal(blob, torch.arange(1000, dtype=torch.float32) * 0.5)
    # pairs: static round-robin, no overlap, full cover
    mine = edist.shard(11, rank, world)
    assert mine == list(range(rank, 11, world))
    # results: each rank registers its pairs (here: a fake record carrying the pair id), equal counts
    n_local = 5
    ids = [rank + world * j for j in range(n_local)]
    rec = torch.tensor([[float(i), float(i) * 2.0, float(rank)] for i in ids])
    allrec = edist.gather_records(rec)
    assert allrec.shape == (n_local * world, 3)
    assert allrec[:, 0].tolist() == [float(i) for i in range(n_local * world)]     # global pair order restored
    assert edist.max_over_ranks(float(rank + 1), torch.device("cpu")) == float(world)
    edist.barrier()
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), allrec.numpy())
    dist.destroy_process_group()


def test_two_rank_plumbing_on_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "ok0.npy"), np.load(tmp_path / "ok1.npy")
    np.testing.assert_array_equal(a, b)


def test_single_process_is_a_no_op():
    assert edist.shard(5, 0, 1) == [0, 1, 2, 3, 4]
    t = torch.ones(3)
    assert edist.broadcast_blob(t) is t
    assert edist.gather_records(t[None]) is not None
    assert edist.max_over_ranks(3.5, torch.device("cpu")) == 3.5
    edist.barrier()


def _model_worker(rank, world, port, out_dir):
    """``broadcast_model`` itself, on the real packed blob (eyoc_model_pack_host: no GPU needed): rank 0 holds the seeded
    weights, rank 1 a differently initialised module; after the broadcast rank 1's buffer must be rank 0's bytes."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import eyoc_amd
    from eyoc_amd import synthetic as syn
    edist.init(backend="gloo")
    Model = eyoc_amd.load_model("ResUNetBN2C")
    model = Model(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    if rank == 0:
        sd = syn.make_weights()
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    own = model.pack_host()                                    # what this rank would pack from its own parameters
    blob = edist.broadcast_model(model, torch.device("cpu"), src=0)
    assert blob.numel() == model.blob_floats()
    if rank == 0:
        assert torch.equal(blob, own)
    else:
        assert not torch.equal(blob, own), "rank 1 packed different weights - the test would prove nothing"
    np.save(os.path.join(out_dir, f"blob{rank}.npy"), blob.numpy().view(np.uint32))
    dist.destroy_process_group()


def test_broadcast_model_ships_the_real_packed_blob(tmp_path):
    from eyoc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libeyoc_hip.so not built")
    world = 2
    mp.spawn(_model_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "blob0.npy"), np.load(tmp_path / "blob1.npy")
    assert a.size > 8_000_000                                       # BN2C: fp32 + split16 packing of every layer
    np.testing.assert_array_equal(a, b)                             # bit-identical on both ranks
    assert np.count_nonzero(a) > a.size // 2


def _expanded_worker(rank, world, port, out_dir):
    """``ResUNetExpBN2C`` (model/resunet.py:487-490) packs like the rest of the family since EYOC_VERSION 111: the blob holds the
    ``norm<i>_2`` scale / shift and the ``block<i>_2`` weights too, and ``broadcast_model`` ships it as one message."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import eyoc_amd
    edist.init(backend="gloo")
    torch.manual_seed(100 + rank)                               # different random parameters per rank
    model = eyoc_amd.load_model("ResUNetExpBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 1.5)
    own = model.pack_host()
    plain = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    extra = 2 * 27 * (32 * 32 + 3 * 64 * 64 + 2 * 128 * 128 + 256 * 256)     # the seven block<i>_2: two convolutions each
    assert own.numel() == model.blob_floats() >= plain.blob_floats() + 2 * extra    # fp32 fragment order + split16 packing
    blob = edist.broadcast_model(model, torch.device("cpu"), src=0)
    assert torch.equal(blob, own) == (rank == 0)
    np.save(os.path.join(out_dir, f"blob{rank}.npy"), blob.numpy())
    dist.destroy_process_group()


def test_broadcast_model_ships_the_blob_of_an_expanded_model(tmp_path):
    from eyoc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libeyoc_hip.so not built")
    mp.spawn(_expanded_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "blob0.npy"), np.load(tmp_path / "blob1.npy")
    np.testing.assert_array_equal(a, b)
    assert np.count_nonzero(a) > a.size // 2


def _model_worker8(rank, world, port, out_dir):
    """W = 8: the start-up broadcast of the real 70 MB blob to seven other ranks over gloo."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import eyoc_amd
    from eyoc_amd import synthetic as syn
    edist.init(backend="gloo")
    Model = eyoc_amd.load_model("ResUNetBN2C")
    model = Model(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    if rank == 0:
        sd = syn.make_weights()
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    blob = edist.broadcast_model(model, torch.device("cpu"), src=0)
    assert blob.numel() == model.blob_floats()
    # a 64-bit checksum of the bytes per rank (the blobs themselves would be 8 x 70 MB on disk)
    words = blob.numpy().view(np.uint32).astype(np.uint64)
    chk = np.array([int(words.sum()), int((words * (np.arange(words.size, dtype=np.uint64) % 65521 + 1)).sum() & 0xFFFFFFFFFFFF),
                    words.size], dtype=np.uint64)
    np.save(os.path.join(out_dir, f"chk{rank}.npy"), chk)
    mine = edist.shard(545, rank, world)
    np.save(os.path.join(out_dir, f"mine{rank}.npy"), np.asarray(mine))
    edist.barrier()
    dist.destroy_process_group()


def test_eight_rank_broadcast_of_the_real_blob_and_545_pair_shards(tmp_path):
    from eyoc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libeyoc_hip.so not built")
    world = 8
    mp.spawn(_model_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    chks = [np.load(tmp_path / f"chk{r}.npy") for r in range(world)]
    assert int(chks[0][2]) > 8_000_000 and int(chks[0][0]) > 0
    for r in range(1, world):
        np.testing.assert_array_equal(chks[0], chks[r])           # every rank holds rank 0's bytes
    shards = [np.load(tmp_path / f"mine{r}.npy") for r in range(world)]
    assert [len(s) for s in shards] == [69] + [68] * 7             # LoKITTI_50: 545 pairs, pair i on rank i % 8
    assert sorted(np.concatenate(shards).tolist()) == list(range(545))
