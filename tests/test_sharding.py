"""N>1 path on CPU: world_size-2 gloo.  The sample sharding + single all-reduce must reproduce the
single-rank moments; the per-shard moments come from the CPU oracle here (the HIP engine plugs
into the same ``accumulate_sharded``)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from bayesnn_fpga_amd.sharding import accumulate_sharded, shard_range


def test_shard_range_partitions():
    for total in (1, 7, 8, 100, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_range(100, r, 8)[1] - shard_range(100, r, 8)[0] for r in range(8)] == [13, 13, 13, 13, 12, 12, 12, 12]
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def _oracle_accumulate(model, x, seed):
    from oracle import mcd

    def fn(S, t0, n):
        logits, probs = mcd.mcd_passes(model, x, n, seed, t_begin=t0)
        S[0] += torch.from_numpy(probs.sum(0))
        S[1] += torch.from_numpy((probs ** 2).sum(0))
        S[2] += torch.from_numpy(logits.sum(0))
    return fn


def _worker(rank, world, port, T, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
    from oracle.resnet18 import ResNet18MCEarlyExit
    torch.manual_seed(0)
    model = synthetic_weights_(ResNet18MCEarlyExit(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 0)
    x = synthetic_images(2, seed=1234)
    S = torch.zeros(3, 4, 2, 10, dtype=torch.float64)
    accumulate_sharded(_oracle_accumulate(model, x, 42), S, T)
    if rank == 0:
        np.save(out_path, S.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_rank(tmp_path):
    T = 5                                          # odd: ranks get 3 and 2 samples
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "S.npy")
    mp.spawn(_worker, args=(2, port, T, out), nprocs=2, join=True)
    S2 = np.load(out)
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
    from oracle.resnet18 import ResNet18MCEarlyExit
    torch.manual_seed(0)
    model = synthetic_weights_(ResNet18MCEarlyExit(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 0)
    x = synthetic_images(2, seed=1234)
    S1 = torch.zeros(3, 4, 2, 10, dtype=torch.float64)
    accumulate_sharded(_oracle_accumulate(model, x, 42), S1, T)         # no process group: single rank
    np.testing.assert_allclose(S2, S1.numpy(), rtol=1e-12, atol=1e-12)
    mean = S2[0] / T
    assert np.allclose(mean.sum(-1), 1.0, atol=1e-6)
