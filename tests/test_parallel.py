"""CPU tests of the multi-GPU plumbing: byte-balanced sharding and the variable-length gather
(gloo, world_size 2 -- the same code path bench.py runs over RCCL with one process per GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tokenizers_amd.parallel import gather_to_root, shard_documents


def test_shard_documents_is_contiguous_and_byte_balanced():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 9000, size=5000)
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    for world in (1, 2, 3, 8):
        sh = shard_documents(off, world)
        assert sh[0][0] == 0 and sh[-1][1] == len(lens)
        assert all(sh[r][1] == sh[r + 1][0] for r in range(world - 1))
        sizes = [off[b] - off[a] for a, b in sh]
        assert max(sizes) - min(sizes) <= 2 * 9000
    assert shard_documents(np.array([0], dtype=np.int64), 4) == [(0, 0)] * 4
    assert shard_documents(np.array([0, 0, 0], dtype=np.int64), 2)[-1][1] == 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(rank)
        n_docs = [7, 0, 5][rank % 3] if world > 1 else 4
        counts = rng.integers(0, 6, size=n_docs)
        ids = torch.arange(int(counts.sum()), dtype=torch.int32) + 1000 * rank
        offs = torch.zeros(n_docs + 1, dtype=torch.int64)
        offs[1:] = torch.from_numpy(np.cumsum(counts))
        out = gather_to_root(ids, offs, torch.device("cpu"))
        if rank == 0:
            q.put((out[0].tolist(), out[1].tolist()))
        else:
            assert out is None
            q.put((ids.tolist(), counts.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_to_root_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # reconstruct the expectation from the same seeds
    exp_ids, exp_counts = [], []
    for r in range(world):
        rng = np.random.default_rng(r)
        n_docs = [7, 0, 5][r % 3]
        counts = rng.integers(0, 6, size=n_docs)
        exp_ids += (np.arange(int(counts.sum())) + 1000 * r).tolist()
        exp_counts += counts.tolist()
    root = [g for g in got if len(g[1]) == len(exp_counts) + 1 and g[0] == exp_ids]
    assert root, got
    assert root[0][1] == [0] + np.cumsum(exp_counts).tolist()
