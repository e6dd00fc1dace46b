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


# ---- the whole multi-rank step: shard -> per-rank encode -> gather == the unsharded encode ----------------------------
def _sharded_docs():
    from oracle import synth
    return synth.gen_lines(700, text_seed=91) + ["", "x" * 3000, ""] + synth.stress_lines(seed=31, n=200) + [""]


def _oracle_encoder(js, device, capacity_view):
    """CPU stand-in for the per-rank HIP encode: the oracle.  With capacity_view it hands the gather a capacity-sized ids buffer
    and the token count as a tensor, like the device path does."""
    from oracle import oracle as orc
    o = orc.Oracle(js)

    def encode_shard(shard, offsets):
        raw = bytes(shard[: int(offsets[-1])])
        docs = [raw[int(offsets[i]):int(offsets[i + 1])].decode("utf-8") for i in range(len(offsets) - 1)]
        r = o.encode_batch(docs)
        ids = torch.from_numpy(r.ids.astype(np.int32))
        to = torch.from_numpy(r.tok_offsets.astype(np.int64))
        if not capacity_view:
            return ids, to
        cap = torch.full((len(ids) + 37,), -1, dtype=torch.int32)
        cap[: len(ids)] = ids
        return cap, to, torch.tensor([len(ids)], dtype=torch.int64)
    return encode_shard


def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import synth
        import tokenizers_amd as ta
        from tokenizers_amd.parallel import encode_batch_sharded
        js = synth.load_or_train_gpt2()
        buf, off = ta.pack_documents(_sharded_docs())
        out = encode_batch_sharded(_oracle_encoder(js, torch.device("cpu"), capacity_view=(world == 3)), buf, off, torch.device("cpu"))
        q.put((rank, None if out is None else (out[0].tolist(), out[1].tolist())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_encode_batch_sharded_equals_unsharded_gloo(world):
    from oracle import oracle as orc
    from oracle import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = orc.Oracle(synth.load_or_train_gpt2()).encode_batch(_sharded_docs())
    assert all(got[r] is None for r in range(1, world))
    assert got[0][0] == exp.ids.astype(np.int32).tolist()
    assert got[0][1] == exp.tok_offsets.tolist()


# ---- the same step with the PRODUCT's kernels on every rank: the SIMT build of csrc/ (tests/harness/simt_build.py) is the encoder, the
# oracle only the checker -- shard cuts, the real pipeline per rank (host entry -> every kernel the device would run), the gather and
# its displacements, compared with the unsharded oracle result.  What a second GPU adds to this is speed and RCCL's own transport.
def _simt_kernel_worker(rank, world, port, q, docs, js, fixed_capacity):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.harness import simt_env
        simt_env.install()                       # this process opens the SIMT build where the product opens libtokenizers_amd.so
        import tokenizers_amd as ta
        from tokenizers_amd.parallel import encode_batch_sharded
        tok = ta.Tokenizer.from_str(js, device=0)

        def encode_shard(shard, offsets):
            r = tok.encode_packed(np.ascontiguousarray(shard[: int(offsets[-1])]), np.ascontiguousarray(offsets))
            ids = torch.from_numpy(np.array(r.ids, copy=True).astype(np.int32))
            to = torch.from_numpy(np.array(r.tok_offsets, copy=True).astype(np.int64))
            if not fixed_capacity:
                return ids, to
            cap = torch.full((len(ids) + 129,), -7, dtype=torch.int32)      # a capacity-sized buffer + the count as a tensor: the device path's shape
            cap[: len(ids)] = ids
            return cap, to, torch.tensor([len(ids)], dtype=torch.int64)
        buf, off = ta.pack_documents(docs)
        out = encode_batch_sharded(encode_shard, buf, off, torch.device("cpu"))
        q.put((rank, None if out is None else (out[0].tolist(), out[1].tolist())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "gpt2"), (3, "gpt2"), (2, "bert_wordpiece_4000"), (3, "llama3_small_6000"), (2, "gpt2-noclone"), (3, "gpt2-noclone")])
def test_sharded_encode_with_the_real_kernels_gloo(world, name, monkeypatch):
    # ("-noclone": TKAMD_GATHER_CLONE=0 -- the messages leave straight from the encoder's buffers, a capacity-sized view with world 3: the
    # path the RCCL gather is to default to once it has run on several GPUs)
    if name.endswith("-noclone"):
        monkeypatch.setenv("TKAMD_GATHER_CLONE", "0")
        name = name[: -len("-noclone")]
    from oracle import oracle as orc
    from oracle import synth
    from tests.helpers import load_tokenizer_json
    from tests.harness import simt_build
    simt_build.build()                           # (once, here: the ranks would race for the build otherwise)
    js = synth.load_or_train_gpt2() if name == "gpt2" else load_tokenizer_json(name)
    docs = synth.gen_lines(260, text_seed=93) + ["", "x" * 3000, ""] + synth.stress_lines(seed=32, n=120) + ["", ""]
    if name.startswith("bert"):
        docs = [d for d in docs if "[" not in d]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_simt_kernel_worker, args=(r, world, port, q, docs, js, world == 3)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = orc.Oracle(js).encode_batch(docs)
    assert all(got[r] is None for r in range(1, world))
    assert got[0][0] == exp.ids.astype(np.int32).tolist()
    assert got[0][1] == exp.tok_offsets.tolist()


@pytest.mark.gpu
@pytest.mark.needs_hw
def test_encode_batch_sharded_on_one_gpu_over_rccl():
    """World size 1 over RCCL (the only size a 1-GPU box offers): the sharded step through the real device encoder and the real
    collective calls equals the plain encode of the whole batch."""
    import subprocess
    import sys
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1')\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "import tokenizers_amd as ta\n"
        "from tokenizers_amd.parallel import encode_batch_sharded, device_encoder\n"
        "from oracle import synth\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "tok = ta.Tokenizer.from_str(synth.load_or_train_gpt2(), device=0)\n"
        "docs = synth.gen_lines(20000, text_seed=92) + ['', 'x' * 3000, '']\n"
        "buf, off = ta.pack_documents(docs)\n"
        "ids, to = encode_batch_sharded(device_encoder(tok, dev), buf, off, dev)\n"
        "ids, to = ids.cpu().numpy().view(np.uint32), to.cpu().numpy()\n"
        "ref = tok.encode_packed(buf, off)\n"
        "assert np.array_equal(ids, ref.ids) and np.array_equal(to, ref.tok_offsets)\n"
        "dist.destroy_process_group(); print('SHARDED_OK')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "SHARDED_OK" in r.stdout, r.stdout + r.stderr


# ---- bench.py's multi-rank control flow (the timed region's contract, the gather leg's watchdog) over gloo ------------------------
def _bench_worker(rank, world, port, q):
    import time
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def step(i):                                    # a stand-in encode: rank r takes (r + 1) x 20 ms a step, and gathers like bench.py's gather leg
            time.sleep(0.02 * (rank + 1))
            calls.append(i)
            ids = torch.arange(5 + rank, dtype=torch.int32)
            offs = torch.tensor([0, 2, 5 + rank], dtype=torch.int64)
            gather_to_root(ids, offs, torch.device("cpu"))
            return i

        def fence():
            dist.barrier()

        def reduce_max(seconds):
            el = torch.tensor([seconds], dtype=torch.float64)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            return float(el.item())
        finished = []
        el, last = bench.run_guarded(lambda: bench.timed_steps(step, 5, 2, fence, reduce_max, finish_last=finished.append), 60.0, lambda: os._exit(3))
        q.put((rank, el, last, calls, finished))
    finally:
        dist.destroy_process_group()


def test_bench_timed_region_and_watchdog_over_gloo():
    """bench.timed_steps / bench.run_guarded are the functions bench.py itself runs: W untimed + exactly K timed steps between two
    barriers, the MAX over ranks of the wall time (every rank reports the slowest rank's figure), the last step's result handed to
    finish_last; and a step that never returns ends the process from the watchdog thread instead of hanging the job."""
    import subprocess
    import sys
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, el, last, calls, finished in got:
        assert calls == [0, 1, 0, 1, 2, 3, 4] and last == 4 and finished == [4]
        assert 5 * 0.04 <= el < 5 * 0.04 + 1.0              # the slow rank's five 40 ms steps, on both ranks
    assert got[0][1] == got[1][1]
    # the watchdog: a "collective" that never completes (a sleep in C, GIL released) -- the thread still fires
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench\n"
            "bench.run_guarded(lambda: time.sleep(60), 0.5, lambda: (print('WATCHDOG', flush=True), os._exit(7)))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 7 and "WATCHDOG" in r.stdout
    import bench
    assert bench.run_guarded(lambda: 41 + 1, 5.0, lambda: os._exit(9)) == 42
