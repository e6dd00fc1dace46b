"""Multi-GPU plumbing for the encode_batch path: one process per GPU, documents sharded by rank.

The reference's only parallelism is a data-parallel map over documents on a Rayon pool
(utils/parallelism.rs:85-106, used at tokenizer/mod.rs:1345-1348) followed by ``collect``.
Here the map is the per-GPU kernel pipeline and ``collect`` is a variable-length gather of the
final CSR buffers to a root rank over RCCL/xGMI: one tiny all_gather of sizes, then one
point-to-point message per (peer, buffer) -- xGMI is a full mesh, so every peer sends to the root
over its own link; there is no ring and no reduction.

``encode_batch_sharded`` is the whole step (shard -> per-rank encode -> gather); ``shard_documents`` and
``gather_to_root`` are its two halves.
"""
from __future__ import annotations

import os

import numpy as np


def shard_documents(doc_offsets: np.ndarray, world: int) -> list[tuple[int, int]]:
    """Contiguous document ranges with ~equal BYTES per rank (order preserved by rank order).

    ``doc_offsets`` is the int64 CSR of the whole batch.  Returns [(doc_lo, doc_hi)] per rank.
    Cuts fall on document boundaries (documents are never split: the pre-tokenizer rules stop
    at document edges).
    """
    n_docs = len(doc_offsets) - 1
    total = int(doc_offsets[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        d = int(np.searchsorted(doc_offsets, target, side="left"))
        d = min(max(d, cuts[-1]), n_docs)
        cuts.append(d)
    cuts.append(n_docs)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def gather_to_root(ids, tok_offsets, device, root: int = 0, group=None, n_tokens_dev=None):
    """Gather every rank's (ids[T_r], tok_offsets[n_r+1]) to ``root`` in rank order.

    Tensors live on ``device`` (cuda for RCCL, cpu for gloo).  Returns on the root
    ``(ids_all[int32 sum T_r], tok_offsets_all[int64 sum n_r + 1])`` and ``None`` elsewhere.

    ``n_tokens_dev``: a one-element int64 tensor on ``device`` holding T_r, for callers whose encode is still in flight on
    the current stream (``ids`` is then a capacity-sized view): the all_gather of the sizes is the step's only host
    synchronisation, and the messages leave straight from the tokenizer's workspace -- no copy, no earlier sync.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_docs = int(tok_offsets.shape[0]) - 1
    counts = (tok_offsets[1:] - tok_offsets[:-1]).to(torch.int32).contiguous()
    if n_tokens_dev is None:
        mine = torch.tensor([int(ids.shape[0]), n_docs], dtype=torch.int64, device=device)
    else:
        mine = torch.cat([n_tokens_dev.reshape(1).to(torch.int64), torch.tensor([n_docs], dtype=torch.int64, device=device)])
    sizes = torch.empty(2 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.cpu().view(world, 2).tolist()
    n_tok = sizes[rank][0]
    if n_tok > int(ids.shape[0]):
        raise RuntimeError(f"gather_to_root: rank {rank} holds {n_tok} tokens but its ids view has {int(ids.shape[0])} elements")
    ids = ids[:n_tok].contiguous()
    # The message leaves from a fresh allocator block by default: sending straight from the hipMalloc'd workspace (no copy) has run on
    # one GPU only (TKAMD_GATHER_CLONE=0 selects it; to be made the default once it has run over RCCL on several).
    if os.environ.get("TKAMD_GATHER_CLONE") != "0":
        ids = ids.clone()
    if rank == root:
        tot_tok = sum(s[0] for s in sizes)
        tot_docs = sum(s[1] for s in sizes)
        ids_all = torch.empty(tot_tok, dtype=ids.dtype, device=device)
        counts_all = torch.empty(tot_docs, dtype=torch.int32, device=device)
        ops = []
        to, do = 0, 0
        for r in range(world):
            t_r, d_r = sizes[r]
            if r == root:
                ids_all[to:to + t_r].copy_(ids)
                counts_all[do:do + d_r].copy_(counts)
            else:
                if t_r:
                    ops.append(dist.P2POp(dist.irecv, ids_all[to:to + t_r], r, group))
                if d_r:
                    ops.append(dist.P2POp(dist.irecv, counts_all[do:do + d_r], r, group))
            to += t_r
            do += d_r
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        offs = torch.zeros(tot_docs + 1, dtype=torch.int64, device=device)
        torch.cumsum(counts_all, dim=0, out=offs[1:])
        return ids_all, offs
    ops = []
    if n_tok:
        ops.append(dist.P2POp(dist.isend, ids, root, group))
    if n_docs:
        ops.append(dist.P2POp(dist.isend, counts, root, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None


def encode_batch_sharded(encode_shard, buf: np.ndarray, doc_offsets: np.ndarray, device, root: int = 0, group=None):
    """One encode_batch over all ranks of ``group``: the reference's map + collect (tokenizer/mod.rs:1345-1348).

    Every rank calls this with the same whole batch (``buf`` = UTF-8 bytes, ``doc_offsets`` = int64 CSR).  Rank r takes its
    byte-balanced contiguous document range (:func:`shard_documents`), rebases the CSR, and hands
    ``(shard_bytes uint8[n + TEXT_PAD], shard_offsets int64[n_r + 1])`` to ``encode_shard``, which returns
    ``(ids, tok_offsets)`` or ``(ids_capacity_view, tok_offsets, n_tokens_dev)`` as tensors on ``device``
    (:func:`device_encoder` wraps a Tokenizer's HIP path that way).  The root gets ``(ids_all, tok_offsets_all)`` -- equal,
    by construction of the document-boundary cuts, to the single-GPU result on the whole batch -- the others ``None``.
    """
    import torch.distributed as dist

    from ._lib import TEXT_PAD
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.int64)
    lo, hi = shard_documents(doc_offsets, world)[rank]
    b0, b1 = int(doc_offsets[lo]), int(doc_offsets[hi])
    shard = np.zeros(b1 - b0 + TEXT_PAD, dtype=np.uint8)
    shard[: b1 - b0] = buf[b0:b1]
    res = encode_shard(shard, doc_offsets[lo:hi + 1] - b0)
    ids, tok_offsets = res[0], res[1]
    return gather_to_root(ids, tok_offsets, device, root=root, group=group, n_tokens_dev=res[2] if len(res) > 2 else None)


def device_encoder(tok, device):
    """``encode_shard`` for :func:`encode_batch_sharded` over a :class:`tokenizers_amd.Tokenizer`: H2D of the shard, the kernel
    pipeline on the current stream, results left in the tokenizer's workspace (no host synchronisation)."""
    import torch

    def encode_shard(shard: np.ndarray, offsets: np.ndarray):
        d_text = torch.from_numpy(shard).to(device)
        d_off = torch.from_numpy(np.ascontiguousarray(offsets)).to(device)
        b = tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), len(offsets) - 1, int(offsets[-1]),
                                    stream=torch.cuda.current_stream(device).cuda_stream, unsynced=True)
        b._keep = (d_text, d_off)                      # inputs stay alive until the results have been consumed
        encode_shard.last = b
        return b.ids_tensor_unsynced(), b.tok_offsets_tensor(), b.n_tokens_tensor()
    return encode_shard
