"""Multi-GPU plumbing for the encode_batch path: one process per GPU, documents sharded by rank.

The reference's only parallelism is a data-parallel map over documents on a Rayon pool
(utils/parallelism.rs:85-106, used at tokenizer/mod.rs:1345-1348) followed by ``collect``.
Here the map is the per-GPU kernel pipeline and ``collect`` is a variable-length gather of the
final CSR buffers to a root rank over RCCL/xGMI: one tiny all_gather of sizes, then one
point-to-point message per (peer, buffer) -- xGMI is a full mesh, so every peer sends to the root
over its own link; there is no ring and no reduction.
"""
from __future__ import annotations

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


def gather_to_root(ids, tok_offsets, device, root: int = 0, group=None):
    """Gather every rank's (ids[T_r], tok_offsets[n_r+1]) to ``root`` in rank order.

    Tensors live on ``device`` (cuda for RCCL, cpu for gloo).  Returns on the root
    ``(ids_all[int32 sum T_r], tok_offsets_all[int64 sum n_r + 1])`` and ``None`` elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_docs = int(tok_offsets.shape[0]) - 1
    n_tok = int(ids.shape[0])
    counts = (tok_offsets[1:] - tok_offsets[:-1]).to(torch.int32).contiguous()
    ids = ids.contiguous()
    mine = torch.tensor([n_tok, n_docs], dtype=torch.int64, device=device)
    sizes = torch.empty(2 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.cpu().view(world, 2).tolist()
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
    # send from framework-allocated memory: `ids` may be a zero-copy view of the tokenizer's own hipMalloc'd
    # workspace, and a fresh allocator block is the buffer kind RCCL's P2P path is exercised with everywhere
    ids = ids.clone()
    ops = []
    if n_tok:
        ops.append(dist.P2POp(dist.isend, ids, root, group))
    if n_docs:
        ops.append(dist.P2POp(dist.isend, counts, root, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None
