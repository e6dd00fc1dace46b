// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  BPE: whole-word lookup and the merge kernels.


__device__ __forceinline__ void load_key16(const uint8_t* __restrict__ text, uint32_t s, uint32_t len, uint64_t* lo, uint64_t* hi) {
    // ONE byte-unaligned global_load_dwordx4 (legal on gfx950; text buffers carry TKAMD_TEXT_PAD readable slack).
    // The texture-address unit costs about a cycle per lane request for divergent addresses, so requests -- not
    // bytes -- are what this path is priced in.
    const Unaligned16 v = *(const Unaligned16*)(text + s);
    // branch-free masking of the bytes past `len` (selects only, so callers can keep many loads in flight)
    uint32_t nl = min(len, 8u), nh = min(len, 16u) - nl;           // bytes kept in the low / high half
    uint32_t m0 = nl >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nl)) - 1u);
    uint32_t m1 = nl >= 8 ? 0xFFFFFFFFu : (nl > 4 ? ((1u << (8 * (nl - 4))) - 1u) : 0u);
    uint32_t m2 = nh >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nh)) - 1u);
    uint32_t m3 = nh >= 8 ? 0xFFFFFFFFu : (nh > 4 ? ((1u << (8 * (nh - 4))) - 1u) : 0u);
    *lo = ((uint64_t)(v.b & m1) << 32) | (v.a & m0);
    *hi = ((uint64_t)(v.d & m3) << 32) | (v.c & m2);
}

// ---- in-batch word claims (kernels/lookup.hip has the whole story): hash and slot of a word of <= 32 bytes, and the step every model
// kernel ends an entry with: a queued pre-token that HOLDS the claim of its word (QLEN_CLAIM in its queue entry, set by the lookup when its
// compare-and-swap won the slot) copies its finished row to the slot's row, where the compaction finds it for the word's other
// occurrences (a row of more than four tokens names its ids by the claimant's first byte, tmp_ids[s + j]: valid for the whole batch), and
// -- when offsets are requested -- leaves its first byte in claim_pos[slot]: k_token_meta takes a sharer's token ends from the claimant's
// slots of tmp_end.  The slot is recomputed from the entry's bytes (the kernel has just read them); the claims table itself is not read again.
constexpr uint32_t CLAIM_MAX_LEN = 32u;
constexpr uint32_t CLAIM_KEY_MAX = 15u;           // words of <= 15 bytes: the claim entry IS the key (lookup.hip); longer ones name their claimant's bytes
// the whole-word table's hash of the first 16 bytes and the whole length, continued over bytes 16..31 (zero padded)
__device__ __forceinline__ uint32_t claim_hash_long(uint32_t h16, uint32_t k4, uint32_t k5, uint32_t k6, uint32_t k7) {
    return mix32(h16 ^ (k4 * 0x9E3779B1u) ^ (k5 * 0x85EBCA77u) ^ (k6 * 0xC2B2AE3Du) ^ (k7 * 0x27D4EB2Fu));
}
__device__ __forceinline__ uint32_t claim_slot(uint32_t h, uint32_t mask) { return (word_hash2(h) >> 7) & mask; }
__device__ __forceinline__ uint32_t claim_slot_of(const uint8_t* __restrict__ text, uint32_t s, uint32_t len, uint32_t seed, uint32_t mask) {
    uint64_t lo, hi;
    load_key16(text, s, min(len, 16u), &lo, &hi);
    uint32_t h = word_hash1(lo, hi, len, seed);
    if (len > 16u) {
        uint64_t lo2, hi2;
        load_key16(text, s + 16u, len - 16u, &lo2, &hi2);
        h = claim_hash_long(h, (uint32_t)lo2, (uint32_t)(lo2 >> 32), (uint32_t)hi2, (uint32_t)(hi2 >> 32));
    }
    return claim_slot(h, mask);
}
// len_word: the queue entry's length word (QLEN_CLAIM and all)
__device__ __forceinline__ void claim_publish_item(const uint8_t* __restrict__ text, uint32_t seed, uint32_t s, uint32_t len_word, const uint4& row,
                                                   uint32_t claim_mask, uint4* __restrict__ crows, uint32_t* __restrict__ cpos) {
    if (!(len_word & QLEN_CLAIM)) return;
    const uint32_t slot = claim_slot_of(text, s, qitem_len(len_word), seed, claim_mask);
    crows[slot] = row;
    if (cpos) cpos[slot] = s;
}
// (t.pub_rows: set by the host when the model kernels are to publish -- DevTables is what every one of them is handed)
#define TKAMD_PUBLISH_ROW(t_, text_, s_, lenw_, row_) do { if ((t_).pub_rows) claim_publish_item((text_), (t_).word_seed, (s_), (lenw_), (row_), (t_).pub_mask, (uint4*)(t_).pub_rows, (t_).pub_pos); } while (0)

// whole-word probe of a key longer than 16 bytes: hash and compare four bytes at a time (dword loads at any alignment; the text
// carries TEXT_PAD readable bytes past its end, the vocabulary blob 16)
__device__ __forceinline__ bool long_probe(const DevTables& t, const uint8_t* __restrict__ w, uint32_t len, uint32_t* id) {
    uint32_t h = long_key_hash_init(len);
    const uint32_t nw = len >> 2, tail = len & 3u, tmask = (1u << (8u * tail)) - 1u;
    for (uint32_t i = 0; i < nw; ++i) h = long_key_hash_step(h, ((const Unaligned4*)(w + 4u * i))->v);
    if (tail) h = long_key_hash_step(h, ((const Unaligned4*)(w + 4u * nw))->v & tmask);
    h &= t.long_mask;
    for (;;) {
        uint32_t e = t.long_table[h];
        if (!e) return false;
        uint32_t o = t.long_off[e - 1], l = t.long_off[e] - o;
        if (l == len) {
            const uint8_t* b = t.long_blob + o;
            uint32_t i = 0;
            while (i < nw && ((const Unaligned4*)(b + 4u * i))->v == ((const Unaligned4*)(w + 4u * i))->v) ++i;
            if (i == nw && (!tail || ((((const Unaligned4*)(b + 4u * nw))->v ^ ((const Unaligned4*)(w + 4u * nw))->v) & tmask) == 0u)) { *id = t.long_id[e - 1]; return true; }
        }
        h = (h + 1) & t.long_mask;
    }
}


// =================================================================================================
// K_bpe_merge<G>: BPE merge resolution, G lanes per pre-token (G=16: one DPP row, 4 pre-tokens per
// wavefront; G=64: one wavefront).  Lane c holds symbol c of the pre-token (byte-level BPE: one
// initial symbol per byte, models/bpe/model.rs:465-499 with the byte alphabet of byte_level.rs:15-39).
// Replaces: Word::merge_all (models/bpe/word.rs:162-250): "pop the (rank, pos)-minimum mergeable
// adjacent pair, merge, re-queue its two new neighbours".  With every live pair's rank cached in its
// left symbol's lane, the heap top is a min-reduction of (rank << 6 | lane) over the row (4 DPP
// steps) and a merge re-probes exactly the two pairs the reference re-queues (word.rs:218-244).
// Pair -> (rank, new_id) probes hit the static 2-choice cuckoo table (two independent 16-byte
// loads; bpe/model.rs:252-275 defines the contents).
// =================================================================================================
// one-slot perfect-hash probe; `disp` may point to an LDS copy of t.merge_disp
__device__ __forceinline__ void merge_probe_d(const DevTables& t, const uint16_t* disp, uint32_t a, uint32_t b, uint32_t* rank, uint32_t* new_id) {
    uint32_t d = disp[merge_hash1(a, b, t.merge_seed) & t.merge_bmask];
    uint4 x = ((const uint4*)t.merges)[ph_slot(merge_hash2(a, b, t.merge_seed), d, t.merge_mask)];
    bool hit = x.x == a && x.y == b;
    *rank = hit ? x.z : RANK_NONE;
    *new_id = hit ? x.w : 0u;
}
// two independent probes (the two new pairs a merge leaves): both displacements, then both slots -- the two loads are in flight together
__device__ __forceinline__ void merge_probe2_d(const DevTables& t, const uint16_t* disp, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t* rank1, uint32_t* rank2) {
    const uint32_t d1 = disp[merge_hash1(a1, b1, t.merge_seed) & t.merge_bmask], d2 = disp[merge_hash1(a2, b2, t.merge_seed) & t.merge_bmask];
    const uint4* const m = (const uint4*)t.merges;
    const uint32_t s1 = ph_slot(merge_hash2(a1, b1, t.merge_seed), d1, t.merge_mask), s2 = ph_slot(merge_hash2(a2, b2, t.merge_seed), d2, t.merge_mask);
    // (left alone the compiler sinks the second symbol's select chain -- and with it the second probe -- below the first probe's wait)
    __builtin_amdgcn_sched_barrier(0);
    const uint4 x1 = m[s1], x2 = m[s2];
    __builtin_amdgcn_sched_barrier(0);
    *rank1 = (x1.x == a1 && x1.y == b1) ? x1.z : RANK_NONE;
    *rank2 = (x2.x == a2 && x2.y == b2) ? x2.z : RANK_NONE;
}
__device__ __forceinline__ void merge_probe(const DevTables& t, uint32_t a, uint32_t b, uint32_t* rank, uint32_t* new_id) {
    merge_probe_d(t, t.merge_disp, a, b, rank, new_id);
}

template <int G>
__global__ __launch_bounds__(256) void k_bpe_merge(DevTables t, const uint8_t* __restrict__ text, QView v, uint4* __restrict__ rows,
                                                   uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end) {
    constexpr int GPW = 64 / G;                                     // pre-tokens per wavefront
    __shared__ uint32_t s_qpre[NSQ + 1];
    const int lane = lane_id();
    const int sub = lane / G, c = lane % G, gbase = sub * G;
    const uint32_t n = qview_prefix(v, s_qpre);
    const uint32_t wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t base = wave_global * GPW; base < n; base += n_waves * GPW) {
        uint32_t item = base + sub;
        bool valid = item < n;
        uint32_t s = 0, len = 0, pos = 0, lenw = 0;            // lenw: the entry's length word (QLEN_CLAIM: it publishes its row)
        if (valid) { pos = qview_pos(s_qpre, v.sq_cap, item); const QItem it = v.q[pos]; s = it.s; lenw = it.len; len = qitem_len(lenw); }
        valid = valid && len != 0u;                                 // length 0: retired by k_long_vocab
        bool act = (uint32_t)c < len;
        uint32_t id = act ? t.byte_id[text[s + c]] : 0xFFFFFFFFu;
        uint64_t am = (len >= 64) ? ~0ull : ((1ull << len) - 1ull);   // alive symbols of my pre-token
        uint32_t rank = RANK_NONE, new_id = 0;
        {
            uint32_t nid = (uint32_t)__shfl((int)id, gbase + ((c + 1) % G), 64);
            if ((uint32_t)(c + 1) < len) merge_probe(t, id, nid, &rank, &new_id);
        }
        while (true) {
            uint32_t key = (rank == RANK_NONE) ? 0xFFFFFFFFu : ((rank << 6) | (uint32_t)c);
            uint32_t mn = (G == 16) ? row16_allmin(key) : wave_allmin(key);
            bool has = mn != 0xFFFFFFFFu;
            if (!__any(has)) break;
            uint32_t wpos = mn & 63u;
            uint32_t npos = 0;
            if (has) {
                uint64_t rest = am >> (wpos + 1);                       // winner always has a live right neighbour
                npos = wpos + 1 + (uint32_t)(__ffsll((unsigned long long)rest) - 1);
                am &= ~(1ull << npos);
                if ((uint32_t)c == wpos) id = new_id;                   // left symbol takes the merged id (word.rs:208)
                if ((uint32_t)c == npos) rank = RANK_NONE;              // right symbol is removed (word.rs:210)
            }
            // id of my next live symbol (after this round's removal)
            uint64_t mine = ((uint32_t)c + 1 < 64u) ? (am >> (c + 1)) : 0ull;
            bool has_next = mine != 0ull;
            uint32_t nx = has_next ? (uint32_t)c + 1 + (uint32_t)(__ffsll((unsigned long long)mine) - 1) : (uint32_t)c;
            uint32_t nid = (uint32_t)__shfl((int)id, gbase + (int)nx, 64);
            bool alive = (am >> c) & 1ull;
            // re-probe exactly the two pairs the reference pushes back: (prev, merged) and (merged, next)
            if (has && alive && ((uint32_t)c == wpos || (has_next && nx == wpos))) {
                if (has_next) merge_probe(t, id, nid, &rank, &new_id);
                else rank = RANK_NONE;
            }
        }
        // emit: token j of the pre-token = j-th live lane.  The first lane writes the result row (ids 1..3 come over by shuffle);
        // with more than four tokens the live lanes also leave ids 1.. in tmp_ids[s + j]
        const uint32_t count = (uint32_t)__popcll(am);
        uint32_t r1 = 0, r2 = 0, r3 = 0;
        {
            uint64_t m = am & ~1ull;
            const int p1 = m ? __ffsll((unsigned long long)m) - 1 : 0; m &= m - 1ull;
            const int p2 = m ? __ffsll((unsigned long long)m) - 1 : 0; m &= m - 1ull;
            const int p3 = m ? __ffsll((unsigned long long)m) - 1 : 0;
            r1 = (uint32_t)__shfl((int)id, gbase + p1, 64);
            r2 = (uint32_t)__shfl((int)id, gbase + p2, 64);
            r3 = (uint32_t)__shfl((int)id, gbase + p3, 64);
        }
        bool alive = act && ((am >> c) & 1ull);
        if (valid && alive) {
            uint32_t j = (uint32_t)__popcll(am & ((1ull << c) - 1ull));
            if (j == 0) { const uint4 row_ = make_row(count, s, id, r1, r2, r3); rows[v.row_base + pos] = row_; TKAMD_PUBLISH_ROW(t, text, s, lenw, row_); }
            else if (count > 4u) tmp_ids[s + j] = id;
            if (tmp_end) {
                uint64_t mine = ((uint32_t)c + 1 < 64u) ? (am >> (c + 1)) : 0ull;
                uint32_t endc = mine ? (uint32_t)c + 1 + (uint32_t)(__ffsll((unsigned long long)mine) - 1) : len;
                tmp_end[s + j] = endc;                                 // token end, bytes from the pre-token start
            }
        }
    }
}
template __global__ void k_bpe_merge<64>(DevTables, const uint8_t*, QView, uint4*, uint32_t*, uint32_t*);

// =================================================================================================
// K_bpe_merge_lane: the same merge loop for pre-tokens of <= 16 bytes, ONE LANE per pre-token with the
// whole Word (ids, pair ranks, pair new-ids) in registers -- 64 pre-tokens per wavefront instead of 4.
// The merge loop is a chain of dependent L2 round trips (probe -> min -> merge -> probe); giving every
// lane its own chain multiplies the memory-level parallelism by 16 over the row-per-word kernel.
// Arrays are indexed with compile-time indices only (selects), so nothing spills to scratch.
// Token boundaries travel as 16 nibbles (start byte of symbol i) for the offsets output.
// Same semantics as k_bpe_merge (models/bpe/word.rs:162-250): merge the (rank, position)-minimum pair,
// the left symbol takes new_id, re-probe the two new neighbours.
// =================================================================================================
// position of the k-th (0-based) set bit of m
__device__ __forceinline__ uint32_t select_bit32(uint32_t m, uint32_t k) {
    uint32_t pos = 0, c;
    c = __popc(m & 0xFFFFu); if (k >= c) { k -= c; pos += 16; m >>= 16; }
    c = __popc(m & 0xFFu);   if (k >= c) { k -= c; pos += 8;  m >>= 8; }
    c = __popc(m & 0xFu);    if (k >= c) { k -= c; pos += 4;  m >>= 4; }
    c = __popc(m & 0x3u);    if (k >= c) { k -= c; pos += 2;  m >>= 2; }
    c = m & 1u;              if (k >= c) { pos += 1; }
    return pos;
}

template <int S>   // S = 16 or 32 symbols per lane
__global__ __launch_bounds__(256) void k_bpe_merge_lane(DevTables t, const uint8_t* __restrict__ text, QView v, uint4* __restrict__ rows,
                                                        uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end) {
    __shared__ uint32_t s_qpre[NSQ + 1];
    constexpr uint32_t PB = (S == 16) ? 4 : 5;              // bits of the pair index inside the reduction key
    __shared__ uint32_t s_byte_id[256];
    __shared__ uint16_t s_disp[DISP_LDS_MAX];
    __shared__ uint32_t s_hist[S + 1];
    __shared__ uint4 s_sort[256];
    s_byte_id[threadIdx.x] = t.byte_id[threadIdx.x];
    const bool disp_in_lds = t.merge_bmask < (uint32_t)DISP_LDS_MAX;
    if (disp_in_lds)
        for (uint32_t i = threadIdx.x; i <= t.merge_bmask; i += 256) s_disp[i] = t.merge_disp[i];
    __syncthreads();
    const uint16_t* disp = disp_in_lds ? (const uint16_t*)s_disp : t.merge_disp;   // generic pointer: LDS or global
    const uint32_t n_items = qview_prefix(v, s_qpre);
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t base = blockIdx.x * 256; base < n_items; base += stride) {
        const uint32_t item = base + threadIdx.x;
        bool valid = item < n_items;
        uint32_t p = 0, s = 0, len = 0, claim = 0;           // p: queue position (names the result row); claim: QLEN_CLAIM of the entry
        if (valid) { p = qview_pos(s_qpre, v.sq_cap, item); const QItem it = v.q[p]; s = it.s; len = qitem_len(it.len); claim = it.len & QLEN_CLAIM; }
        valid = valid && len != 0u;                          // length 0: retired by k_long_vocab
        // The loop below runs until the slowest lane of a wavefront is done (~len - 2 rounds), so the 256 items of
        // this workgroup are counting-sorted by length first: each wavefront then holds one quartile of the lengths.
        {
            __syncthreads();
            if (threadIdx.x <= S) s_hist[threadIdx.x] = 0;
            __syncthreads();
            const uint32_t bin = valid ? len : (uint32_t)S;            // invalid lanes sort last (len <= S, so bin S is theirs + len == S)
            const uint32_t within = atomicAdd(&s_hist[bin], 1u);
            __syncthreads();
            uint32_t before = 0;
            for (uint32_t b = 0; b < bin; ++b) before += s_hist[b];
            const uint32_t slot = before + within;
            s_sort[slot] = make_uint4(p, s, len | claim, valid ? 1u : 0u);
            __syncthreads();
            const uint4 it = s_sort[threadIdx.x];
            p = it.x; s = it.y; len = qitem_len(it.z); claim = it.z & QLEN_CLAIM; valid = it.w != 0u;
        }
        uint64_t key[S / 8];
#pragma unroll
        for (int q = 0; q < S / 8; ++q) key[q] = 0;
        if (valid) {
            load_key16(text, s, min(len, 16u), &key[0], &key[1]);
            if (S == 32 && len > 16) load_key16(text, s + 16, len - 16, &key[S / 8 - 2], &key[S / 8 - 1]);
        }
        uint32_t ids[S], rk[S], nd[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            uint32_t b = (uint32_t)((key[i / 8] >> (8 * (i % 8))) & 0xFFu);
            ids[i] = s_byte_id[b];
            rk[i] = RANK_NONE;
            nd[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < S - 1; ++i)
            if ((uint32_t)(i + 1) < len) merge_probe_d(t, disp, ids[i], ids[i + 1], &rk[i], &nd[i]);
        uint32_t n = len;                                   // live symbols
        uint32_t starts = (len >= 32) ? 0xFFFFFFFFu : ((1u << len) - 1u);   // bit b: a symbol starts at byte b
        bool active = valid && len > 1;
        while (__any(active)) {
            if (active) {
                uint32_t best = 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < S - 1; ++i) {
                    uint32_t k = (rk[i] << PB) | (uint32_t)i;
                    if ((uint32_t)(i + 1) < n && rk[i] != RANK_NONE) best = min(best, k);
                }
                if (best == 0xFFFFFFFFu) active = false;
                else {
                    const uint32_t w = best & (uint32_t)(S - 1);
                    uint32_t new_id = 0;
#pragma unroll
                    for (int i = 0; i < S - 1; ++i) new_id = ((uint32_t)i == w) ? nd[i] : new_id;
                    // symbol w takes new_id, symbol w+1 disappears, everything right of it shifts left
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        uint32_t nxt_id = (i + 1 < S) ? ids[i + 1] : 0u;
                        ids[i] = ((uint32_t)i == w) ? new_id : (((uint32_t)i > w) ? nxt_id : ids[i]);
                    }
#pragma unroll
                    for (int i = 0; i < S - 1; ++i) {
                        uint32_t nr = (i + 1 < S - 1) ? rk[i + 1] : RANK_NONE, nn = (i + 1 < S - 1) ? nd[i + 1] : 0u;
                        if ((uint32_t)i >= w) { rk[i] = nr; nd[i] = nn; }
                    }
                    if (tmp_end) starts &= ~(1u << select_bit32(starts, w + 1));
                    n -= 1;
                    // re-probe (w-1, w) and (w, w+1)
                    uint32_t left = 0, right = 0;
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        left = ((uint32_t)i + 1 == w) ? ids[i] : left;
                        right = ((uint32_t)i == w + 1) ? ids[i] : right;
                    }
                    uint32_t r1 = RANK_NONE, n1 = 0, r2 = RANK_NONE, n2 = 0;
                    if (w > 0) merge_probe_d(t, disp, left, new_id, &r1, &n1);
                    if (w + 1 < n) merge_probe_d(t, disp, new_id, right, &r2, &n2);
#pragma unroll
                    for (int i = 0; i < S - 1; ++i) {
                        if ((uint32_t)i + 1 == w) { rk[i] = r1; nd[i] = n1; }
                        if ((uint32_t)i == w) { rk[i] = r2; nd[i] = n2; }
                    }
                    if (n < 2) active = false;
                }
            }
        }
        if (valid) {
            if (tmp_end && n >= 2u && n <= 4u) {               // (the token ends ride in the row: results.hip row_boundary; tmp_end is written as well here)
                uint32_t m = starts & (starts - 1u);
#pragma unroll
                for (int j = 1; j < 4; ++j)
                    if (m) { ids[j] |= row_boundary(text, s, (uint32_t)(__ffs(m) - 1), true); m &= m - 1u; }
            }
            { const uint4 row_ = make_row(n, s, ids[0], ids[1], ids[2], ids[3]); rows[v.row_base + p] = row_; TKAMD_PUBLISH_ROW(t, text, s, len | claim, row_); }
            if (n > 4u) {
#pragma unroll
                for (int j = 1; j < S; ++j)
                    if ((uint32_t)j < n) tmp_ids[s + j] = ids[j];
            }
            if (tmp_end) {
                uint32_t m = starts & (starts - 1u);          // drop the first start: ends are the later starts, then len
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    if ((uint32_t)j < n) {
                        uint32_t e = m ? (uint32_t)(__ffs(m) - 1) : len;
                        tmp_end[s + j] = e;
                        m &= m - 1u;
                    }
                }
            }
        }
    }
}
template __global__ void k_bpe_merge_lane<16>(DevTables, const uint8_t*, QView, uint4*, uint32_t*, uint32_t*);
template __global__ void k_bpe_merge_lane<32>(DevTables, const uint8_t*, QView, uint4*, uint32_t*, uint32_t*);

// =================================================================================================
// K_bpe_merge_lds: the lane-per-pre-token merge loop with the Word in LDS instead of registers.
// The register kernel above indexes its arrays with compile-time indices only, so every dynamic access is a
// 16- or 32-way select chain and one merge costs ~300 vector instructions; it is ALU-issue bound.  Here
//   * symbols stay IN PLACE: sym[i] is the symbol that starts at byte i, a 32-bit `alive` mask says which
//     positions still start a symbol (it doubles as the token-boundary mask for offsets), neighbours are
//     found with ctz/clz -- nothing shifts;
//   * key[i] = (rank << PB) | i for the pair (i, next alive), 0xFFFFFFFF when there is none: the
//     (rank, position)-minimum of word.rs:177-208 is one min3 tree over S LDS words, and the winner's
//     position and rank come out of the key itself;
//   * new_id = rank + constant (host-verified for the loaded vocabulary), so no new-id array is kept;
//   * sym/key live in LDS as [slot][thread] words: a lane only ever touches its own bank column, every
//     access is conflict-free, and a dynamic index costs one multiply-add.
// One merge is ~85 vector instructions.  LDS per lane is 8 * S bytes, which with the 32 KB displacement cache
// allows NT = 768 (S = 16) lanes per CU.
// Same semantics as k_bpe_merge / k_bpe_merge_lane (models/bpe/word.rs:162-250).
// =================================================================================================
// v2 (v2.q != nullptr): a second queue, taken after the first -- the 32-symbol kernel also serves the <= 16-byte class when the
// in-batch claims have thinned both queues to the distinct words: each launch then lasts as long as its longest word's chain of
// dependent merge probes whatever the queue holds, and one launch is cheaper than two.
// CHARS: BPE over CHARACTERS (no ByteLevel pre-tokenizer; the instantiation a tokenizer with t.cb & CB_ON runs).  Only the start differs:
// a symbol per char -- the vocabulary entry of the char with the affixes its place in the word glues on (t.char_id), an unk symbol
// (one per unknown char, or one per run of them: fuse_unk), the <0xXX> tokens of its bytes (byte_fallback), or nothing at all (no
// unk_token: the char is dropped) -- BPE::merge_word, bpe/model.rs:465-550.  Symbols still sit at the byte position they start at, so
// the merge loop, the token-boundary mask and the row format are the byte-level ones.
template <int S, int NT, bool DISP_LDS, bool SYM_REGS, bool CHARS = false>
__global__ __launch_bounds__(NT) void k_bpe_merge_lds(DevTables t, const uint8_t* __restrict__ text, QView v, QView v2, uint4* __restrict__ rows,
                                                      uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end) {
    __shared__ uint32_t s_qpre[NSQ + 1];
    __shared__ uint32_t s_qpre2[S == 32 ? NSQ + 1 : 1];    // (only the 32-symbol kernel takes a second queue: two 704-lane workgroups of the 16-symbol one fill a CU's LDS to the last KB)
    __shared__ uint32_t s_probes;                          // t.probes (profiling runs): merge-table probes of this workgroup's words, SURVEY 8d's model
    if (threadIdx.x == 0) s_probes = 0u;
    constexpr uint32_t PB = (S == 16) ? 4 : 5;
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) uint32_t, lds_words)
    uint32_t* s_key = lds_words;                              // [S][NT]
    uint32_t* s_sym = s_key + S * NT;                         // [S][NT], absent when the symbols stay in registers
    uint32_t* s_byte_id = s_sym + (SYM_REGS ? 0 : S * NT);    // [256]
    uint32_t* s_hist = s_byte_id + 256;                       // [S + 1] (+ padding to 64)
    uint16_t* s_disp = (uint16_t*)(s_hist + 64);              // [DISP_LDS_MAX]
    uint4* s_sort = (uint4*)s_key;                            // [NT], aliases the key area between items
    const uint32_t tid = threadIdx.x;
    // Which kernel takes the <= 16-byte class is decided HERE, from the queue's fill, when the host asks for it (t.thin_limit != 0: both
    // kernels are launched).  Thin -- the in-batch claims left the distinct words only -- it rides along in the 32-symbol launch (one
    // launch lasts as long as its longest word's chain of merges whatever it holds, and this kernel returns at once); fat -- text that
    // repeats nothing -- it stays with this kernel's twice as many lanes per CU (13 M short words: 0.85 ms against 1.51).  Both kernels
    // read the same counters, so exactly one of them takes the queue.
    const uint32_t n_first = qview_prefix(v, s_qpre);
    if (S == 16 && t.thin_limit && n_first < t.thin_limit) return;
    for (uint32_t i = tid; i < 256; i += NT) s_byte_id[i] = t.byte_id[i];
    // (DISP_LDS is the launcher's choice: only for a displacement table that fits.  A pointer that is the LDS copy or the table in memory
    // by a RUN-TIME test is a flat pointer: its loads count on both memory counters and return in no order, so the compiler waits for
    // everything in flight behind every one of them -- up to round 5 each of a word's up to 31 first probes, and both probes of every
    // merge, was a round trip of its own: s_waitcnt vmcnt(0) lgkmcnt(0) around each in the ISA)
    constexpr bool disp_in_lds = DISP_LDS;
    if (disp_in_lds) {
        // sixteen bytes a load (eight displacements): with thin queues the kernel is as long as its prologue plus one word's chain of
        // merges, and 2-byte loads made the prologue twenty dependent round trips per lane
        static_assert((DISP_LDS_MAX * 2) % 16 == 0, "whole 16-byte words");
        const uint32_t n16 = (t.merge_bmask + 8u) >> 3;                             // (the table is a power of two of entries; the upload is padded)
        for (uint32_t i = tid; i < n16; i += NT) ((uint4*)s_disp)[i] = ((const uint4*)t.merge_disp)[i];
    }
    __syncthreads();
    const uint16_t* const disp = disp_in_lds ? (const uint16_t*)s_disp : t.merge_disp;
    const uint32_t nid_base = t.newid_base;
    uint32_t n_second = (S == 32 && v2.q) ? qview_prefix(v2, s_qpre2) : 0u;
    if (t.thin_limit && n_second >= t.thin_limit) n_second = 0u;   // (fat: the 16-symbol kernel's)
    const uint32_t n_items = n_first + n_second;
    // a short queue (the in-batch claims leave the distinct words only) is spread over the whole grid, a few wavefronts of every
    // workgroup busy, instead of filling the first workgroups and leaving most CUs idle
    const uint32_t take = min((uint32_t)NT, ((n_items + gridDim.x - 1u) / gridDim.x + 63u) & ~63u);
    const uint32_t stride = gridDim.x * take;
    // Rounds alternate their direction over the workgroups: the queue of the LONGER class comes first in the item order, so it is the
    // first workgroups whose round 0 is as long as a 32-symbol word's chain of merges -- a second round on top of that made them the
    // launch's critical path (two rounds at C2: 296 k words on 196 k lanes); reversed, round 1 goes to the workgroups whose round 0 was short.
    uint32_t round = 0u;
    for (uint32_t base0 = 0u; base0 < n_items; base0 += stride, ++round) {
        const uint32_t base = base0 + ((round & 1u) ? (gridDim.x - 1u - blockIdx.x) : blockIdx.x) * take;
        if (base >= n_items) { if (base0 + stride >= n_items) break; continue; }         // (workgroup-uniform)
        const uint32_t item = base + tid;
        bool valid = tid < take && item < n_items;
        uint32_t s = 0, len = 0, qidx = 0, claim = 0;         // qidx: the result row (named by the position in the work queue); claim: QLEN_CLAIM of the entry
        if (valid) {
            if (item < n_first) { const uint32_t qp = qview_pos(s_qpre, v.sq_cap, item); const QItem it = v.q[qp]; s = it.s; len = qitem_len(it.len); claim = it.len & QLEN_CLAIM; qidx = v.row_base + qp; }
            else { const uint32_t qp = qview_pos(s_qpre2, v2.sq_cap, item - n_first); const QItem it = v2.q[qp]; s = it.s; len = qitem_len(it.len); claim = it.len & QLEN_CLAIM; qidx = v2.row_base + qp; }
        }
        valid = valid && len != 0u;                           // length 0: retired by k_long_vocab
        // counting sort of the workgroup's items by length: a wavefront loops until its slowest lane is done
        {
            __syncthreads();                                  // previous item's key/sym area is dead
            if (tid <= S) s_hist[tid] = 0;
            __syncthreads();
            const uint32_t bin = valid ? len : (uint32_t)S;
            const uint32_t within = atomicAdd(&s_hist[bin], 1u);
            __syncthreads();
            uint32_t before = 0;
            for (uint32_t b = 0; b < bin; ++b) before += s_hist[b];
            s_sort[before + within] = make_uint4(claim, s, len, valid ? qidx + 1u : 0u);
            __syncthreads();
            const uint4 it = s_sort[tid];
            __syncthreads();                                  // everyone has read its item before keys overwrite the area
            claim = it.x; s = it.y; len = it.z; valid = it.w != 0u;
            qidx = it.w - 1u;
        }
        uint32_t* my_key = s_key + tid;                       // slot i at my_key[i * NT]
        uint32_t* my_sym = s_sym + tid;
        // SYM_REGS: symbols stay in registers (a dynamic index is a select chain -- the ALU has the headroom) and only
        // the keys take LDS, which is what bounds the number of pre-tokens in flight per CU.  The selects are written
        // out at every use: taking the array's address (a lambda, a helper) would send it to scratch.
        uint32_t ids[S];
#define TKAMD_SYM_AT(dst, pos)                                                                 \
        do {                                                                                   \
            if (!SYM_REGS) (dst) = my_sym[(pos) * NT];                                         \
            else {                                                                             \
                (dst) = ids[0];                                                                \
                _Pragma("unroll") for (int q_ = 1; q_ < S; ++q_) (dst) = ((pos) == (uint32_t)q_) ? ids[q_] : (dst); \
            }                                                                                  \
        } while (0)
        uint32_t alive0 = 0u;                                  // the positions a symbol starts at
        {
            uint64_t kb[S / 8];
#pragma unroll
            for (int q = 0; q < S / 8; ++q) kb[q] = 0;
            if (valid) {
                load_key16(text, s, min(len, 16u), &kb[0], &kb[1]);
                if (S == 32 && len > 16) load_key16(text, s + 16, len - 16, &kb[S / 8 - 2], &kb[S / 8 - 1]);
            }
#define TKAMD_BYTE_AT(i_) ((i_) < S ? (uint32_t)((kb[((i_) < S ? (i_) : 0) / 8] >> (8 * (((i_) < S ? (i_) : 0) % 8))) & 0xFFu) : 0u)
            if (!CHARS) {
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    ids[i] = s_byte_id[TKAMD_BYTE_AT(i)];
                    if (!SYM_REGS) my_sym[i * NT] = ids[i];
                }
                // The first probes, one per pair of neighbours: EIGHT at a time, unconditionally (a lane whose word ends earlier probes the
                // pair of whatever its padding bytes map to and drops the answer) -- the eight loads are in flight together, where a probe
                // under `if (i + 1 < len)` that is also consumed there is a round trip of its own.  A group no word of the wavefront reaches
                // (the workgroup's words are sorted by length) is skipped as a whole: a scalar branch.
                uint32_t wmax = valid ? len : 0u;
#pragma unroll
                for (int d_ = 32; d_ >= 1; d_ >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d_, 64));
                wmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wmax);
                constexpr int PG = 8;
#pragma unroll
                for (int g = 0; g < S; g += PG) {
                    if ((uint32_t)(g + 1) < wmax) {
                        uint32_t rk[PG];
#pragma unroll
                        for (int q = 0; q < PG; ++q) {
                            const int i = g + q;
                            rk[q] = RANK_NONE;
                            if (i < S - 1) { uint32_t nd; merge_probe_d(t, disp, ids[i], ids[i + 1], &rk[q], &nd); }
                        }
#pragma unroll
                        for (int q = 0; q < PG; ++q) {
                            const int i = g + q;
                            my_key[i * NT] = ((uint32_t)(i + 1) < len && rk[q] != RANK_NONE) ? ((rk[q] << PB) | (uint32_t)i) : 0xFFFFFFFFu;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < PG; ++q) my_key[(g + q) * NT] = 0xFFFFFFFFu;
                    }
                }
                alive0 = (len >= 32) ? 0xFFFFFFFFu : ((1u << len) - 1u);
            } else {
                static_assert(!CHARS || SYM_REGS, "the char start keeps its symbols in registers");
                // 1. every char's own entry: S independent loads (a continuation byte asks for nothing)
                uint32_t own[S];
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const uint32_t b0 = TKAMD_BYTE_AT(i), b1 = TKAMD_BYTE_AT(i + 1) & 0x3Fu, b2 = TKAMD_BYTE_AT(i + 2) & 0x3Fu, b3 = TKAMD_BYTE_AT(i + 3) & 0x3Fu;
                    const bool lead = (uint32_t)i < len && (b0 & 0xC0u) != 0x80u;
                    const uint32_t cl = b0 < 0x80u ? 1u : b0 < 0xE0u ? 2u : b0 < 0xF0u ? 3u : 4u;
                    const uint32_t cp = b0 < 0x80u ? b0 : b0 < 0xE0u ? ((b0 & 0x1Fu) << 6) | b1 : b0 < 0xF0u ? ((b0 & 0x0Fu) << 12) | (b1 << 6) | b2
                                                                                                              : ((b0 & 0x07u) << 18) | (b1 << 12) | (b2 << 6) | b3;
                    const uint32_t var = ((i != 0 && (t.cb & CB_PREFIX)) ? 1u : 0u) | (((uint32_t)i + cl >= len && (t.cb & CB_SUFFIX)) ? 2u : 0u);
                    own[i] = (lead && cp < 0x110000u) ? t.char_id[(cp << 2) | var] : CHAR_NONE;
                }
                // 2. what stands where (in order: an unknown char's fate depends on the one before it)
                bool unknown = false, prev_unk = false;
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    ids[i] = 0u;
                    if ((uint32_t)i < len) {
                        const uint32_t b0 = TKAMD_BYTE_AT(i);
                        const bool lead = (b0 & 0xC0u) != 0x80u;
                        if (lead) unknown = own[i] == CHAR_NONE;
                        if (!unknown) { if (lead) { ids[i] = own[i]; alive0 |= 1u << i; prev_unk = false; } }
                        else if (t.cb & CB_BYTES) { ids[i] = s_byte_id[b0]; alive0 |= 1u << i; prev_unk = false; }
                        else if (lead) {
                            if (t.cb & CB_UNK) {
                                if (!(prev_unk && (t.cb & CB_FUSE))) { ids[i] = t.unk_id; alive0 |= 1u << i; }
                                prev_unk = true;
                            } else if (t.cb & CB_UNK_MISSING) atomicOr(t.err, ERR_UNK_OOV);
                        }                                      // (no unk_token at all: dropped without trace)
                    }
                }
                // 3. the rank of every symbol's pair with the next one standing
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    uint32_t k = 0xFFFFFFFFu;
                    const uint32_t above = (i < S - 1) ? (alive0 & ~((2u << i) - 1u)) : 0u;
                    if (((alive0 >> i) & 1u) && above) {
                        const uint32_t j = (uint32_t)__ffs(above) - 1u;
                        uint32_t right = ids[0];
#pragma unroll
                        for (int q_ = 1; q_ < S; ++q_) right = (j == (uint32_t)q_) ? ids[q_] : right;
                        uint32_t r, nd;
                        merge_probe_d(t, disp, ids[i], right, &r, &nd);
                        if (r != RANK_NONE) k = (r << PB) | (uint32_t)i;
                    }
                    my_key[i * NT] = k;
                }
            }
#undef TKAMD_BYTE_AT
        }
        uint32_t alive = alive0;
        bool active = valid && __popc(alive) > 1;
        while (__any(active)) {
            if (active) {
                uint32_t best = 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < S - 1; ++i) best = min(best, my_key[i * NT]);
                if (best == 0xFFFFFFFFu) active = false;
                else {
                    const uint32_t i = best & (uint32_t)(S - 1);
                    const uint32_t nid = (best >> PB) + nid_base;
                    uint32_t above = alive & ~((2u << i) - 1u);           // live positions right of i (the pair's right symbol is the first)
                    const uint32_t j = (uint32_t)__ffs(above) - 1u;
                    above &= above - 1u;
                    alive &= ~(1u << j);
                    const uint32_t below = alive & ((1u << i) - 1u);
                    const bool has_k = above != 0u, has_h = below != 0u;
                    const uint32_t k = has_k ? (uint32_t)__ffs(above) - 1u : i;
                    const uint32_t h = has_h ? 31u - (uint32_t)__clz(below) : i;
                    uint32_t sr, sl;
                    TKAMD_SYM_AT(sr, k);
                    TKAMD_SYM_AT(sl, h);
                    if (SYM_REGS) {
#pragma unroll
                        for (int q = 0; q < S; ++q) ids[q] = (i == (uint32_t)q) ? nid : ids[q];
                    } else my_sym[i * NT] = nid;
                    uint32_t r1, r2;
                    merge_probe2_d(t, disp, sl, nid, nid, sr, &r1, &r2);
                    my_key[j * NT] = 0xFFFFFFFFu;
                    my_key[i * NT] = (has_k && r2 != RANK_NONE) ? ((r2 << PB) | i) : 0xFFFFFFFFu;
                    if (has_h) my_key[h * NT] = (r1 != RANK_NONE) ? ((r1 << PB) | h) : 0xFFFFFFFFu;
                    if (!has_k && !has_h) active = false;                 // one symbol left
                }
            }
        }
        if (valid) {
            // result row named by the queue position (the lookup kernel already pointed tok0 at it); beyond four tokens the
            // ids 1.. go to tmp_ids[s + j]
            const uint32_t c = (uint32_t)__popc(alive);
            uint32_t r[4] = {ids[0], 0u, 0u, 0u};
            if (!SYM_REGS) r[0] = my_sym[0];
            uint32_t m = alive & ~1u;
            if (CHARS) {                                       // (a dropped char may leave position 0 empty -- or the whole word)
                m = alive;
                r[0] = 0u;
                if (m) { const uint32_t p0 = (uint32_t)__ffs(m) - 1u; TKAMD_SYM_AT(r[0], p0); m &= m - 1u; }
            }
            // (byte-level words of <= 4 tokens: the token ends ride in the row, results.hip row_boundary -- nothing goes to tmp_end)
            const bool carry = !CHARS && tmp_end != nullptr && c <= 4u;
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                if (m) {
                    const uint32_t pos = (uint32_t)__ffs(m) - 1u;
                    TKAMD_SYM_AT(r[j], pos);
                    if (carry) r[j] |= row_boundary(text, s, pos, true);
                    else if (tmp_end) tmp_end[s + j - 1] = pos;
                    m &= m - 1u;
                }
            }
            if (m) {
                tmp_ids[s + 1] = r[1]; tmp_ids[s + 2] = r[2]; tmp_ids[s + 3] = r[3];
                for (uint32_t j = 4; m; m &= m - 1u, ++j) {
                    const uint32_t pos = (uint32_t)__ffs(m) - 1u;
                    uint32_t v_;
                    TKAMD_SYM_AT(v_, pos);
                    tmp_ids[s + j] = v_;
                    if (tmp_end) tmp_end[s + j - 1] = pos;
                }
            }
            if (tmp_end && c && !carry) tmp_end[s + c - 1] = len;
            { const uint4 row_ = make_row(c, s, r[0], r[1], r[2], r[3]); rows[qidx] = row_; TKAMD_PUBLISH_ROW(t, text, s, len | claim, row_); }
            // (k - 1) + 2 m probes for a word of k symbols and m merges: every initial pair once, two new pairs per merge
            if (t.probes) { const uint32_t k0 = (uint32_t)__popc(alive0); if (k0) atomicAdd(&s_probes, (k0 - 1u) + 2u * (k0 - c)); }
        }
    }
    if (t.probes) {                                        // (uniform)
        __syncthreads();
        if (threadIdx.x == 0 && s_probes) atomicAdd(t.probes, s_probes);
    }
}
#undef TKAMD_SYM_AT
constexpr int lds_merge_bytes(int S, int NT, bool disp_lds, bool sym_regs) { return ((sym_regs ? 1 : 2) * S * NT + 256 + 64) * 4 + (disp_lds ? DISP_LDS_MAX * 2 : 0); }
template <int S, int NT, bool DISP_LDS, bool SYM_REGS>
static int prepare_lds_merge() {
    int rc = (int)hipFuncSetAttribute((const void*)k_bpe_merge_lds<S, NT, DISP_LDS, SYM_REGS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_merge_bytes(S, NT, DISP_LDS, SYM_REGS));
    if (rc == 0) rc = (int)hipFuncSetAttribute((const void*)k_bpe_merge_lds<S, NT, DISP_LDS, SYM_REGS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_merge_bytes(S, NT, DISP_LDS, SYM_REGS));
    return rc;
}
template <int S, int NT, bool DISP_LDS, bool SYM_REGS>
static void launch_lds_merge(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v, const QView& v2, uint4* rows, uint32_t* tmp_ids, uint32_t* tmp_end) {
    if (t.cb & CB_ON)
        hipLaunchKernelGGL((k_bpe_merge_lds<S, NT, DISP_LDS, SYM_REGS, true>), dim3(grid), dim3(NT), lds_merge_bytes(S, NT, DISP_LDS, SYM_REGS), st, t, text, v, v2, rows, tmp_ids, tmp_end);
    else
        hipLaunchKernelGGL((k_bpe_merge_lds<S, NT, DISP_LDS, SYM_REGS, false>), dim3(grid), dim3(NT), lds_merge_bytes(S, NT, DISP_LDS, SYM_REGS), st, t, text, v, v2, rows, tmp_ids, tmp_end);
}

// =================================================================================================
// K_bpe_merge_long: pre-tokens longer than 64 bytes, one workgroup each, symbols as a doubly linked
// list in LDS (the same Symbol{c, prev, next, len} of models/bpe/word.rs:38-54), up to LONG_PT_MAX
// symbols.  Each round: workgroup-wide min over the cached (rank, pos) keys, one merge, two
// re-probes.  Rare path (long letter/digit runs); exactness over speed.
// =================================================================================================
__global__ __launch_bounds__(256) void k_bpe_merge_long(DevTables t, const uint8_t* __restrict__ text,
                                                        QView v, uint4* __restrict__ rows,
                                                        uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end,
                                                        uint32_t* __restrict__ list_huge, uint32_t* __restrict__ n_huge) {
    __shared__ uint32_t s_qpre[NSQ + 1];
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) uint8_t, lds_raw)
    uint32_t* sym = (uint32_t*)lds_raw;                 // [LONG_PT_MAX]
    uint32_t* rnk = sym + LONG_PT_MAX;                     // [LONG_PT_MAX] rank of pair (i, next[i]) or NONE
    uint32_t* nid = rnk + LONG_PT_MAX;                     // [LONG_PT_MAX] new id of that pair
    uint16_t* nxt = (uint16_t*)(nid + LONG_PT_MAX);        // [LONG_PT_MAX] 0xFFFF = none
    uint16_t* prv = nxt + LONG_PT_MAX;                     // [LONG_PT_MAX]
    __shared__ unsigned long long red[4];
    __shared__ uint32_t cnt_s;
    const int tid = (int)threadIdx.x;
    __shared__ uint32_t first_s;
    const uint32_t n = qview_prefix(v, s_qpre);
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        const uint32_t pos = qview_pos(s_qpre, v.sq_cap, item);
        const QItem it = v.q[pos];
        const uint32_t s = it.s, len = qitem_len(it.len);
        if (len == 0u) continue;                           // retired by k_long_vocab
        if (len > (uint32_t)LONG_PT_MAX) {                 // too long for LDS: hand over to k_bpe_merge_huge (by queue position)
            if (tid == 0) {
                if (t.cb & CB_ON) atomicOr(t.err, ERR_PRETOKEN_TOO_LONG);      // (BPE over characters: no global-scratch variant -- a "word" of more than 8 KB)
                else list_huge[atomicAdd(n_huge, 1u)] = pos;
            }
            if (t.cb & CB_ON) {                            // (uniform) an empty row keeps the compaction well defined until the error is reported
                if (tid == 0) rows[v.row_base + pos] = make_uint4(ROW_CNT_MORE << ROW_CNT_SHIFT, s, 0u, 0u);
            }
            continue;
        }
        __syncthreads();
        if (!(t.cb & CB_ON)) {                             // (uniform) byte-level: a symbol per byte
            for (uint32_t i = tid; i < len; i += 256) {
                sym[i] = t.byte_id[text[s + i]];
                nxt[i] = (i + 1 < len) ? (uint16_t)(i + 1) : (uint16_t)0xFFFF;
                prv[i] = (i > 0) ? (uint16_t)(i - 1) : (uint16_t)0xFFFF;
            }
        } else {
            // BPE over characters (merge_word, bpe/model.rs:465-550; the same start as k_bpe_merge_lds<.., CHARS>): every lead byte asks
            // for its char's entry -- in parallel, parked in nid[] --, then ONE thread links up what stands where: an unknown char's fate
            // (its own unk symbol, part of the previous one, its bytes' tokens, nothing) depends on the char before it
            for (uint32_t i = tid; i < len; i += 256) {
                const uint32_t b0 = text[s + i];
                uint32_t own = CHAR_NONE;
                if ((b0 & 0xC0u) != 0x80u) {
                    const uint32_t cl = b0 < 0x80u ? 1u : b0 < 0xE0u ? 2u : b0 < 0xF0u ? 3u : 4u;
                    uint32_t cp = b0 < 0x80u ? b0 : b0 < 0xE0u ? (b0 & 0x1Fu) : b0 < 0xF0u ? (b0 & 0x0Fu) : (b0 & 0x07u);
                    for (uint32_t q = 1; q < cl; ++q) cp = (cp << 6) | (i + q < len ? (text[s + i + q] & 0x3Fu) : 0u);
                    const uint32_t var = ((i != 0 && (t.cb & CB_PREFIX)) ? 1u : 0u) | ((i + cl >= len && (t.cb & CB_SUFFIX)) ? 2u : 0u);
                    if (cp < 0x110000u) own = t.char_id[(cp << 2) | var];
                }
                nid[i] = own;
                sym[i] = 0xFFFFFFFFu;                      // nothing stands here (yet)
                nxt[i] = prv[i] = (uint16_t)0xFFFF;
                rnk[i] = RANK_NONE;
            }
            __syncthreads();
            if (tid == 0) {
                bool unknown = false, prev_unk = false;
                uint32_t last = 0xFFFFu;
                for (uint32_t i = 0; i < len; ++i) {
                    const uint32_t b0 = text[s + i];
                    const bool lead = (b0 & 0xC0u) != 0x80u;
                    if (lead) unknown = nid[i] == CHAR_NONE;
                    uint32_t id = 0xFFFFFFFFu;
                    if (!unknown) { if (lead) { id = nid[i]; prev_unk = false; } }
                    else if (t.cb & CB_BYTES) { id = t.byte_id[b0]; prev_unk = false; }
                    else if (lead) {
                        if (t.cb & CB_UNK) {
                            if (!(prev_unk && (t.cb & CB_FUSE))) id = t.unk_id;
                            prev_unk = true;
                        } else if (t.cb & CB_UNK_MISSING) atomicOr(t.err, ERR_UNK_OOV);
                    }
                    if (id != 0xFFFFFFFFu) {
                        sym[i] = id;
                        prv[i] = (uint16_t)last;
                        if (last != 0xFFFFu) nxt[last] = (uint16_t)i;
                        last = i;
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < len; i += 256) {
            uint32_t r = RANK_NONE, ni = 0;
            const uint32_t j = nxt[i];
            if (sym[i] != 0xFFFFFFFFu && j != 0xFFFFu) merge_probe(t, sym[i], sym[j], &r, &ni);
            rnk[i] = r;
            nid[i] = ni;
        }
        __syncthreads();
        while (true) {
            unsigned long long best = ~0ull;
            for (uint32_t i = tid; i < len; i += 256) {
                uint32_t r = rnk[i];
                if (r != RANK_NONE) {
                    unsigned long long k = ((unsigned long long)r << 32) | i;
                    best = k < best ? k : best;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                unsigned long long o = __shfl_xor(best, d, 64);
                best = o < best ? o : best;
            }
            if ((tid & 63) == 0) red[tid >> 6] = best;
            __syncthreads();
            unsigned long long m01 = red[0] < red[1] ? red[0] : red[1];
            unsigned long long m23 = red[2] < red[3] ? red[2] : red[3];
            best = m01 < m23 ? m01 : m23;
            __syncthreads();
            if (best == ~0ull) break;
            if (tid == 0) {
                uint32_t w = (uint32_t)best;
                uint32_t r = nxt[w];
                uint32_t rn = nxt[r];
                sym[w] = nid[w];
                rnk[r] = RANK_NONE;
                sym[r] = 0xFFFFFFFFu;                          // dead
                nxt[w] = (uint16_t)rn;
                if (rn != 0xFFFFu) prv[rn] = (uint16_t)w;
                uint32_t pw = prv[w];
                uint32_t r1 = RANK_NONE, n1 = 0, r2 = RANK_NONE, n2 = 0;
                if (pw != 0xFFFFu) merge_probe(t, sym[pw], sym[w], &r1, &n1);
                if (rn != 0xFFFFu) merge_probe(t, sym[w], sym[rn], &r2, &n2);
                if (pw != 0xFFFFu) { rnk[pw] = r1; nid[pw] = n1; }
                rnk[w] = r2;
                nid[w] = n2;
            }
            __syncthreads();
        }
        // emit in order: walk is sequential per symbol; do a parallel rank instead
        if (tid == 0) cnt_s = 0;
        __syncthreads();
        // chunked ordered compaction: 256 symbols per step
        for (uint32_t base = 0; base < len; base += 256) {
            uint32_t i = base + tid;
            bool alive = i < len && sym[i] != 0xFFFFFFFFu;
            uint64_t bm = __ballot(alive);
            __shared__ uint32_t wcnt[4];
            if ((tid & 63) == 0) wcnt[tid >> 6] = (uint32_t)__popcll(bm);
            __syncthreads();
            uint32_t off = cnt_s;
            for (int w = 0; w < (tid >> 6); ++w) off += wcnt[w];
            if (alive) {
                uint32_t j = off + (uint32_t)mbcnt64(bm);
                if (j == 0) first_s = sym[i];
                else tmp_ids[s + j] = sym[i];
                if (tmp_end) {                                 // (the end of a token = where the next one starts: k_token_meta's convention)
                    uint32_t e = nxt[i];
                    tmp_end[s + j] = (e == 0xFFFFu) ? len : e;
                }
            }
            __syncthreads();
            if (tid == 0) cnt_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        // long pre-tokens always use the "ids 1.. in tmp_ids" row form, whatever their count
        if (tid == 0) {
            const uint4 row_ = make_uint4((cnt_s ? first_s : 0u) | (ROW_CNT_MORE << ROW_CNT_SHIFT), s, cnt_s, 0u);
            rows[v.row_base + pos] = row_;
            TKAMD_PUBLISH_ROW(t, text, s, it.len, row_);   // (short words come here when the LDS kernels cannot take them: BPE over characters whose new ids are not in merge order)
        }
    }
}
