// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Whole-word lookup straight from the pre-tokenizer's bitmasks.

// =================================================================================================
// K_lookup: every pre-token -> its token id if the whole pre-token settles it, else a place in a work queue.
// Replaces, per pre-token:
//   * BPE::tokenize_with_cache's shortcuts (models/bpe/model.rs:558-587): ignore_merges' vocab.get(sequence) (:559-567, exact)
//     and the per-thread word cache (:573-586) -- here a STATIC table of vocabulary entries whose own merge result was
//     verified at load time, by the device merge kernel, to be exactly [id] (WORD_DIRECT): a hit is provably what
//     merge_word returns; everything else is queued for the merge kernels by length class;
//   * WordLevel::tokenize (models/wordlevel/mod.rs:162-178): hit -> id, miss -> unk id or MissingUnkToken;
//   * the first candidate of WordPiece::tokenize (models/wordpiece/mod.rs:245-258 starts at end = len): the whole word;
//     misses are queued for the trie walk.
// and the Vec<Split> bookkeeping of pre_tokenizer.rs:73-103: the kernel reads the start (/ end) BITMASKS and never needs
// the pre-token offsets in global memory.
//
// One 512-lane workgroup per 16 KB tile of text (256 mask words), two workgroups per CU:
//   1. the tile's text (+ 64 bytes) is staged in LDS with wavefront-wide 16-byte loads;
//   2. one lane per mask word walks its set bits and drops the positions into an LDS array, indexed by the pre-token's
//      rank inside the tile (the global rank of the tile's first pre-token comes from the mask prefix sums);
//   3. wavefront w takes a contiguous range of ranks, lane l of step k the rank range_start + 64 k + l: neighbouring lanes
//      hold neighbouring pre-tokens, so the key reads hit neighbouring LDS words, the tok0 stores are one 256-byte line
//      per wavefront, and the work queues come out in text order (the merge kernels' accesses stay local);
//   4. the key (first <= 16 bytes, zero padded) is probed in an LDS copy of the HOT table -- the lowest-id (= most
//      frequent, the trainers append tokens in frequency order) settled words of <= 12 bytes, direct mapped, 16-byte slots
//      -- and only on a miss in the perfect-hash table in HBM (one displacement load + one 32-byte slot);
//   5. queue positions: ballots per step, one atomic per queue per workgroup and round -- on the fill counter of the
//      workgroup's own sub-queue (results.hip: same-address atomics serialise); the queue entry is (start, length),
//      the tok0 word of a queued pre-token names the row its result will be written to (results.hip).
// =================================================================================================
constexpr int LU_NT = 512;
constexpr int LU_WAVES = LU_NT / 64;
constexpr int LU_TILE_WORDS = 256;                       // mask words per tile
constexpr int LU_TILE = LU_TILE_WORDS * 64;              // bytes of text per tile (16 KB)
constexpr int LU_TEXT_SLACK = 64;                        // staged past the tile: a key may start at its last byte
constexpr int LU_POS_CAP = 4096;                         // pre-tokens expanded per round (more in a tile: another round)
constexpr int LU_STEPS = LU_POS_CAP / LU_NT;             // steps per lane and round (8)
constexpr int LU_GROUP = 4;                              // steps whose table probes are kept in flight together
constexpr int HOT_SLOTS = 2048;                          // 32 KB of LDS: two workgroups per CU overlap each other's load / probe / queue phases

struct LookupArgs {
    const uint8_t* text;
    int64_t n_bytes_host;            // host-side bound of the text length (buffer is readable to n_bytes_host + TEXT_PAD)
    const int64_t* len_dev;          // effective length when the text was derived on the device, else null
    const unsigned long long* startmask;
    const unsigned long long* endmask;   // "Removed" pre-tokenizers: explicit ends; null: a pre-token ends where the next starts
    const uint32_t* wprefix;         // #starts before each mask word
    uint32_t* tok0;
    QView v[4];                      // queues: <= 16 bytes, <= 32, <= 64, longer (sub-queue blockIdx % NSQ of each)
    int* err;
    const unsigned long long* matchmask;   // added-token matches: one pre-token, id patched in later
    const uint4* hot;                // [HOT_SLOTS] {k0, k1, k2, id | len << 24}, len 0 = empty
    const WordSlot* words;           // perfect-hash table behind the hot table
    const uint16_t* word_disp;
    uint32_t word_mask, word_seed, word_bmask;
    uint32_t any_hit_final;          // ignore_merges / WordLevel / WordPiece: every hit is final (else only WORD_DIRECT ones)
    uint32_t no_hits;                // WordPiece with max_input_chars_per_word < 16: every word takes the trie walk
    uint32_t miss_is_unk;            // WordLevel: a miss of <= 16 bytes is the unk id (or MissingUnkToken), never queued
    uint32_t unk_id, has_unk;
};

template <bool HAS_END>
__global__ __launch_bounds__(LU_NT) void k_lookup(LookupArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lu_lds[];
    uint4* s_hot = (uint4*)lu_lds;                                              // [HOT_SLOTS]
    uint32_t* s_text32 = (uint32_t*)(s_hot + HOT_SLOTS);                        // [(LU_TILE + LU_TEXT_SLACK) / 4 + 4]
    uint16_t* s_pos = (uint16_t*)(s_text32 + (LU_TILE + LU_TEXT_SLACK) / 4 + 4);    // [LU_POS_CAP + 2] start of rank r, relative to the tile
    uint16_t* s_end = s_pos + LU_POS_CAP + 2;                                   // [LU_POS_CAP + 2] explicit ends (0xFFFF: beyond the tile)
    __shared__ uint4 s_kmask[17];                                                // byte masks of a key of 0..16 bytes
    __shared__ uint32_t s_n, s_pbase, s_last_end;
    __shared__ uint32_t s_wcnt[LU_WAVES][4];
    __shared__ uint32_t s_qbase[4];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t sq = blockIdx.x & (uint32_t)(NSQ - 1);          // this workgroup's sub-queue
    for (int i = tid; i < HOT_SLOTS; i += LU_NT) s_hot[i] = a.hot[i];
    if (tid < 17) {
        const uint32_t l = (uint32_t)tid;
        auto m = [&](uint32_t lo) -> uint32_t { return l >= lo + 4u ? 0xFFFFFFFFu : (l > lo ? ((1u << (8u * (l - lo))) - 1u) : 0u); };
        s_kmask[tid] = make_uint4(m(0u), m(4u), m(8u), m(12u));
    }
    const int64_t n_bytes = a.len_dev ? *a.len_dev : a.n_bytes_host;
    const int64_t total_words = (n_bytes + 63) >> 6;                 // start bits exist only below n_bytes
    const int64_t end_words = (n_bytes >> 6) + 1;                    // an end bit can sit at byte n_bytes
    const int64_t n_tiles = (total_words + LU_TILE_WORDS - 1) / LU_TILE_WORDS;
    constexpr bool has_end = HAS_END;
    const uint32_t hit_mask = a.no_hits ? 0u : 0xFFFFFFFFu;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t w0 = tile * LU_TILE_WORDS;
        const int64_t t0 = w0 << 6;                                  // first byte of the tile
        __syncthreads();                                             // previous tile's LDS use is over
        // ---- 1. text tile -> LDS (16-byte loads at any alignment; zero past the readable end) ----
        {
            const int64_t readable = a.n_bytes_host + TEXT_PAD;
            for (int c = tid; c < (LU_TILE + LU_TEXT_SLACK) / 16; c += LU_NT) {
                const int64_t g = t0 + 16 * (int64_t)c;
                Unaligned16 v{0u, 0u, 0u, 0u};
                if (g + 16 <= readable) v = *(const Unaligned16*)(a.text + g);
                ((uint4*)s_text32)[c] = make_uint4(v.a, v.b, v.c, v.d);
            }
        }
        // ---- 2. the tile's mask words: local rank of each word's first start ----
        unsigned long long ms = 0ull, me = 0ull;
        uint32_t rbase = 0;
        if (tid < LU_TILE_WORDS) {
            const int64_t w = w0 + tid;
            if (w < total_words) ms = a.startmask[w];
            if (has_end && w < end_words) me = a.endmask[w];
            const uint32_t first = a.wprefix[w0];
            rbase = (w < total_words ? a.wprefix[w] : 0u) - first;
            if (tid == 0) s_pbase = first;
            if (w == total_words - 1 || (tid == LU_TILE_WORDS - 1 && w < total_words)) s_n = rbase + (uint32_t)__popcll(ms);
            if (w >= total_words) rbase = 0xFFFFFFFFu;
        }
        // end of the tile's LAST pre-token when it lies beyond the tile: the next start (or end bit) after the tile, found by
        // wavefront 0 walking the mask forward 64 words at a time (almost always the very first word)
        if (wave == 0) {
            const unsigned long long* mk = has_end ? a.endmask : a.startmask;
            const int64_t lim = has_end ? end_words : total_words;
            int64_t w = w0 + LU_TILE_WORDS;
            uint32_t found = (uint32_t)n_bytes;
            while (w < lim) {
                const unsigned long long m = (w + lane < lim) ? mk[w + lane] : 0ull;
                const uint64_t any = __ballot(m != 0ull);
                if (any) {
                    const int l = __ffsll((unsigned long long)any) - 1;
                    const unsigned long long mm = ((unsigned long long)(uint32_t)__shfl((int)(m >> 32), l, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)m, l, 64);
                    found = (uint32_t)(((w + l) << 6) + (__ffsll(mm) - 1));
                    break;
                }
                w += 64;
            }
            if (lane == 0) s_last_end = found;
        }
        __syncthreads();
        const uint32_t n = s_n;                                      // pre-tokens starting in this tile
        const uint32_t pbase = s_pbase;                              // global rank of the first one
        const uint32_t last_end = s_last_end;
        for (uint32_t rb = 0; rb < n; rb += LU_POS_CAP) {
            if (rb) __syncthreads();                                 // previous round has read s_pos / s_end
            const uint32_t cnt = min((uint32_t)LU_POS_CAP, n - rb);
            if (has_end) {
                for (int i = tid; i < LU_POS_CAP + 2; i += LU_NT) s_end[i] = 0xFFFFu;
                __syncthreads();
            }
            // ---- 2b. positions of the set bits, by rank (ranks rb .. rb + cnt, the extra one is the next start) ----
            if (tid < LU_TILE_WORDS && rbase != 0xFFFFFFFFu) {
                uint32_t r = rbase - rb;
                for (unsigned long long m = ms; m; m &= m - 1ull, ++r)
                    if (r <= cnt) s_pos[r] = (uint16_t)(tid * 64 + (__ffsll(m) - 1));
                if (has_end) {
                    for (unsigned long long m = me; m; m &= m - 1ull) {
                        const int b = __ffsll(m) - 1;
                        // the pre-token this end closes: the last start before it
                        const uint32_t before = rbase + (uint32_t)__popcll(ms & ((1ull << b) - 1ull));
                        if (before == 0u) continue;                  // closes a pre-token of an earlier tile
                        const uint32_t rel = before - 1u - rb;
                        if (rel < cnt) s_end[rel] = (uint16_t)(tid * 64 + b);
                    }
                }
            }
            __syncthreads();
            // ---- 3./4. lookup ----
            // wavefront w owns ranks [w * per_wave, (w + 1) * per_wave) of this round, per_wave a multiple of 64
            const uint32_t per_wave = (((cnt + LU_WAVES - 1) / LU_WAVES) + 63u) & ~63u;
            const uint32_t wlo = wave * per_wave;
            uint32_t keep_s[LU_STEPS], keep_len[LU_STEPS], keep_q[LU_STEPS];       // queued items: start, length, class << 28 | index inside the wavefront
            uint32_t wq0 = 0u, wq1 = 0u, wq2 = 0u, wq3 = 0u;                        // this wavefront's queue counts so far
#pragma unroll
            for (int k = 0; k < LU_STEPS; ++k) { keep_q[k] = 0u; keep_s[k] = 0u; keep_len[k] = 0u; }
#pragma unroll
            for (int g = 0; g < LU_STEPS; g += LU_GROUP) {
                if (64u * g >= per_wave) break;                                     // wavefront-uniform
                // LU_GROUP steps at a time: keys and hot-table probes of all of them, then all displacement loads, then all
                // slot loads -- the global probes of a group are in flight together instead of one round trip per step.
                // Straight-line integer code (the kernel is instruction-issue bound): an invalid lane carries length 0, which
                // nothing matches; `pend` is 1 while a pre-token is neither settled nor an added-token match.
                uint32_t s_abs[LU_GROUP], len[LU_GROUP], k0[LU_GROUP], k1[LU_GROUP], k2[LU_GROUP], k3[LU_GROUP], out[LU_GROUP], pend[LU_GROUP], hh[LU_GROUP];
#pragma unroll
                for (int j = 0; j < LU_GROUP; ++j) {
                    const uint32_t rel = wlo + 64u * (g + j) + lane;
                    const bool v = 64u * (g + j) < per_wave && rel < cnt;
                    const uint32_t relc = v ? rel : 0u;
                    const uint32_t s_rel = s_pos[relc];
                    uint32_t e_abs;
                    if (HAS_END) {
                        const uint32_t e_rel = s_end[relc];
                        e_abs = e_rel == 0xFFFFu ? last_end : (uint32_t)t0 + e_rel;
                    } else {
                        e_abs = rb + relc + 1u >= n ? last_end : (uint32_t)t0 + (uint32_t)s_pos[relc + 1u];
                    }
                    s_abs[j] = (uint32_t)t0 + s_rel;
                    len[j] = v ? e_abs - s_abs[j] : 0u;
                    // key: first min(len, 16) bytes from the LDS copy of the text, zero padded
                    const uint32_t wi = s_rel >> 2, sh = s_rel & 3u;
                    const uint32_t d0 = s_text32[wi], d1 = s_text32[wi + 1], d2 = s_text32[wi + 2], d3 = s_text32[wi + 3], d4 = s_text32[wi + 4];
                    const uint4 km = s_kmask[min(len[j], 16u)];
                    k0[j] = __builtin_amdgcn_alignbyte(d1, d0, sh) & km.x;
                    k1[j] = __builtin_amdgcn_alignbyte(d2, d1, sh) & km.y;
                    k2[j] = __builtin_amdgcn_alignbyte(d3, d2, sh) & km.z;
                    k3[j] = __builtin_amdgcn_alignbyte(d4, d3, sh) & km.w;
                    // hot table (LDS): settled words of <= 12 bytes (k3 is 0 for those; an empty slot has length 0)
                    hh[j] = hot_hash(k0[j], k1[j], k2[j], len[j], a.word_seed);
                    const uint4 h = s_hot[hh[j] & (uint32_t)(HOT_SLOTS - 1)];
                    const uint32_t diff = (h.x ^ k0[j]) | (h.y ^ k1[j]) | (h.z ^ k2[j]) | ((h.w >> 24) ^ len[j]) | k3[j];
                    const bool hit = diff == 0u && len[j] != 0u;
                    out[j] = hit ? ((TOK_ONE | (h.w & TOK_ID_MASK)) & hit_mask) : 0u;
                    pend[j] = (v && out[j] == 0u) ? 1u : 0u;
                }
                if (a.matchmask) {                                                  // wavefront-uniform: tokenizers with added tokens only
#pragma unroll
                    for (int j = 0; j < LU_GROUP; ++j)
                        if (len[j] && ((a.matchmask[s_abs[j] >> 6] >> (s_abs[j] & 63u)) & 1ull)) { pend[j] = 0u; out[j] = 0u; }
                }
                uint32_t h2[LU_GROUP], dsp[LU_GROUP], probe[LU_GROUP];
#pragma unroll
                for (int j = 0; j < LU_GROUP; ++j) {
                    probe[j] = (pend[j] && len[j] <= (uint32_t)WORD_MAX_KEY) ? hit_mask : 0u;      // all ones / zero
                    const uint32_t h1 = word_hash1_from_hot(hh[j], k3[j]);
                    h2[j] = word_hash2(h1);
                    dsp[j] = (uint32_t)a.word_disp[h1 & a.word_bmask & probe[j]];                // (non-probing lanes share address 0)
                }
                uint4 a0[LU_GROUP], a1[LU_GROUP];
#pragma unroll
                for (int j = 0; j < LU_GROUP; ++j) {
                    const uint4* q = (const uint4*)&a.words[ph_slot(h2[j], dsp[j], a.word_mask) & probe[j]];
                    a0[j] = q[0]; a1[j] = q[1];
                }
#pragma unroll
                for (int j = 0; j < LU_GROUP; ++j) {
                    const int k = g + j;
                    const uint32_t rel = wlo + 64u * k + lane;
                    const uint32_t diff = (a0[j].x ^ k0[j]) | (a0[j].y ^ k1[j]) | (a0[j].z ^ k2[j]) | (a0[j].w ^ k3[j]) | (a1[j].x ^ len[j]);
                    const bool hit = probe[j] && diff == 0u && (a.any_hit_final | (a1[j].z & WORD_DIRECT));
                    if (hit) { out[j] = TOK_ONE | a1[j].y; pend[j] = 0u; }
                    if (a.miss_is_unk) {                                            // wavefront-uniform: WordLevel (wordlevel/mod.rs:170-177)
                        if (pend[j] && len[j] <= (uint32_t)WORD_MAX_KEY) {
                            if (a.has_unk) out[j] = TOK_ONE | a.unk_id;
                            else atomicOr(a.err, ERR_MISSING_UNK);
                            pend[j] = 0u;
                        }
                    }
                    // still pending: a model kernel's work.  Queue 0 takes <= 16 bytes; the longer classes are rare
                    const bool q1 = pend[j] && len[j] <= 16u;
                    const uint64_t b1 = __ballot(q1);
                    uint32_t kq = q1 ? ((1u << 28) | (wq0 + (uint32_t)mbcnt64(b1))) : 0u;
                    wq0 += (uint32_t)__popcll(b1);
                    const bool qu = pend[j] && len[j] > 16u;
                    if (__ballot(qu)) {                                             // wavefront-uniform, rare
                        const uint32_t c = len[j] <= 32u ? 2u : (len[j] <= 64u ? 3u : 4u);
                        const uint64_t b2 = __ballot(qu && c == 2u), b3 = __ballot(qu && c == 3u), b4 = __ballot(qu && c == 4u);
                        if (qu) kq = (c << 28) | (c == 2u ? wq1 + (uint32_t)mbcnt64(b2) : c == 3u ? wq2 + (uint32_t)mbcnt64(b3) : wq3 + (uint32_t)mbcnt64(b4));
                        wq1 += (uint32_t)__popcll(b2);
                        wq2 += (uint32_t)__popcll(b3);
                        wq3 += (uint32_t)__popcll(b4);
                    }
                    keep_q[k] = kq;
                    keep_s[k] = s_abs[j];
                    keep_len[k] = len[j];
                    if (len[j] && !kq) a.tok0[pbase + rb + rel] = out[j];            // settled (or an added-token placeholder)
                }
            }
            // ---- 5. queue space: one atomic per queue for the whole workgroup, wavefronts in order ----
            if (lane < 4) s_wcnt[wave][lane] = lane == 0 ? wq0 : lane == 1 ? wq1 : lane == 2 ? wq2 : wq3;
            __syncthreads();
            if (tid < 4) {
                uint32_t tot = 0;
                for (int w = 0; w < LU_WAVES; ++w) { const uint32_t x = s_wcnt[w][tid]; s_wcnt[w][tid] = tot; tot += x; }
                uint32_t* const cnt_p = (tid == 0 ? a.v[0].counts : tid == 1 ? a.v[1].counts : tid == 2 ? a.v[2].counts : a.v[3].counts) + sq * QCNT_STRIDE;
                s_qbase[tid] = tot ? atomicAdd(cnt_p, tot) : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < LU_STEPS; ++k) {
                const uint32_t kq = keep_q[k];
                if (kq) {
                    const uint32_t c = (kq >> 28) - 1u;
                    const uint32_t pos = s_qbase[c] + s_wcnt[wave][c] + (kq & 0x0FFFFFFFu);
                    const uint32_t rel = wlo + 64u * k + lane;
                    // (selects, not indexed loads: the argument arrays must stay in scalar registers)
                    QItem* const qp = c == 0u ? a.v[0].q : c == 1u ? a.v[1].q : c == 2u ? a.v[2].q : a.v[3].q;
                    const uint32_t qcap = c == 0u ? a.v[0].sq_cap : c == 1u ? a.v[1].sq_cap : c == 2u ? a.v[2].sq_cap : a.v[3].sq_cap;
                    const uint32_t rowb = c == 0u ? a.v[0].row_base : c == 1u ? a.v[1].row_base : c == 2u ? a.v[2].row_base : a.v[3].row_base;
                    if (pos < qcap) {
                        qp[sq * qcap + pos] = QItem{keep_s[k], keep_len[k]};
                        a.tok0[pbase + rb + rel] = TOK_ROW | (rowb + sq * qcap + pos);
                    } else {
                        atomicOr(a.err, ERR_QUEUE_FULL);                             // the host grows the queues and runs the batch again
                        a.tok0[pbase + rb + rel] = 0u;
                    }
                }
            }
        }
    }
}

constexpr int lookup_lds_bytes(bool has_end) { return HOT_SLOTS * 16 + (LU_TILE + LU_TEXT_SLACK) + 16 + (has_end ? 2 : 1) * (LU_POS_CAP + 2) * 2; }
