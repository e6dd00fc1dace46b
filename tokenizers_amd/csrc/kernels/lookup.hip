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
// One 512-lane workgroup per 16 KB tile of text (256 mask words), three (or two) workgroups per CU.  Passes 1 and the staging are bound
// by vector-ALU issue (a wave64 instruction occupies its SIMD16 for four cycles; all operands sit in LDS), so they are built to spend
// few instructions per pre-token; passes 2 and 3 -- two thirds of the kernel's time -- by the random lines they fetch from beyond the
// L2 (round 4's sessions i-m, tables.hpp): one per probe of the short-word table, two or three per claim.
//   1. the tile's text (+ 64 bytes), its mask words and their prefix counts are loaded into registers one tile AHEAD and
//      dropped into LDS when the tile starts;
//   2. one lane per HALF mask word walks its set bits and drops the positions into an LDS array, indexed by the pre-token's
//      rank inside the tile (the global rank of the tile's first pre-token comes from the mask prefix sums);
//   3. PASS 1, every pre-token: lane l of a wavefront takes rank 64 c + l of chunk c (chunks dealt round robin to the eight
//      wavefronts, so the tok0 stores are one 256-byte line per wavefront); the first <= 12 bytes of the key are probed in an LDS
//      copy of the HOT table -- the lowest-id (= most frequent, the trainers append tokens in frequency order) settled words of
//      <= 12 bytes, hash-and-displace, 16-byte slots.  A hit stores its id; a miss costs one more LDS store: its rank goes to the
//      workgroup's miss list;
//   4. PASS 2, the misses only, densely packed 64 to a step (four to five tenths of the pre-tokens on natural text): the full
//      16-byte key in the short-word table in HBM (tables.hpp: one 16-byte slot behind an 8-bit displacement); what still misses is queued by
//      length class: every workgroup appends to its OWN sub-queue of each queue (results.hip) -- the position comes from an
//      LDS counter, no global atomic -- the queue entry is (start, length), and
//      the tok0 word of a queued pre-token names the row its result will be written to (results.hip).
// =================================================================================================
constexpr int LU_NT = 512;
constexpr int LU_WAVES = LU_NT / 64;
constexpr int LU_TILE_WORDS = 256;                       // mask words per tile
constexpr int LU_TILE = LU_TILE_WORDS * 64;              // bytes of text per tile (16 KB)
static_assert(LU_TILE == LOOKUP_TILE_BYTES, "the host sizes the sub-queues per tile (capi.cpp queue_sizes)");
constexpr int LU_TEXT_SLACK = 64;                        // staged past the tile: a key may start at its last byte
// The shape: a hot-word table of 1,024 slots (16.5 KB), 3,072 pre-tokens expanded per round (a tile of prose holds ~2,700): 52 KB of LDS and
// <= 80 VGPRs, THREE workgroups per CU -- passes 2 and 3 wait for memory, and more wavefronts hide more of it; the short-word
// displacements are read from memory (a hot 8 KB array), pass 2 takes one step at a time.  (Rounds 3-5 also shipped a two-workgroup shape
// with 2,048 hot slots and the displacements in LDS -- 0.2279 against 0.2237 ms on C2, slower on C3 -- and a FUSED shape with the
// pre-tokenizer and the mask scan inside this kernel -- parity-green, 0.474 ms against 0.285 for the three kernels; both were deleted in
// round 6, HISTORY.md has their measurements.)
constexpr int HOT = HOT_SLOTS;                           // hot-word slots (tables.hpp)
constexpr int LU_POS_CAP = 3072;                         // pre-tokens expanded per round (more in a tile: another round)
constexpr int LU_WAVES_PER_SIMD = 6;                     // LU_NT / 64 wavefronts a workgroup, four SIMDs a CU: three workgroups (<= 80 VGPRs)
constexpr uint32_t CLAIM_ADAPT_MIN = 768u;               // candidates a workgroup looks at before it judges the claims' yield (about two tiles of prose)

struct LookupArgs {
    const uint8_t* text;
    int64_t n_bytes_host;            // host-side bound of the text length (buffer is readable to n_bytes_host + TEXT_PAD)
    const int64_t* len_dev;          // effective length when the text was derived on the device, else null
    const unsigned long long* startmask;
    const unsigned long long* endmask;   // "Removed" pre-tokenizers: explicit ends; null: a pre-token ends where the next starts
    const uint32_t* wprefix;         // #starts before each mask word
    uint32_t* tok0;
    QView v[4];                      // queues: <= 16 bytes, <= 32, <= 64, longer (this workgroup fills sub-queue blockIdx of each)
    int* err;
    const unsigned long long* matchmask;   // added-token matches: one pre-token, id patched in later
    const uint4* hot;                // [HOT] {k0, k1, k2, id | len << 24}, len 0 = empty; then [HOT / 4] 16-bit displacements (tables.hpp)
    uint32_t word_seed;
    const uint4* shortw;             // the short-word table behind the hot table (tables.hpp): 16-byte slots {k0, k1, k2, id | len << 24 | SHORTW_DIRECT}
    const uint32_t* shortw_k3;       // bytes 12..15 of the key in slot i (read by the pre-tokens longer than 12 bytes only)
    const uint8_t* shortw_disp;      // [SHORTW_BUCKETS] the displacements: copied into LDS (shape HOT = 2048)
    uint32_t shortw_mask;
    uint32_t shortw_bmask;           // displacement buckets - 1
    uint32_t any_hit_final;          // ignore_merges / WordLevel / WordPiece: every hit is final (else only WORD_DIRECT ones)
    uint32_t no_hits;                // WordPiece with max_input_chars_per_word < 16: every word takes the trie walk
    uint32_t miss_is_unk;            // WordLevel: a miss of <= 16 bytes is the unk id (or MissingUnkToken), never queued
    uint32_t unk_id, has_unk;
    const CacheKey* cache_keys;      // word cache (kernels.hpp): words an earlier batch merged, or null
    unsigned long long* claims;      // in-batch word claims (below), or null
    uint32_t claim_mask;             // slots - 1
    uint32_t* counters;              // the batch's device counters (kernels.hpp CNT_*), or null
    unsigned long long* phases;      // PROF instantiation only (TKAMD_PHASES, tkamd_debug_phases): [workgroup][8] shader-clock ticks per phase
};
// phases of k_lookup<.., true>, as wavefront 0 sees the workgroup's barriers: staging the tile (LDS stores, the last pre-token's end,
// next tile's prefetch issued), expanding the mask bits into positions, pass 1, pass 2, pass 3 (+ waiting for the slowest wavefront);
// slot 7: the whole kernel
enum { LU_PH_STAGE = 0, LU_PH_EXPAND = 1, LU_PH_PASS1 = 2, LU_PH_PASS2 = 3, LU_PH_PASS3 = 4, LU_PH_TOTAL = 7 };

// slot of a word in the word cache, from the bucket hash of the whole-word table
__device__ __forceinline__ uint32_t cache_slot(uint32_t h1) { return (word_hash2(h1) >> 7) & ((1u << WORD_CACHE_BITS) - 1u); }

// (claim_hash_long / claim_slot / CLAIM_MAX_LEN: bpe.hip, next to the publish helper the model kernels call)
template <bool HAS_END, bool PROF = false>
__global__ __launch_bounds__(LU_NT, LU_WAVES_PER_SIMD) void k_lookup(LookupArgs a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) uint8_t, lu_lds)
    uint4* s_hot = (uint4*)lu_lds;                                              // [HOT]
    uint16_t* s_hdisp = (uint16_t*)(s_hot + HOT);                               // [HOT / 4]
    uint32_t* s_text32 = (uint32_t*)(s_hdisp + HOT / 4);                        // [(LU_TILE + LU_TEXT_SLACK) / 4 + 4]
    uint16_t* s_pos = (uint16_t*)(s_text32 + (LU_TILE + LU_TEXT_SLACK) / 4 + 4);    // [LU_POS_CAP + 2] start of rank r, relative to the tile
    uint16_t* s_miss = s_pos + LU_POS_CAP + 2;                                  // [LU_POS_CAP] ranks the hot table did not settle
    uint16_t* s_end = s_miss + LU_POS_CAP;                          // [LU_POS_CAP + 2] explicit ends (0xFFFF: beyond the tile)
    static_assert(hot_table_bytes(HOT) % 16 == 0, "the text behind the hot table is 16-byte aligned");
    __shared__ uint4 s_kmask[17];                                                // byte masks of a key of 0..16 bytes
    __shared__ uint32_t s_n, s_pbase, s_last_end, s_nmiss, s_ncand, s_ncandl, s_nretry;
    // the claims' yield as this workgroup sees it: candidates it looked at, how many of them were another pre-token's word.  Text that
    // never repeats a word pays two dependent round trips per candidate for nothing: a workgroup that has seen CLAIM_ADAPT_MIN
    // candidates and shared fewer than one in eight stops claiming for the rest of its tiles (its candidates are queued like any
    // other miss -- the claims are an optimisation, a word nobody claims is simply merged every time).
    __shared__ uint32_t s_seen, s_shared, s_claims_on;
    __shared__ uint32_t s_fill[4];                                               // fill of this workgroup's sub-queues so far
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (PROF: thread 0 stamps the shader clock behind every barrier that ends a phase; compiled out of the product instantiation)
    unsigned long long ph_t = 0ull, ph_t0 = 0ull, ph_acc[7] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    auto tick = [&](int k) {
        if (PROF && tid == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph_acc[k] += now - ph_t; ph_t = now; }
    };
    if (PROF && tid == 0) ph_t = ph_t0 = __builtin_amdgcn_s_memtime();
    const uint32_t sq = blockIdx.x;                                // this workgroup's PRIVATE sub-queue (the launcher keeps the grid <= NSQ)
    if (tid < 4) s_fill[tid] = 0u;
    if (tid == 0) { s_seen = 0u; s_shared = 0u; s_claims_on = a.claims ? 1u : 0u; }
    static_assert(hot_table_bytes(HOT) % 16 == 0, "whole 16-byte words");
    for (int i = tid; i < hot_table_bytes(HOT) / 16; i += LU_NT) s_hot[i] = a.hot[i];      // (slots and displacements: one buffer)
    const uint8_t* const wdisp = a.shortw_disp;                    // the short-word displacements: a hot 8 KB array in memory
    if (tid < 17) {
        const uint32_t l = (uint32_t)tid;
        auto m = [&](uint32_t lo) -> uint32_t { return l >= lo + 4u ? 0xFFFFFFFFu : (l > lo ? ((1u << (8u * (l - lo))) - 1u) : 0u); };
        s_kmask[tid] = make_uint4(m(0u), m(4u), m(8u), m(12u));
    }
    const int64_t n_bytes = uniform_i64(a.len_dev ? *a.len_dev : a.n_bytes_host);      // (scalar registers: the tile count and the bounds derived from it were two spilled pairs)
    const int64_t total_words = (n_bytes + 63) >> 6;                 // start bits exist only below n_bytes
    const int64_t end_words = (n_bytes >> 6) + 1;                    // an end bit can sit at byte n_bytes
    const int64_t n_tiles = (total_words + LU_TILE_WORDS - 1) / LU_TILE_WORDS;
    constexpr bool has_end = HAS_END;
    const bool hits_on = a.no_hits == 0u;
    const int64_t readable = a.n_bytes_host + TEXT_PAD;
    const int half = tid & 1, hword = tid >> 1;                      // lane pair (2 w, 2 w + 1) shares mask word w: low / high 32 bits
    // What a tile needs from memory -- its text, its mask words with their prefix counts, and the 64 mask words behind it (where its
    // last pre-token ends) -- is loaded into registers one tile AHEAD: the loads of tile k+1 are issued before the lookup phase of
    // tile k and have long arrived when tile k+1 starts, so no phase of a tile begins with a memory round trip.
    constexpr int LU_SLACK_CHUNKS = LU_TEXT_SLACK / 16;
    static_assert((LU_TILE + LU_TEXT_SLACK) / 16 == 2 * LU_NT + LU_SLACK_CHUNKS, "two 16-byte text chunks per lane (+ the slack chunks of the first lanes)");
    static_assert(LU_NT == 2 * LU_TILE_WORDS, "two lanes per mask word");
    unsigned long long pf_ms = 0ull, pf_me = 0ull, pf_scan = 0ull;
    uint32_t pf_wp = 0u, pf_first = 0u;
    // (80 VGPRs: the masks are kept a tile ahead, the text is loaded when the tile starts -- the other two workgroups of the CU cover that
    // round trip)
    // (lane offsets in 32 bits against uniform 64-bit bases: the 64-bit per-lane addresses and bounds of rounds 3-5 were loop invariants the
    // compiler kept in registers it did not have -- 80 VGPRs, three spilled pairs -- and a reload from scratch is a vector-memory operation:
    // the s_waitcnt vmcnt(0) behind it also waited for the text load just issued, so a tile's two text loads went out one round trip
    // after the other)
    auto load_text = [&](int64_t tile, Unaligned16& x0, Unaligned16& x1, Unaligned16& xs) {
        x0 = x1 = xs = Unaligned16{0u, 0u, 0u, 0u};
        if (tile >= n_tiles) return;
        const int64_t t0 = (tile * LU_TILE_WORDS) << 6;
        const uint8_t* const tb = uniform_ptr(a.text + t0);
        const uint32_t room = (uint32_t)min(readable - t0, (int64_t)(LU_TILE + LU_TEXT_SLACK + 64));      // readable bytes from t0 on, as far as this tile cares
        const uint32_t g0 = 16u * (uint32_t)tid, g1 = g0 + 16u * (uint32_t)LU_NT, gs = 16u * (uint32_t)(2 * LU_NT + tid);
        // (the tile's text, its masks and the tok0 words are read / written once: non-temporal accesses, kernels.hip -- level here, 0.2279
        // against 0.229 ms, 3 % in the compaction; profiles/r4m_ab_c2.txt)
        if (g0 + 16u <= room) x0 = load_nt16(tb + g0);
        if (g1 + 16u <= room) x1 = load_nt16(tb + g1);
        if (tid < LU_SLACK_CHUNKS && gs + 16u <= room) xs = load_nt16(tb + gs);
    };
    auto prefetch = [&](int64_t tile) {
        pf_ms = pf_me = pf_scan = 0ull;
        pf_wp = pf_first = 0u;
        if (tile >= n_tiles) return;
        const int64_t w0 = tile * LU_TILE_WORDS;
        // (uniform bases, 32-bit lane indices and bounds: see load_text)
        const int words_left = (int)min(total_words - w0, (int64_t)(LU_TILE_WORDS + 64)), ends_left = (int)min(end_words - w0, (int64_t)(LU_TILE_WORDS + 64));
        const unsigned long long* const sm0 = uniform_ptr(a.startmask + w0);
        const unsigned long long* const em0 = has_end ? uniform_ptr(a.endmask + w0) : sm0;
        const uint32_t* const wp0 = uniform_ptr(a.wprefix + w0);
        if (hword < words_left) { pf_ms = load_nt(sm0 + (uint32_t)hword); pf_wp = load_nt(wp0 + (uint32_t)hword); }
        if (has_end && hword < ends_left) pf_me = load_nt(em0 + (uint32_t)hword);
        pf_first = wp0[0];
        if (wave == 0) {
            const uint32_t ws = (uint32_t)(LU_TILE_WORDS + lane);
            if ((int)ws < (has_end ? ends_left : words_left)) pf_scan = em0[ws];
        }
    };
    prefetch(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t w0 = tile * LU_TILE_WORDS;
        const int64_t t0 = w0 << 6;                                  // first byte of the tile
        __syncthreads();                                             // previous tile's LDS use is over
        tick(LU_PH_PASS3);
        if (tid == 0 && s_claims_on && s_seen >= CLAIM_ADAPT_MIN && s_shared * 8u < s_seen) s_claims_on = 0u;   // (read behind the next barrier)
        // ---- 1. the tile's mask words (loaded a tile ago): local rank of each word's first start ----
        // (in front of the text loads: what reads a register a load of the LAST tile filled waits for every load issued since)
        const unsigned long long ms = pf_ms, me = pf_me;
        uint32_t rbase = 0xFFFFFFFFu;
        {
            const int wl = (int)min(total_words - w0, (int64_t)(LU_TILE_WORDS + 64));       // words of the text from w0 on (uniform; 32-bit lane compares)
            if (hword < wl) rbase = pf_wp - pf_first;
            if (tid == 0) s_pbase = pf_first;
            if (half && hword < wl && (hword == wl - 1 || hword == LU_TILE_WORDS - 1)) s_n = rbase + (uint32_t)__popcll(ms);
        }
        // ---- 2. text tile -> LDS (loaded now, dropped into LDS behind the work below) ----
        Unaligned16 tx0, tx1, txs;
        load_text(tile, tx0, tx1, txs);
        auto stage_text = [&]() {
            ((uint4*)s_text32)[tid] = make_uint4(tx0.a, tx0.b, tx0.c, tx0.d);
            ((uint4*)s_text32)[tid + LU_NT] = make_uint4(tx1.a, tx1.b, tx1.c, tx1.d);
            if (tid < LU_SLACK_CHUNKS) ((uint4*)s_text32)[2 * LU_NT + tid] = make_uint4(txs.a, txs.b, txs.c, txs.d);
        };
        // end of the tile's LAST pre-token when it lies beyond the tile: the next start (or end bit) after the tile -- almost always in
        // the 64 prefetched words behind it; otherwise wavefront 0 walks the mask on
        if (wave == 0) {
            const unsigned long long* mk = has_end ? a.endmask : a.startmask;
            const int64_t lim = has_end ? end_words : total_words;
            int64_t w = w0 + LU_TILE_WORDS;
            uint32_t found = (uint32_t)n_bytes;
            unsigned long long m = pf_scan;
            while (w < lim) {
                const uint64_t any = __ballot(m != 0ull);
                if (any) {
                    const int l = __ffsll((unsigned long long)any) - 1;
                    const unsigned long long mm = ((unsigned long long)(uint32_t)__shfl((int)(m >> 32), l, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)m, l, 64);
                    found = (uint32_t)(((w + l) << 6) + (__ffsll(mm) - 1));
                    break;
                }
                w += 64;
                m = (w + lane < lim) ? mk[w + lane] : 0ull;
            }
            if (lane == 0) s_last_end = found;
        }
        stage_text();
        prefetch(tile + gridDim.x);                                   // the next tile's loads fly while this one is looked up
        __syncthreads();
        tick(LU_PH_STAGE);
        const bool claims_now = s_claims_on != 0u;                   // (workgroup-uniform for the whole tile)
        const uint32_t n = s_n;                                      // pre-tokens starting in this tile
        const uint32_t pbase = s_pbase;                              // global rank of the first one
        const uint32_t last_rel = s_last_end - (uint32_t)t0;         // (positions below are relative to the tile)
        for (uint32_t rb = 0; rb < n; rb += LU_POS_CAP) {
            if (rb) __syncthreads();                                 // previous round has read s_pos / s_end
            const uint32_t cnt = min((uint32_t)LU_POS_CAP, n - rb);
            if (has_end) {
                for (int i = tid; i < LU_POS_CAP + 2; i += LU_NT) s_end[i] = 0xFFFFu;
                __syncthreads();
            }
            // ---- 2b. positions of the set bits, by rank (ranks rb .. rb + cnt, the extra one is the next start) ----
            if (tid == 0) { s_nmiss = 0u; s_ncand = 0u; s_ncandl = 0u; s_nretry = 0u; }
            if (rbase != 0xFFFFFFFFu) {
                const uint32_t lo32 = (uint32_t)ms, bit0 = (uint32_t)hword * 64u + (uint32_t)half * 32u;
                uint32_t r = rbase - rb + (half ? (uint32_t)__popc(lo32) : 0u);
                for (uint32_t m = half ? (uint32_t)(ms >> 32) : lo32; m; m &= m - 1u, ++r)
                    if (r <= cnt) s_pos[r] = (uint16_t)(bit0 + (uint32_t)(__ffs(m) - 1));
                if (has_end) {
                    for (uint32_t m = half ? (uint32_t)(me >> 32) : (uint32_t)me; m; m &= m - 1u) {
                        const int b = (__ffs(m) - 1) + half * 32;
                        // the pre-token this end closes: the last start before it
                        const uint32_t before = rbase + (uint32_t)__popcll(ms & ((1ull << b) - 1ull));
                        if (before == 0u) continue;                  // closes a pre-token of an earlier tile
                        const uint32_t rel = before - 1u - rb;
                        if (rel < cnt) s_end[rel] = (uint16_t)((uint32_t)hword * 64u + (uint32_t)b);
                    }
                }
            }
            __syncthreads();
            tick(LU_PH_EXPAND);
            // start, end (tile relative) and the first 16 key bytes of the pre-token of local rank rel (< cnt)
            auto load_key = [&](uint32_t rel, uint32_t& s_rel, uint32_t& len, uint32_t& k0, uint32_t& k1, uint32_t& k2, uint32_t& k3, bool want_k3) {
                s_rel = s_pos[rel];
                uint32_t e_rel;
                if (HAS_END) { e_rel = s_end[rel]; e_rel = e_rel == 0xFFFFu ? last_rel : e_rel; }
                else e_rel = rb + rel + 1u >= n ? last_rel : (uint32_t)s_pos[rel + 1u];
                len = e_rel - s_rel;
                const uint32_t wi = s_rel >> 2, sh = s_rel & 3u;
                const uint32_t d0 = s_text32[wi], d1 = s_text32[wi + 1], d2 = s_text32[wi + 2], d3 = s_text32[wi + 3];
                const uint4 km = s_kmask[min(len, 16u)];
                k0 = __builtin_amdgcn_alignbyte(d1, d0, sh) & km.x;
                k1 = __builtin_amdgcn_alignbyte(d2, d1, sh) & km.y;
                k2 = __builtin_amdgcn_alignbyte(d3, d2, sh) & km.z;
                k3 = want_k3 ? (__builtin_amdgcn_alignbyte(s_text32[wi + 4], d3, sh) & km.w) : 0u;
            };
            // ---- 3. pass 1: the hot table, every pre-token ----
            const uint32_t n_chunks = (cnt + 63u) >> 6;
            for (uint32_t chunk = (uint32_t)wave; chunk < n_chunks; chunk += LU_WAVES) {
                const uint32_t rel = chunk * 64u + lane;
                const bool v = rel < cnt;
                uint32_t s_rel, len, k0, k1, k2, k3;
                load_key(v ? rel : cnt - 1u, s_rel, len, k0, k1, k2, k3, false);
                // settled words of <= 12 bytes (a slot's length is <= 12 and 0 when empty, so a hit proves bytes 12.. are not part of the key)
                const uint32_t hh = hot_hash(k0, k1, k2, len, a.word_seed);
                const uint4 h = s_hot[hot_slot(hh, (uint32_t)s_hdisp[hot_bucket(hh, (uint32_t)HOT)], (uint32_t)HOT)];
                const uint32_t diff = (h.x ^ k0) | (h.y ^ k1) | (h.z ^ k2) | ((h.w >> 24) ^ len);
                bool hit = v && diff == 0u && len != 0u && hits_on;
                bool miss = v && !hit;
                if (a.matchmask) {                                                  // wavefront-uniform: tokenizers with added tokens only
                    const uint32_t s_abs = (uint32_t)t0 + s_rel;
                    if (v && len && ((a.matchmask[s_abs >> 6] >> (s_abs & 63u)) & 1ull)) {       // an added-token match: its id is patched in later
                        a.tok0[pbase + rb + rel] = 0u;
                        hit = false;
                        miss = false;
                    }
                }
                // (the misses store a placeholder too, so that a wavefront's 64 words leave as whole lines; pass 2 / 3 overwrite them:
                // 0.2337 -> 0.2296 ms on C2 against storing the hits alone.  PLAIN stores since round 6: the L2 then merges the placeholder
                // with the word pass 2 / 3 write microseconds later, and the compaction finds tok0 there -- non-temporal stores sent
                // every partial line out on its own, WRITE_SIZE 272 MB for 79 MB of tok0.  Same-session A/B, twice: lookup 0.209 ->
                // 0.202 ms, compact 0.115 -> 0.111, the step 0.489 -> 0.478 ms, profiles/r6h_ab_c2_plain_tok0.txt)
                if (hit || miss) {
                    const uint32_t w0 = hit ? (TOK_ONE | (h.w & TOK_ID_MASK)) : 0u;
                    a.tok0[pbase + rb + rel] = w0;
                }
                const uint64_t mb = __ballot(miss);
                if (mb) {                                                           // (wavefront-uniform) the workgroup's miss list
                    uint32_t base = 0u;
                    if (lane == 0) base = atomicAdd(&s_nmiss, (uint32_t)__popcll(mb));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (miss) s_miss[base + (uint32_t)mbcnt64(mb)] = (uint16_t)rel;
                }
            }
            __syncthreads();
            tick(LU_PH_PASS1);
            // what pass 2 and pass 3 end a lane with: a pre-token still pending is a model kernel's work, queued by length class
            // (<= 16 bytes, <= 32, <= 64, longer); its tok0 word names the row, any other lane's its result.  (wavefront-wide: ballots)
            // (holds: this pre-token won the claim of its word -- QLEN_CLAIM in its queue entry makes the model kernel publish its row)
            auto finish = [&](bool v, bool pend, uint32_t rel, uint32_t s_rel, uint32_t len, uint32_t out, bool holds) {
                const uint32_t c = len <= 16u ? 0u : (len <= 32u ? 1u : (len <= 64u ? 2u : 3u));
                const uint64_t b0 = __ballot(pend && c == 0u), b1 = __ballot(pend && c == 1u), b2 = __ballot(pend && c == 2u), b3 = __ballot(pend && c == 3u);
                if (b0 | b1 | b2 | b3) {                                            // wavefront-uniform
                    uint32_t base = 0u;
                    if (lane < 4) {
                        const uint32_t take = (uint32_t)__popcll(lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3);
                        if (take) base = atomicAdd(&s_fill[lane], take);             // LDS: the sub-queue is this workgroup's alone
                    }
                    base = (uint32_t)__shfl((int)base, (int)c, 64);
                    if (pend) {
                        const uint32_t pos = base + (uint32_t)mbcnt64(c == 0u ? b0 : c == 1u ? b1 : c == 2u ? b2 : b3);
                        // (selects, not indexed loads: the argument arrays must stay in scalar registers)
                        QItem* const qp = c == 0u ? a.v[0].q : c == 1u ? a.v[1].q : c == 2u ? a.v[2].q : a.v[3].q;
                        const uint32_t qcap = c == 0u ? a.v[0].sq_cap : c == 1u ? a.v[1].sq_cap : c == 2u ? a.v[2].sq_cap : a.v[3].sq_cap;
                        const uint32_t rowb = c == 0u ? a.v[0].row_base : c == 1u ? a.v[1].row_base : c == 2u ? a.v[2].row_base : a.v[3].row_base;
                        if (pos < qcap) {
                            qp[sq * qcap + pos] = QItem{(uint32_t)t0 + s_rel, len | (holds ? QLEN_CLAIM : 0u)};
                            out = TOK_ROW | (rowb + sq * qcap + pos);
                        } else {
                            atomicOr(a.err, ERR_QUEUE_FULL);                         // the host grows the queues and runs the batch again
                        }
                    }
                }
                if (v) a.tok0[pbase + rb + rel] = out;
            };
            // In-batch claims (the section behind this kernel).  The claim of a candidate -- a pre-token of <= 32 bytes the tables did
            // not settle -- is two more dependent round trips (the slot, then the claimant's bytes), and a step of pass 2 waits for its
            // slowest lane: inline they doubled the chain of EVERY step for the third of its lanes that are candidates.  Without end
            // masks there is LDS left for a list of the candidates: pass 2 only notes them, PASS 3 takes them packed 64 to a step --
            // a third as many steps pay the two round trips.  (With end masks the list does not fit: the claims stay inline.)
            constexpr bool CAND_PASS = !HAS_END;
            uint16_t* const s_cand = s_end;                                         // (the end array's place: unused without end masks)
            // bytes 16..31 of a key of 17..32 bytes, masked; and the hash the word claims with
            auto long_key = [&](uint32_t s_rel, uint32_t len, uint32_t h16, uint32_t& k4, uint32_t& k5, uint32_t& k6, uint32_t& k7, uint4& kmh) -> uint32_t {
                const uint32_t wi = (s_rel >> 2) + 4u, sh = s_rel & 3u;
                const uint32_t d4 = s_text32[wi], d5 = s_text32[wi + 1], d6 = s_text32[wi + 2], d7 = s_text32[wi + 3], d8 = s_text32[wi + 4];
                kmh = s_kmask[len - 16u];
                k4 = __builtin_amdgcn_alignbyte(d5, d4, sh) & kmh.x; k5 = __builtin_amdgcn_alignbyte(d6, d5, sh) & kmh.y;
                k6 = __builtin_amdgcn_alignbyte(d7, d6, sh) & kmh.z; k7 = __builtin_amdgcn_alignbyte(d8, d7, sh) & kmh.w;
                return claim_hash_long(h16, k4, k5, k6, k7);
            };
            // The claim protocol of one candidate.  Returns CLAIM_SHARED if the word is another pre-token's (tok0 -> TOK_SLOT | slot, not queued),
            // CLAIM_HOLDS if this pre-token now holds the claim (queued, its entry flagged: the model kernel publishes its row to the slot),
            // CLAIM_NONE if the slot is another word's or the claim could not be read whole (queued like any other miss).  A slot only ever
            // goes from 0 to its claim, so a claim read is final; a 0 is followed by the compare-and-swap.
            // A word of <= CLAIM_KEY_MAX (15) bytes claims with ITSELF: word 0 of the entry = bytes 0..6 | length << 56 (never 0, never the long
            // form's 0xFF), word 1 = bytes 7..14, stored by the winner right behind its compare-and-swap.  A later occurrence compares the
            // entry with its own key -- ONE line, where round 4's entry named the claimant's first byte and every candidate fetched a second
            // line somewhere in the text (2.4 M random HBM lines a batch on C2: tools/claims_sim.py).  An entry whose word 0
            // matches while word 1 does not (another word with the same first seven bytes and length, or the winner's second store still
            // on its way) is NOT waited for: the pre-token is queued and merged on its own, which is always right.
            // A plain (cached) read first: a claim it shows is final, and the repeats of a frequent word then hit the L2 instead of crossing
            // the fabric each time.  A 0 may be stale (the L2 of an XCD keeps what it read whatever another XCD's CAS did since): the
            // device-scope read confirms it (tools/microbench/claims_probe.hip).
            // (CLAIM_RETRY, only when the caller can come back -- `last` false: word 0 is this word's, word 1 not there yet)
            enum : uint32_t { CLAIM_NONE = 0u, CLAIM_SHARED = 1u, CLAIM_HOLDS = 2u, CLAIM_RETRY = 3u };
            auto claim_short = [&](uint32_t slot, uint32_t len, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, bool last) -> uint32_t {
                unsigned long long* const e = a.claims + 2u * (size_t)slot;
                const unsigned long long w0 = (unsigned long long)k0 | ((unsigned long long)(k1 & 0x00FFFFFFu) << 32) | ((unsigned long long)len << 56);
                const unsigned long long w1 = (unsigned long long)(k1 >> 24) | ((unsigned long long)k2 << 8) | ((unsigned long long)(k3 & 0x00FFFFFFu) << 40);
                // A word of 8..15 bytes whose bytes 7.. are all NUL has w1 == 0 -- indistinguishable from "word 1 not written yet" of another
                // word with the same first seven bytes and length.  Such a word neither claims nor shares: it is queued and merged on its own.
                if (len > 7u && w1 == 0ull) return CLAIM_NONE;
                const ulonglong2 c = *(const ulonglong2*)e;
                unsigned long long c0 = c.x, c1 = c.y;
                bool fresh1 = false, won = false;
                // (a 0 may be stale -- the L2 of an XCD keeps what it read whatever another XCD's compare-and-swap did since.  With the
                // candidates' own pass the 0 is confirmed by a device-scope load first; inline in pass 2 -- end masks -- the
                // compare-and-swap itself is the fresh read: one round trip less in the chain of every step, C3's lookup 0.240 -> 0.222 ms,
                // where the candidates' pass lost on out-of-distribution text, 0.332 -> 0.365 ms with a tenth more words merged on their
                // own -- the sharers then arrive inside the window of the winner's second store: profiles/r7o_*)
                if (c0 == 0ull && CAND_PASS) c0 = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c0 == 0ull) { c0 = atomicCAS(e, 0ull, w0); won = c0 == 0ull; }
                // The winner's second store comes BEFORE any loser reads word 1, in program order: lanes of one wavefront step often hold
                // the same word (text of few distinct words), and a loser polling in its own branch would spin on a store its wavefront
                // has not issued yet.
                if (won) {
                    if (w1) __hip_atomic_store(e + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return CLAIM_HOLDS;
                }
                if (c0 != w0) return CLAIM_NONE;
                if (c.x == 0ull && w1) { c1 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); fresh1 = true; }      // (claimed since the stale read)
                if (c1 == w1) return CLAIM_SHARED;
                // word 1 is written once, so a value read is final; a 0 may be a stale line or the winner's store still on its way (it follows
                // the compare-and-swap by a round trip, and on a small batch of few distinct words -- every tile's first steps at once --
                // a quarter of the occurrences arrive inside that window: profiles/r5d_pytest.txt).  One fresh read; then the candidate
                // goes to the tile's retry list (pass 3 comes back to it behind its other steps), or, where there is no coming back, a
                // few more reads -- never a wait without end: a word that cannot be settled is queued, which is always right.
                if (c1 == 0ull && !fresh1) c1 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c1 == 0ull && !last) return CLAIM_RETRY;
#pragma unroll 1
                for (int tries = 0; c1 == 0ull && tries < 3; ++tries) c1 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return c1 == w1 ? CLAIM_SHARED : CLAIM_NONE;
            };
            // 16..32 bytes (9 % of C2's candidates): the entry names the claimant -- 0xFF << 56 | length << 32 | first byte -- and its BYTES in
            // the text are the key (immutable: nothing waits for another lane's writes, no result depends on which occurrence wins)
            auto claim_long = [&](uint32_t slot, uint32_t s_rel, uint32_t len, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3,
                                  uint32_t k4, uint32_t k5, uint32_t k6, uint32_t k7, const uint4& kmh) -> uint32_t {
                unsigned long long* const e = a.claims + 2u * (size_t)slot;
                const unsigned long long tag = (0xFFull << 56) | ((unsigned long long)len << 32);
                unsigned long long c = *e;
                if (c == 0ull && CAND_PASS) c = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (see claim_short)
                if (c == 0ull) {
                    c = atomicCAS(e, 0ull, tag | (unsigned long long)((uint32_t)t0 + s_rel));
                    if (c == 0ull) return CLAIM_HOLDS;
                }
                if ((c >> 32) != (tag >> 32)) return CLAIM_NONE;
                const Unaligned16 o = *(const Unaligned16*)(a.text + (uint32_t)c);  // (readable: the text carries TEXT_PAD bytes of slack)
                const Unaligned16 o2 = *(const Unaligned16*)(a.text + (uint32_t)c + 16u);
                return ((((o.a) ^ k0) | ((o.b) ^ k1) | ((o.c) ^ k2) | ((o.d) ^ k3) |
                         ((o2.a & kmh.x) ^ k4) | ((o2.b & kmh.y) ^ k5) | ((o2.c & kmh.z) ^ k6) | ((o2.d & kmh.w) ^ k7)) == 0u) ? CLAIM_SHARED : CLAIM_NONE;
            };
            // the claim of one candidate of any length (<= CLAIM_MAX_LEN); h1: the whole-word table's hash of its first 16 bytes
            auto claim_any = [&](uint32_t s_rel, uint32_t len, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t h1, uint32_t& slot) -> uint32_t {
                if (len <= CLAIM_KEY_MAX) {
                    slot = claim_slot(h1, a.claim_mask);
                    return claim_short(slot, len, k0, k1, k2, k3, true);
                }
                uint32_t k4 = 0u, k5 = 0u, k6 = 0u, k7 = 0u, hc = h1;
                uint4 kmh = make_uint4(0u, 0u, 0u, 0u);
                if (len > (uint32_t)WORD_MAX_KEY) hc = long_key(s_rel, len, h1, k4, k5, k6, k7, kmh);
                slot = claim_slot(hc, a.claim_mask);
                return claim_long(slot, s_rel, len, k0, k1, k2, k3, k4, k5, k6, k7, kmh);
            };
            // ---- 4. pass 2: the misses, packed 64 to a step, steps dealt round robin ----
            // A step is the key + hash + the bucket's displacement (LDS), ONE memory round trip -- the word's one slot of the short-word
            // table (tables.hpp), 16 bytes a lane -- and the verdict.  (Until round 4 the table had 32-byte slots and its displacements in
            // HBM: two dependent round trips, three requests.  Two of THOSE steps side by side, stage by stage, had measured slower, 0.2475
            // against 0.2337 ms, as batching the probes had in round 2: profiles/r4d_ab_c2.txt.)
            struct P2 { bool v, probe; uint32_t rel, s_rel, len, k0, k1, k2, k3, h1, k3t; uint4 a0; };
            const uint32_t n_miss = s_nmiss;
            auto p2_key = [&](uint32_t m0, P2& x) {
                x.v = m0 + lane < n_miss;
                x.rel = s_miss[x.v ? m0 + lane : m0];
                load_key(x.rel, x.s_rel, x.len, x.k0, x.k1, x.k2, x.k3, true);
                x.h1 = 0u;
                x.k3t = 0u;
                x.probe = x.v && hits_on && x.len != 0u && x.len <= (uint32_t)WORD_MAX_KEY;
                x.a0 = make_uint4(0u, 0u, 0u, 0u);
                if (x.probe) {                                                      // displacement from LDS, then ONE 16-byte request (+ 4 bytes for the few keys longer than 12)
                    x.h1 = word_hash1_from_hot(hot_hash(x.k0, x.k1, x.k2, x.len, a.word_seed), x.k3);
                    const uint32_t slot = shortw_slot(x.h1, shortw_kmix(x.k0, x.k1, x.k2, x.k3), (uint32_t)wdisp[x.h1 & a.shortw_bmask], a.shortw_mask);
                    x.a0 = a.shortw[slot];
                    if (x.len > (uint32_t)HOT_MAX_KEY) x.k3t = a.shortw_k3[slot];
                }
            };
            auto p2_done = [&](P2& x) {
                const bool v = x.v;
                const uint32_t rel = x.rel, s_rel = x.s_rel, len = x.len, k0 = x.k0, k1 = x.k1, k2 = x.k2, k3 = x.k3, h1 = x.h1;
                uint32_t out = 0u;
                bool pend = v;
                if (x.probe) {
                    const bool found = ((x.a0.x ^ k0) | (x.a0.y ^ k1) | (x.a0.z ^ k2) | (x.k3t ^ k3) | (((x.a0.w >> SHORTW_LEN_SHIFT) & SHORTW_LEN_MASK) ^ len)) == 0u;
                    const uint32_t id = x.a0.w & SHORTW_ID_MASK;
                    const bool direct = (x.a0.w & SHORTW_DIRECT) != 0u;
                    if (found && (a.any_hit_final || direct)) { out = TOK_ONE | id; pend = false; }
                    if (a.cache_keys && pend) {                                     // (the outer test is wavefront-uniform) the word cache: merged by an earlier batch?
                        // probed only by the lanes the table did not settle: riding along with the table probe for every lane was
                        // measured 20 % slower (the extra 32-byte reads cost more than the round trip they save)
                        const uint32_t cslot = cache_slot(h1);
                        const CacheKey* const ck = a.cache_keys + cslot;
                        const uint4 ckey = *(const uint4*)ck->k;
                        const uint32_t cstate = ck->state;                          // (an empty slot has state 0, never a length)
                        if (((ckey.x ^ k0) | (ckey.y ^ k1) | (ckey.z ^ k2) | (ckey.w ^ k3) | (cstate ^ len)) == 0u) { out = TOK_SLOT | cslot; pend = false; }
                    }
                }
                if (a.miss_is_unk) {                                                // wavefront-uniform: WordLevel (wordlevel/mod.rs:170-177)
                    if (pend && len <= (uint32_t)WORD_MAX_KEY) {
                        if (a.has_unk) out = TOK_ONE | a.unk_id;
                        else atomicOr(a.err, ERR_MISSING_UNK);
                        pend = false;
                    }
                }
                bool cand = false, holds = false;
                if (claims_now) {                                                   // wavefront-uniform
                    cand = pend && hits_on && len != 0u && len <= CLAIM_MAX_LEN;
                    if (CAND_PASS) {
                        // two lists in the one array: the words whose claim entry is their key from the front, the 16..32-byte ones -- a second
                        // dependent round trip, and a step waits for its slowest lane -- from the back: their steps are few and their own
                        const bool lng = cand && len > CLAIM_KEY_MAX;
                        const uint64_t cb = __ballot(cand && !lng), lb = __ballot(lng);
                        if (cb | lb) {                                              // (wavefront-uniform) the workgroup's candidate lists
                            uint32_t base = 0u;
                            if (lane == 0 && cb) base = atomicAdd(&s_ncand, (uint32_t)__popcll(cb));
                            if (lane == 1 && lb) base = atomicAdd(&s_ncandl, (uint32_t)__popcll(lb));
                            const uint32_t bs = (uint32_t)__builtin_amdgcn_readlane((int)base, 0), bl = (uint32_t)__builtin_amdgcn_readlane((int)base, 1);
                            if (cand && !lng) s_cand[bs + (uint32_t)mbcnt64(cb)] = (uint16_t)rel;
                            if (lng) s_cand[(uint32_t)LU_POS_CAP - 1u - (bl + (uint32_t)mbcnt64(lb))] = (uint16_t)rel;
                        }
                    } else if (cand) {
                        // (Round 5 tried a candidate list of the WAVEFRONT's own here -- in the miss-list slots it has already read, its steps of
                        // candidates behind its steps of misses, no barrier: C3's lookup 0.2545 -> 0.2685 ms, profiles/r5h_ab_c3.txt.  A
                        // wavefront has ~40 candidates a tile: two more sparsely filled steps cost more than the claim chain inside the steps
                        // it runs anyway.)
                        uint32_t slot = 0u;                                         // (h1 is the probe's: a word beyond the table's 16-byte keys was not probed)
                        const uint32_t r = claim_any(s_rel, len, k0, k1, k2, k3, x.probe ? h1 : word_hash1_from_hot(hot_hash(k0, k1, k2, len, a.word_seed), k3), slot);
                        if (r == CLAIM_SHARED) { out = TOK_SLOT | slot; pend = false; }
                        holds = r == CLAIM_HOLDS;
                    }
                    if (!CAND_PASS) {                                               // (the inline variant keeps the yield per step)
                        const uint64_t cb = __ballot(cand), sb = __ballot(cand && !pend);
                        if (cb && lane == 0) { atomicAdd(&s_seen, (uint32_t)__popcll(cb)); if (sb) atomicAdd(&s_shared, (uint32_t)__popcll(sb)); }
                        cand = false;
                    }
                } else if (a.claims) {                                              // (wavefront-uniform) given up: the candidates still count -- the host's pause rule wants the batch's yield
                    const uint64_t cb = __ballot(pend && hits_on && len != 0u && len <= CLAIM_MAX_LEN);
                    if (cb && lane == 0) atomicAdd(&s_seen, (uint32_t)__popcll(cb));
                }
                finish(v && !cand, pend && !cand, rel, s_rel, len, out, holds);     // (a listed candidate is finished by pass 3)
            };
            // (one step at a time: the 80-register shape has no room for a second step's state side by side)
            for (uint32_t m0 = (uint32_t)wave * 64u; m0 < n_miss; m0 += (uint32_t)LU_NT) {
                P2 x;
                p2_key(m0, x);
                p2_done(x);
            }
            // ---- 5. pass 3: the candidates, packed 64 to a step ----
            // (Round 4 parked a wavefront's step of candidates in registers and advanced its chain -- slot, compare-and-swap, the
            // claimant's bytes -- one link behind each barrier of the NEXT tile: slower, 0.255 against 0.243 ms and 0.64 against 0.46 ms on
            // out-of-distribution text, profiles/r4h_ab_c2*.txt.  A wait for one load is a wait for every memory operation issued
            // before it -- s_waitcnt counts in order -- so each link also waited for the next tile's prefetch and the tok0 stores just
            // issued; and the 17 extra live registers cost the other passes.)
            if (CAND_PASS && claims_now) {                                          // wavefront-uniform
                __syncthreads();
                tick(LU_PH_PASS2);
                const uint32_t n_cand = s_ncand, n_candl = s_ncandl;
                if (tid == 0) s_seen += n_cand + n_candl;                           // (thread 0 alone writes it)
                // the short words' steps, then the long ones' (from the back of the list); steps dealt round robin over both.  A short word
                // whose claim is half written (CLAIM_RETRY) goes to the retry list -- the miss list's place, pass 2 is done with it -- and
                // is looked at once more behind a barrier, when the winner's second store has long landed.
                uint16_t* const s_retry = s_miss;
                const uint32_t steps_s = (n_cand + 63u) >> 6, steps_l = (n_candl + 63u) >> 6;
                auto claim_step = [&](bool v, uint32_t rel, bool lng, bool last) {
                    uint32_t s_rel, len, k0, k1, k2, k3;
                    load_key(rel, s_rel, len, k0, k1, k2, k3, true);
                    const uint32_t h1 = word_hash1_from_hot(hot_hash(k0, k1, k2, len, a.word_seed), k3);
                    uint32_t out = 0u, slot = 0u, r = CLAIM_NONE;
                    bool pend = v;
                    if (v) {
                        if (lng) r = claim_any(s_rel, len, k0, k1, k2, k3, h1, slot);
                        else { slot = claim_slot(h1, a.claim_mask); r = claim_short(slot, len, k0, k1, k2, k3, last); }
                    }
                    if (r == CLAIM_SHARED) { out = TOK_SLOT | slot; pend = false; }
                    const bool again = r == CLAIM_RETRY;
                    const uint64_t rbm = __ballot(again);
                    if (rbm) {                                                      // (wavefront-uniform; never in the last round)
                        uint32_t base = 0u;
                        if (lane == 0) base = atomicAdd(&s_nretry, (uint32_t)__popcll(rbm));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (again) s_retry[base + (uint32_t)mbcnt64(rbm)] = (uint16_t)rel;
                    }
                    const uint64_t sb = __ballot(v && !pend && !again);
                    if (sb && lane == 0) atomicAdd(&s_shared, (uint32_t)__popcll(sb));
                    finish(v && !again, pend && !again, rel, s_rel, len, out, r == CLAIM_HOLDS);
                };
                for (uint32_t st = (uint32_t)wave; st < steps_s + steps_l; st += (uint32_t)LU_WAVES) {
                    const bool lng = st >= steps_s;                                 // wavefront-uniform
                    const uint32_t c0 = (lng ? st - steps_s : st) * 64u, nc = lng ? n_candl : n_cand;
                    const bool v = c0 + lane < nc;
                    const uint32_t ci = v ? c0 + lane : c0;
                    claim_step(v, s_cand[lng ? (uint32_t)LU_POS_CAP - 1u - ci : ci], lng, false);
                }
                __syncthreads();
                const uint32_t n_retry = s_nretry;                                  // (workgroup-uniform; 0 on all but a batch's first steps)
                for (uint32_t c0 = (uint32_t)wave * 64u; c0 < n_retry; c0 += (uint32_t)LU_NT) {
                    const bool v = c0 + lane < n_retry;
                    claim_step(v, s_retry[v ? c0 + lane : c0], false, true);
                }
            }
        }
    }
    // the fill of this workgroup's sub-queues (the counters were zeroed by the host; a workgroup without tiles leaves them 0)
    __syncthreads();
    tick(LU_PH_PASS3);
    if (PROF && tid == 0 && a.phases) {
        unsigned long long* const o = a.phases + (size_t)blockIdx.x * 8;
        for (int k = 0; k < 5; ++k) o[k] += ph_acc[k];
        o[LU_PH_TOTAL] += ph_t - ph_t0;
    }
    if (tid == 0 && a.claims && a.counters) {                      // the batch's totals: what the host's pause rule reads (capi.cpp read_scalars)
        if (s_seen) atomicAdd(a.counters + CNT_CLAIM_CANDS, s_seen);
        if (s_shared) atomicAdd(a.counters + CNT_CLAIM_SHARED, s_shared);
        if (!s_claims_on) atomicAdd(a.counters + CNT_CLAIM_GAVE_UP, 1u);
    }
    if (tid < 4) {
        uint32_t* const cnt_p = tid == 0 ? a.v[0].counts : tid == 1 ? a.v[1].counts : tid == 2 ? a.v[2].counts : a.v[3].counts;
        const uint32_t cap = tid == 0 ? a.v[0].sq_cap : tid == 1 ? a.v[1].sq_cap : tid == 2 ? a.v[2].sq_cap : a.v[3].sq_cap;
        cnt_p[sq * QCNT_STRIDE] = min(s_fill[tid], cap);
    }
}

// =================================================================================================
// In-batch word claims.  Natural text repeats its words: of the pre-tokens the static tables do not settle (13 % on C2) a few
// percent are distinct.  The reference's per-thread cache exploits that (BPE::tokenize_with_cache, bpe/model.rs:573-586); here the
// FIRST occurrence of such a word claims the slot of its hash in a table of 16-byte entries (claim_short / claim_long in k_lookup:
// pass 3; inline in pass 2 with end masks), is queued for the model kernel with QLEN_CLAIM in its queue entry, and every other
// occurrence finds the claim, checks it -- against the KEY in the entry for a word of <= 15 bytes (round 5: one line per candidate),
// against the claimant's BYTES in the text for a longer one (immutable: nothing waits for another lane's writes, and no result depends
// on which occurrence wins) -- is not queued and points its tok0 at the slot's row (TOK_SLOT | slot).  The model kernels copy the
// claimants' finished rows there (claim_publish_item, bpe.hip: the flag in the queue entry says who, the table is not read again); the
// compaction reads them like the rows of the word cache, and k_token_meta takes the token ends of a shared row from the claimant's slots
// of tmp_end (claim_pos[slot]).  A word whose slot another word holds is simply merged every time.  The table is zeroed per batch: no
// state crosses batches.
// Round 5 measured what the candidates ARE before changing the entry (tools/claims_sim.py over the bench's own first batch: 2.37 M
// candidates, 219 k distinct words, a flat tail -- the 4,096 most frequent of them cover 10 %): 0.2 % repeat inside their 16 KB tile and
// 1.6 % inside everything their workgroup ever sees, so a workgroup-local table in front of the claims would answer next to nothing;
// 39 % are <= 7 bytes, 47 % <= 8, 91 % <= 15 (profiles/r5_claims_sim.txt).
// Measured on the way here (C2, 2.56 M candidates; tools/microbench/claims_probe.hip, tools/cm_probe.py):
//   * all reads of the table are device-scope loads: the L2 of an XCD keeps a line it read as 0 whatever another XCD's CAS did since
//     (a plain load after a remote store was stale in 63 of 63 workgroups, a device-scope load fresh in all);
//   * the claims as kernels of their own between the lookup and the model kernels (mark the repeats, compact the sub-queues in place)
//     cost 0.19 ms against 0.06 ms inside the lookup: re-reading the queue entries and keys (0.04), the claim reads (0.05), the
//     claimants' bytes (0.04), the rank of every repeat for its tok0 word (0.03) -- all of which pass 2 has in registers or in LDS;
//     four entries per lane made it slower (0.30), a seeding pass for the frequent words changed nothing;
//   * two slots per word, and a separate chain for the 17..32-byte words, each added a serialised round trip to nearly every step
//     of pass 2 (a step waits for its slowest lane): one slot, one chain;
//   * reading the slot alongside the table probe (more device-scope loads, no shorter chain in practice) was slower than reading it
//     after the probe has missed (0.286 against 0.251 ms).
// =================================================================================================
constexpr int lookup_lds_bytes() {      // (the end array's place holds the candidate list when there are no end masks)
    return hot_table_bytes(HOT) + (LU_TILE + LU_TEXT_SLACK) + 16 + 2 * (LU_POS_CAP + 2) * 2 + LU_POS_CAP * 2;
}
static_assert(3 * (lookup_lds_bytes() + 1024) <= 163840, "three workgroups share a CU's 160 KB (1 KB: the kernel's static LDS)");

// =================================================================================================
// K_word_cache_insert: after the merge kernels, every queued pre-token of <= 16 bytes whose result fits a row (<= 4 tokens) is
// offered to the word cache: the lane claims the slot of its word if it is still empty (a slot never changes hands: like the
// reference's cache, utils/cache.rs, the table only fills) and writes key, row and length.  The lookup kernel of the NEXT batch is the
// only reader, so readers never see a half-written entry.
// =================================================================================================
__global__ __launch_bounds__(256) void k_word_cache_insert(DevTables t, const uint8_t* __restrict__ text, QView v, const uint4* __restrict__ rows,
                                                           WordCache wc) {
    __shared__ uint32_t s_qpre[NSQ + 1];
    const uint32_t n = qview_prefix(v, s_qpre);
    for (uint32_t item = blockIdx.x * 256 + threadIdx.x; item < n; item += gridDim.x * 256) {
        const uint32_t qpos = qview_pos(s_qpre, v.sq_cap, item);
        QItem it = v.q[qpos];
        it.len = qitem_len(it.len);
        if (it.len == 0u || it.len > 16u) continue;
        const uint4 row = rows[v.row_base + qpos];
        if ((row.x >> ROW_CNT_SHIFT) > 4u) continue;                                // longer results stay with the merge kernels
        uint64_t lo, hi;
        load_key16(text, it.s, it.len, &lo, &hi);
        const uint32_t slot = cache_slot(word_hash1(lo, hi, it.len, t.word_seed));
        CacheKey* const ck = wc.keys + slot;
        if (atomicCAS(&ck->state, 0u, CACHE_CLAIMED) != 0u) continue;               // taken (by this word or another)
        *(uint4*)ck->k = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        ((uint4*)wc.rows)[slot] = row;
        ck->state = it.len;                                                         // (read by later kernels only: no fence needed)
    }
}
