// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Start-mask scans and pre-token offset emission.

// =================================================================================================
// Prefix sums over the start bitmask (popcount per 64-byte word), then offsets emission.
// Replaces: the Vec<Split> a PreTokenizedString accumulates (tokenizer/pre_tokenizer.rs:73-103);
// here the "splits" of the whole batch are one u32 array pt_start[P+1] (pt_start[P] = n_bytes).
// =================================================================================================
// (eight words per lane: the single-workgroup scan over the workgroup totals then has a few hundred entries for a 120 MB batch and
// is one round instead of eight)
constexpr int WS_PER = 8;
// (len_dev: the mask covers a text whose length only exists on the device -- the normaliser's output, launched over the host's bound of
// three times the input: only the words that text has are read, summed and given a prefix; the others' block sums are zero)
__global__ __launch_bounds__(256) void k_words_reduce(const unsigned long long* __restrict__ mask, int64_t n_words,
                                                      uint32_t* __restrict__ bsum, const int64_t* __restrict__ len_dev) {
    __shared__ uint32_t sm[4];
    if (len_dev) n_words = min(n_words, (*len_dev >> 6) + 2);
    const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * WS_PER;
    uint32_t v = 0;
    if (w0 + WS_PER <= n_words) {
        const ulonglong2* const q = (const ulonglong2*)(mask + w0);                 // (w0 is a multiple of 8: 64-byte aligned)
#pragma unroll
        for (int k = 0; k < WS_PER / 2; ++k) { const ulonglong2 m = q[k]; v += (uint32_t)(__popcll(m.x) + __popcll(m.y)); }
    } else {
        for (int k = 0; k < WS_PER; ++k) v += (w0 + k < n_words) ? (uint32_t)__popcll(mask[w0 + k]) : 0u;
    }
    uint32_t tot;
    block256_excl_scan(v, sm, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// tile_w (not null: the start mask, with offsets requested): tile_w[k] = the word that holds set bit number k * META_TILE -- where tile k
// of k_token_meta starts reading the mask (kernels/output.hip MASKS)
__global__ __launch_bounds__(256) void k_words_down(const unsigned long long* __restrict__ mask, int64_t n_words,
                                                    const uint32_t* __restrict__ bsum, uint32_t* __restrict__ wprefix, const int64_t* __restrict__ len_dev,
                                                    uint32_t* __restrict__ tile_w) {
    __shared__ uint32_t sm[4];
    if (len_dev) n_words = min(n_words, (*len_dev >> 6) + 2);
    if ((int64_t)blockIdx.x * 256 * WS_PER >= n_words) return;      // (the whole workgroup)
    const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * WS_PER;
    uint32_t c[WS_PER], v = 0;
    const bool whole = w0 + WS_PER <= n_words;
    if (whole) {
        const ulonglong2* const q = (const ulonglong2*)(mask + w0);
#pragma unroll
        for (int k = 0; k < WS_PER / 2; ++k) { const ulonglong2 m = q[k]; c[2 * k] = (uint32_t)__popcll(m.x); c[2 * k + 1] = (uint32_t)__popcll(m.y); }
    } else {
#pragma unroll
        for (int k = 0; k < WS_PER; ++k) c[k] = (w0 + k < n_words) ? (uint32_t)__popcll(mask[w0 + k]) : 0u;
    }
#pragma unroll
    for (int k = 0; k < WS_PER; ++k) v += c[k];
    uint32_t tot;
    uint32_t run = bsum[blockIdx.x] + block256_excl_scan(v, sm, &tot);
    if (tile_w) {
        uint32_t ex = run;
#pragma unroll
        for (int k = 0; k < WS_PER; ++k) {                 // (a word holds at most 64 bits: at most one multiple of the tile lies in it)
            const uint32_t kk = (ex + (uint32_t)META_TILE - 1u) / (uint32_t)META_TILE;
            if (c[k] && kk * (uint32_t)META_TILE < ex + c[k]) tile_w[kk] = (uint32_t)(w0 + k);
            ex += c[k];
        }
    }
    if (whole) {
        uint4 o0, o1;
        o0.x = run; run += c[0]; o0.y = run; run += c[1]; o0.z = run; run += c[2]; o0.w = run; run += c[3];
        o1.x = run; run += c[4]; o1.y = run; run += c[5]; o1.z = run; run += c[6]; o1.w = run;
        ((uint4*)(wprefix + w0))[0] = o0;
        ((uint4*)(wprefix + w0))[1] = o1;
    } else {
#pragma unroll
        for (int k = 0; k < WS_PER; ++k) { if (w0 + k < n_words) wprefix[w0 + k] = run; run += c[k]; }
    }
}

// A wavefront takes 64 consecutive mask words (4 KB of text): one coalesced load of the words and their
// prefixes, then word by word (broadcast with readlane) lane l tests bit l and stores pt_start[rank] = position.
// All loads are issued up front; the per-word work is a handful of VALU ops and one masked, rank-ordered store.
__global__ __launch_bounds__(256) void k_emit_pretok(const unsigned long long* __restrict__ startmask,
                                                     const uint32_t* __restrict__ wprefix, int64_t n_bytes,
                                                     const int64_t* __restrict__ len_dev,
                                                     const int64_t* __restrict__ n_pretok,
                                                     uint32_t* __restrict__ pt_start) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave == 0 && lane == 0) pt_start[*n_pretok] = (uint32_t)(len_dev ? *len_dev : n_bytes);          // sentinel
    if (len_dev) n_bytes = min(n_bytes, *len_dev);                // (start bits exist only below the text's own length)
    const int64_t n_words = (n_bytes + 63) >> 6;
    const int64_t w0 = wave * 64;
    if (w0 >= n_words) return;
    const int64_t wi = w0 + lane;
    const unsigned long long mw = (wi < n_words) ? startmask[wi] : 0ull;
    const uint32_t pw = (wi < n_words) ? wprefix[wi] : 0u;
    const uint32_t mlo = (uint32_t)mw, mhi = (uint32_t)(mw >> 32);
    if (__ballot(mw != 0ull) == 0ull) return;                    // nothing starts in these 4 KB
    const int kmax = (int)min((int64_t)64, n_words - w0);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k = 0; k < kmax; ++k) {
        const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mhi, k) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)mlo, k);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)pw, k);
        if ((m >> lane) & 1ull) pt_start[base + (uint32_t)__popcll(m & below)] = (uint32_t)(((w0 + k) << 6) + lane);
    }
}

// exclusive end of every pre-token for the "Removed" pre-tokenizers: an end bit at byte i closes the
// pre-token that started most recently before i.  Same wavefront-per-64-words structure as k_emit_pretok.
__global__ __launch_bounds__(256) void k_emit_pretok_end(const unsigned long long* __restrict__ startmask,
                                                         const unsigned long long* __restrict__ endmask,
                                                         const uint32_t* __restrict__ wprefix, int64_t n_bytes, const int64_t* __restrict__ len_dev,
                                                         uint32_t* __restrict__ pt_end) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (len_dev) n_bytes = min(n_bytes, *len_dev);
    const int64_t n_words = (n_bytes >> 6) + 1;               // an end bit can sit at byte n_bytes
    const int64_t w0 = wave * 64;
    if (w0 >= n_words) return;
    const int64_t wi = w0 + lane;
    const unsigned long long ms = (wi < n_words) ? startmask[wi] : 0ull, me = (wi < n_words) ? endmask[wi] : 0ull;
    const uint32_t pw = (wi < n_words) ? wprefix[wi] : 0u;
    const uint32_t slo = (uint32_t)ms, shi = (uint32_t)(ms >> 32), elo = (uint32_t)me, ehi = (uint32_t)(me >> 32);
    if (__ballot(me != 0ull) == 0ull) return;
    const int kmax = (int)min((int64_t)64, n_words - w0);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k = 0; k < kmax; ++k) {
        const unsigned long long e = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ehi, k) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)elo, k);
        if (e == 0ull) continue;                              // wave-uniform
        const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)shi, k) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)slo, k);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)pw, k);
        if ((e >> lane) & 1ull) pt_end[base + (uint32_t)__popcll(m & below) - 1u] = (uint32_t)(((w0 + k) << 6) + lane);
    }
}

// doc_pt[d] = index of the first pre-token at or after the first byte of document d (d = 0..n_docs)
// chunk_lo[c] = the first document d with doc_pt[d] >= c * chunk (c = 0 .. P / chunk + 1; n_docs + 1: none; chunk = the compaction's): the
// compaction (output.hip) finds the documents that start in a chunk of pre-tokens there and writes their token offsets itself.
// Document d fills the entries of the chunks between its predecessor's first pre-token and its own.
__device__ __forceinline__ uint32_t doc_first_rank(int64_t g, int64_t n_bytes, const unsigned long long* __restrict__ startmask,
                                                   const uint32_t* __restrict__ wprefix, const int64_t* __restrict__ n_pretok) {
    if (g >= n_bytes) return (uint32_t)*n_pretok;
    const unsigned long long m = startmask[g >> 6];
    const int b = (int)(g & 63);
    return wprefix[g >> 6] + (uint32_t)__popcll(m & ((1ull << b) - 1ull));
}
// san != nullptr (the lean prologue, capi.cpp): doc_off is the CALLER's array, err holds the verdict of its validation
// (k_mark_doc_starts, finished: an earlier launch), and this kernel doubles as k_sanitize_csr -- it reads the array through the same
// rule and writes the validated copy the later stages (offsets, epilogues) read.
__global__ void k_doc_first_pretok(const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t n_bytes,
                                   const unsigned long long* __restrict__ startmask, const uint32_t* __restrict__ wprefix,
                                   const int64_t* __restrict__ n_pretok, uint32_t* __restrict__ doc_pt, uint32_t* __restrict__ chunk_lo, uint32_t chunk,
                                   const int* __restrict__ err, int64_t* __restrict__ san) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    const bool bad = san && (*err & ERR_BAD_OFFSETS) != 0;
    const int64_t g = bad ? (d == n_docs ? n_bytes : 0) : doc_off[d];
    if (san) san[d] = g;
    const uint32_t r = doc_first_rank(g, n_bytes, startmask, wprefix, n_pretok);
    doc_pt[d] = r;
    const uint32_t c_hi = r / chunk;
    uint32_t c = d ? doc_first_rank(bad ? 0 : doc_off[d - 1], n_bytes, startmask, wprefix, n_pretok) / chunk + 1u : 0u;
    for (; c <= c_hi; ++c) chunk_lo[c] = (uint32_t)d;
    if (d == n_docs) chunk_lo[c_hi + 1u] = (uint32_t)n_docs + 1u;
}

// data[i] += delta (rebasing a CSR slice)
__global__ void k_add_u32(uint32_t* __restrict__ data, int64_t n, uint32_t delta) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += delta;
}
__global__ void k_add_i64(int64_t* __restrict__ data, int64_t n, int64_t delta) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += delta;
}
