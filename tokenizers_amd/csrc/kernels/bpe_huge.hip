// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  BPE merges of pre-tokens beyond the LDS path.

// =================================================================================================
// K_bpe_merge_huge: pre-tokens longer than LONG_PT_MAX bytes (e.g. a 1 MB run of letters).  One workgroup per
// pre-token, the Symbol list (c, prev, next) and the cached pair ranks live in a global scratch slab, and the
// heap of models/bpe/word.rs:163-180 becomes a two-level minimum: one (rank, pos) minimum per 64-symbol chunk,
// reduced across the workgroup every round; a merge recomputes at most three chunk minima.  Rounds cost
// O(len / 16384 + 64) loads per lane, so a 1 MB word finishes in seconds instead of O(len^2).
// =================================================================================================
constexpr int HUGE_CHUNK = 64;

__device__ __forceinline__ unsigned long long huge_chunk_min(const uint32_t* __restrict__ rnk, uint32_t c, uint32_t len) {
    unsigned long long best = ~0ull;
    uint32_t lo = c * HUGE_CHUNK, hi = min(lo + (uint32_t)HUGE_CHUNK, len);
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t r = rnk[i];
        if (r != RANK_NONE) { unsigned long long k = ((unsigned long long)r << 32) | i; best = k < best ? k : best; }
    }
    return best;
}

__global__ __launch_bounds__(256) void k_bpe_merge_huge(DevTables t, const uint8_t* __restrict__ text,
                                                        const QItem* __restrict__ q, uint4* __restrict__ rows, uint32_t row_base,
                                                        const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list,
                                                        uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end,
                                                        uint32_t* __restrict__ scratch, unsigned long long scratch_words,
                                                        unsigned long long* __restrict__ scratch_used, int* __restrict__ err) {
    __shared__ unsigned long long red[4];
    __shared__ unsigned long long base_s;
    __shared__ uint32_t touched[3];
    __shared__ uint32_t cnt_s;
    __shared__ uint32_t wcnt[4];
    const int tid = (int)threadIdx.x;
    __shared__ uint32_t first_s;
    const uint32_t n = *n_list;
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        const uint32_t p = list[item];                               // position in the long queue: names the result row
        const QItem it = q[p];
        const uint32_t s = it.s, len = qitem_len(it.len);      // (far beyond the claims' 32 bytes: the flag is never set here)
        const uint32_t n_chunks = (len + HUGE_CHUNK - 1) / HUGE_CHUNK;
        const unsigned long long need = 5ull * len + 2ull * n_chunks + 1ull;     // u32 words (+1 to 8-byte-align cmin)
        __syncthreads();
        if (tid == 0) base_s = atomicAdd(scratch_used, need);
        __syncthreads();
        if (base_s + need > scratch_words) {
            if (tid == 0) { atomicOr(err, ERR_PRETOKEN_TOO_LONG); rows[row_base + p] = make_uint4(0u, 0u, 0u, 0u); }
            continue;
        }
        uint32_t* sym = scratch + base_s;
        uint32_t* rnk = sym + len;
        uint32_t* nid = rnk + len;
        uint32_t* nxt = nid + len;
        uint32_t* prv = nxt + len;
        unsigned long long* cmin = (unsigned long long*)(prv + len + ((base_s + 5ull * len) & 1ull));   // 8-byte aligned
        for (uint32_t i = tid; i < len; i += 256) {
            sym[i] = t.byte_id[text[s + i]];
            nxt[i] = (i + 1 < len) ? i + 1 : 0xFFFFFFFFu;
            prv[i] = (i > 0) ? i - 1 : 0xFFFFFFFFu;
        }
        __syncthreads();
        for (uint32_t i = tid; i < len; i += 256) {
            uint32_t r = RANK_NONE, ni = 0;
            if (i + 1 < len) merge_probe(t, sym[i], sym[i + 1], &r, &ni);
            rnk[i] = r;
            nid[i] = ni;
        }
        __syncthreads();
        for (uint32_t c = tid; c < n_chunks; c += 256) cmin[c] = huge_chunk_min(rnk, c, len);
        __syncthreads();
        while (true) {
            unsigned long long best = ~0ull;
            for (uint32_t c = tid; c < n_chunks; c += 256) { unsigned long long k = cmin[c]; best = k < best ? k : best; }
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
                const uint32_t w = (uint32_t)best;
                const uint32_t r = nxt[w], rn = nxt[r], pw = prv[w];
                sym[w] = nid[w];
                rnk[r] = RANK_NONE;
                sym[r] = 0xFFFFFFFFu;                          // removed
                nxt[w] = rn;
                if (rn != 0xFFFFFFFFu) prv[rn] = w;
                uint32_t r1 = RANK_NONE, n1 = 0, r2 = RANK_NONE, n2 = 0;
                if (pw != 0xFFFFFFFFu) merge_probe(t, sym[pw], sym[w], &r1, &n1);
                if (rn != 0xFFFFFFFFu) merge_probe(t, sym[w], sym[rn], &r2, &n2);
                if (pw != 0xFFFFFFFFu) { rnk[pw] = r1; nid[pw] = n1; }
                rnk[w] = r2;
                nid[w] = n2;
                touched[0] = w / HUGE_CHUNK;
                touched[1] = r / HUGE_CHUNK;
                touched[2] = (pw != 0xFFFFFFFFu) ? pw / HUGE_CHUNK : w / HUGE_CHUNK;
            }
            __syncthreads();
            if (tid < 3) {
                uint32_t c = touched[tid];
                bool dup = (tid == 1 && c == touched[0]) || (tid == 2 && (c == touched[0] || c == touched[1]));
                if (!dup) cmin[c] = huge_chunk_min(rnk, c, len);
            }
            __syncthreads();
        }
        // ordered emission of the surviving symbols
        if (tid == 0) cnt_s = 0;
        __syncthreads();
        for (uint32_t base = 0; base < len; base += 256) {
            uint32_t i = base + tid;
            bool alive = i < len && sym[i] != 0xFFFFFFFFu;
            uint64_t bm = __ballot(alive);
            if ((tid & 63) == 0) wcnt[tid >> 6] = (uint32_t)__popcll(bm);
            __syncthreads();
            uint32_t off = cnt_s;
            for (int w = 0; w < (tid >> 6); ++w) off += wcnt[w];
            if (alive) {
                uint32_t j = off + (uint32_t)mbcnt64(bm);
                if (j == 0) first_s = sym[i];
                else tmp_ids[s + j] = sym[i];
                if (tmp_end) { uint32_t e = nxt[i]; tmp_end[s + j] = (e == 0xFFFFFFFFu) ? len : e; }
            }
            __syncthreads();
            if (tid == 0) cnt_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        if (tid == 0) rows[row_base + p] = make_uint4(first_s | (ROW_CNT_MORE << ROW_CNT_SHIFT), s, cnt_s, 0u);
    }
}
