// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Llama-3 split: tile kernel, per-lane kernel, sequential per-document matcher.

// =================================================================================================
// K_pretok_llama3: the Llama-3 / tiktoken cl100k-style split
//   (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
// Replaces Split::pre_tokenize (pre_tokenizers/split.rs:96-104, Oniguruma find_iter, Isolated) inside
// Sequence[Split, ByteLevel(use_regex=false)] (sequence.rs:40-45).  Every byte is covered by a match, so
// the output is again "a match starts at byte i".  Unlike GPT-2 the predicate is RUN-local, not
// window-local: digit runs are cut every 3 from the run start, a whitespace run
// is cut after its LAST CR/LF and before its last char, an O-run swallows the CR/LFs that follow it.
// Fast path (this kernel): one lane per byte, 2 KB tile + 128 B halo in LDS, every run question answered
// by a bounded walk over the run; a lane whose walk leaves the staged region records its byte position
// and the whole document is redone by k_pretok_llama3_slow (sequential, exact for any run length).
// classes: 0 other, 1 letter, 2 number, 3 whitespace (not CR/LF), 4 CR/LF
// =================================================================================================
constexpr int L3_HALO = 128;
constexpr int L3_R = PT_TILE + 2 * L3_HALO;
constexpr uint32_t L3_CLS = 7, L3_LEAD = 8, L3_DOC = 16, L3_VALID = 32, L3_SP = 64;

__device__ __forceinline__ uint32_t cls_llama3(uint32_t cp, const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2) {
    if (cp == '\r' || cp == '\n') return 4;
    return cls_lns(cp, uc1, uc2);
}

struct L3View {
    const uint8_t* sb;   // staged bytes
    const uint8_t* si;   // info
    __device__ __forceinline__ int prev(int k) const {         // lead byte of the previous code point (k > 0)
        int j = k - 1;
        if (!(si[j] & L3_LEAD)) { --j; if (!(si[j] & L3_LEAD)) { --j; if (!(si[j] & L3_LEAD)) --j; } }
        return j;
    }
    __device__ __forceinline__ int next(int k) const {         // first byte after the code point at k
        uint32_t b = sb[k];
        return k + (b < 0x80u ? 1 : b < 0xE0u ? 2 : b < 0xF0u ? 3 : 4);
    }
    // is there a previous code point in the same document?
    __device__ __forceinline__ bool has_prev(int k) const { return !(si[k] & L3_DOC); }
    // is position k (a lead byte index) inside the text and the same document as its predecessor?
    __device__ __forceinline__ bool inside(int k) const { return (si[k] & L3_VALID) && !(si[k] & L3_DOC); }
};

// contraction letter at lead byte k: returns 's','t','m','d','r','v','e','l' (any ASCII letter, lower case) or 0; *nx = next position.
// fold: the literals are case-insensitive ((?i:...): upper case matches, and U+017F folds to 's'); else only lower case does
__device__ __forceinline__ uint32_t l3_letter(const L3View& v, int k, int* nx, bool fold) {
    uint32_t b = v.sb[k];
    if (fold && b == 0xC5u && v.sb[k + 1] == 0xBFu) { *nx = k + 2; return 's'; }       // U+017F LATIN SMALL LETTER LONG S folds to 's'
    *nx = k + 1;
    uint32_t f = fold ? (b | 0x20u) : b;
    return (f - 'a' < 26u) ? f : 0u;
}

__global__ __launch_bounds__(256) void k_pretok_llama3(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                       const int64_t* __restrict__ len_dev,
                                                       const unsigned long long* __restrict__ docmask,
                                                       const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                       unsigned long long* __restrict__ startmask,
                                                       unsigned long long* __restrict__ slowmask, const unsigned long long* __restrict__ tileflags, SplitRule rule) {
    // rule: which member of the family (tables.hpp SplitRule; this kernel: split_rule_fast ones -- the contraction alternative absent /
    // case-insensitive / case-sensitive, digit runs cut every 1, 2, 3 code points or not at all)
    // tileflags (refine): k_pretok_llama3_lane has run and left a bit for every tile in which it left bytes undecided; a workgroup takes
    // the 64 tiles of one flag word and redoes the flagged ones (natural text: none -- a launch of one load a workgroup, where a
    // workgroup per tile testing the tile's 32 slow-mask words was 0.022 ms of C4's step).  Null: every tile, one a workgroup.
    unsigned long long todo = 1ull;
    if (tileflags) { todo = tileflags[blockIdx.x]; if (!todo) return; }            // (uniform)
    for (; todo; todo &= todo - 1ull) {
    const int64_t tile = tileflags ? (int64_t)blockIdx.x * 64 + __builtin_ctzll(todo) : (int64_t)blockIdx.x;
    __shared__ __attribute__((aligned(16))) uint8_t sb[L3_R + 8];
    __shared__ uint8_t si[L3_R + 8];
    __shared__ uint8_t sc[L3_R + 8];     // con: number of letters (1|2) swallowed by a contraction starting at this apostrophe
    __shared__ unsigned long long sdoc[L3_R / 64 + 2];
    const int tid = (int)threadIdx.x;
    const int64_t t0 = tile * PT_TILE;
    const int64_t r0 = t0 - L3_HALO;
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    {
        uint32_t* sb32 = (uint32_t*)sb;                          // r0 is a multiple of 4: aligned dword staging
        for (int k = tid; k < (L3_R + 8) / 4; k += 256) {
            int64_t g = r0 + 4 * (int64_t)k;
            uint32_t v = 0;
            if (g >= 0 && g + 4 <= n_bytes) v = *(const uint32_t*)(text + g);
            else if (g + 4 > 0 && g < n_bytes) {
                for (int q = 0; q < 4; ++q)
                    if (g + q >= 0 && g + q < n_bytes) v |= (uint32_t)text[g + q] << (8 * q);
            }
            sb32[k] = v;
        }
        if (tid < L3_R / 64 + 2) {                               // doc-start words covering [t0 - 128, ...)
            int64_t w = (r0 >> 6) + tid;
            sdoc[tid] = (w >= 0 && (w << 6) < n_bytes_host + 64) ? docmask[w] : 0ull;
        }
    }
    __syncthreads();
    for (int k = tid; k < L3_R + 8; k += 256) {
        int64_t g = r0 + k;
        uint32_t info = 0;
        if (k < L3_R && g >= 0 && g < n_bytes) {
            uint32_t b = sb[k];
            info = L3_VALID;
            if ((sdoc[k >> 6] >> (k & 63)) & 1ull) info |= L3_DOC;      // r0 is a multiple of 64: bit k of the staged words
            if ((b & 0xC0u) != 0x80u) {
                uint32_t len;
                uint32_t cp = utf8_at(sb, k, &len);
                info |= L3_LEAD | cls_llama3(cp, uc1, uc2);
                if (b == 0x20u) info |= L3_SP;
            }
        }
        si[k] = (uint8_t)info;
    }
    __syncthreads();
    L3View v{sb, si};
    // contraction literals: fire only where the apostrophe is itself a match start
    for (int k = tid; k < L3_R; k += 256) {
        uint32_t con = 0;
        if (rule.contr && k >= 4 && k < L3_R - 8 && sb[k] == '\'' && (si[k] & L3_VALID)) {
            const bool fold = rule.contr == 1;
            int k1 = k + 1, k2, k3;
            if (v.inside(k1) && (si[k1] & L3_LEAD)) {
                uint32_t a = l3_letter(v, k1, &k2, fold);
                uint32_t lit = 0;
                if (a == 's' || a == 't' || a == 'm' || a == 'd') lit = 1;
                else if ((a == 'r' || a == 'v' || a == 'l') && v.inside(k2) && (si[k2] & L3_LEAD) && sb[k2] < 0x80u) {
                    uint32_t b2 = l3_letter(v, k2, &k3, fold);
                    if ((a == 'l') ? (b2 == 'l') : (b2 == 'e')) lit = 2;
                }
                if (lit) {
                    bool cond;
                    if (!v.has_prev(k)) cond = true;
                    else {
                        uint32_t pi = si[v.prev(k)], pc = pi & L3_CLS;
                        cond = (pc == 1 || pc == 2 || pc == 4 || (pc == 3 && !(pi & L3_SP)));
                    }
                    if (cond) con = lit;
                }
            }
        }
        sc[k] = (uint8_t)con;
    }
    __syncthreads();
    for (int it = 0; it < PT_TILE / 256; ++it) {
        const int k = L3_HALO + it * 256 + tid;
        const int64_t g = t0 + it * 256 + tid;
        const uint32_t info = si[k];
        bool start = false, unresolved = false;
        if ((info & (L3_VALID | L3_LEAD)) == (L3_VALID | L3_LEAD)) {
            const uint32_t c = info & L3_CLS;
            if (info & L3_DOC) start = true;
            else {
                const int p1 = v.prev(k);
                const uint32_t i1 = si[p1], c1 = i1 & L3_CLS;
                // eaten(x): x is a letter swallowed by a contraction (the apostrophe is 1 or 2 code points back)
                bool eaten = false, eaten_prev = false;
                if (c == 1) {
                    if (sb[p1] == '\'' && sc[p1] >= 1) eaten = true;
                    else if (c1 == 1 && v.has_prev(p1)) { int p2 = v.prev(p1); if (sb[p2] == '\'' && sc[p2] == 2) eaten = true; }
                }
                if (c1 == 1 && v.has_prev(p1)) {
                    int p2 = v.prev(p1);
                    if (sb[p2] == '\'' && sc[p2] >= 1) {
                        // p1 is the first swallowed letter; it is the LAST one iff the literal has 1 letter
                        eaten_prev = (sc[p2] == 1);
                    } else if ((si[p2] & L3_CLS) == 1 && v.has_prev(p2)) {
                        int p3 = v.prev(p2);
                        if (sb[p3] == '\'' && sc[p3] == 2) eaten_prev = true;
                    }
                }
                if (c == 1) {
                    if (eaten) start = false;
                    else if (eaten_prev) start = true;
                    else if (c1 == 1) start = false;
                    else {
                        // first letter of a run: the previous char joins as the optional prefix iff it is a match start
                        bool prefixable = false;
                        if (c1 == 3) prefixable = true;
                        else if (c1 == 0 && sc[p1] == 0) {
                            if (!v.has_prev(p1)) prefixable = true;
                            else { uint32_t i2 = si[v.prev(p1)]; prefixable = !((i2 & L3_CLS) == 0 || (i2 & L3_SP)); }
                        }
                        start = !prefixable;
                    }
                } else if (c == 0) {
                    start = !(c1 == 0 || (i1 & L3_SP));
                    if (eaten_prev) start = true;
                } else if (c == 2) {
                    // position inside the digit run, mod digit_max (1: every digit starts a match; 0: \p{N}+, the run's first one does)
                    if (rule.digit_max == 1) start = true;
                    else if (rule.digit_max == 0) start = c1 != 2;
                    else {
                        int cnt = 0, j = k;
                        bool ok = true;
                        while (true) {
                            if (!v.has_prev(j)) break;
                            int pj = v.prev(j);
                            if (pj < 4) { ok = false; break; }
                            if ((si[pj] & L3_CLS) != 2) break;
                            j = pj;
                            ++cnt;
                        }
                        if (!ok) unresolved = true;
                        start = (cnt % (int)rule.digit_max) == 0;
                    }
                } else {
                    // whitespace run [q, e); q' = q + leading CR/LFs swallowed by a preceding O-run match
                    int q = k;
                    bool ok = true;
                    while (v.has_prev(q)) {
                        int pq = v.prev(q);
                        if (pq < 4) { ok = false; break; }
                        if ((si[pq] & L3_CLS) < 3) break;
                        q = pq;
                    }
                    bool absorb = ok && v.has_prev(q) && (si[v.prev(q)] & L3_CLS) == 0;
                    int qe = q;                                   // effective run start
                    if (absorb) while (qe < L3_R - 4 && v.inside(qe) && (si[qe] & L3_CLS) == 4) qe = v.next(qe);
                    if (qe >= L3_R - 4) ok = false;
                    // run end and last CR/LF at or after k
                    int e = k, last_crlf = -1, lastcp = k;
                    while (true) {
                        if (e >= L3_R - 4) { ok = false; break; }
                        if (e != k && !v.inside(e)) break;           // document / text end
                        if ((si[e] & L3_CLS) < 3) break;
                        if ((si[e] & L3_CLS) == 4) last_crlf = e;
                        lastcp = e;
                        e = v.next(e);
                    }
                    bool at_doc_end = ok && !v.inside(e);
                    if (!ok) unresolved = true;
                    else if (k < qe) start = false;                  // swallowed by the O-run's [\r\n]* tail
                    else {
                        // is there a CR/LF in [qe, k)?  only needed to know whether k == (last CR/LF)+1
                        bool crlf_before = (c1 == 4) && p1 >= qe;    // previous char is a CR/LF inside the run
                        if (k == qe) start = true;
                        else if (last_crlf < 0 && crlf_before) start = true;          // k == lc + 1, remainder starts here
                        else if (last_crlf < 0 && k == lastcp && !at_doc_end) {
                            // last char of the run, followed by a non-space: split before it if the remainder has >= 2 chars
                            // remainder start r = (last CR/LF before k) + 1 or qe; k > r  <=>  previous char is in the run,
                            // after qe, and is not a CR/LF
                            start = (p1 >= qe) && (c1 == 3);
                        } else start = false;
                    }
                }
            }
        }
        uint64_t m = __ballot(start), mu = __ballot(unresolved);
        if ((tid & 63) == 0 && g <= n_bytes_host) { startmask[g >> 6] = m; slowmask[g >> 6] = mu; }
    }
    __syncthreads();                                             // (the next flagged tile stages into the same LDS)
    }
}

// =================================================================================================
// K_pretok_llama3_lane: the same split, bit-parallel PER LANE (the GPT-2 kernel's scheme): a lane owns 48 bytes
// inside a 64-byte window (8 bytes of context on each side, four 16-byte loads), deposits one-hot byte
// flags from a small LDS table into 64-bit masks and runs l3_window_starts (pretok_l3_core.hpp) -- the mask algebra
// that tests/test_pretok_core.py checks on the CPU, function for function, against a sequential matcher.  Bytes whose run
// leaves the window are reported in slowmask; k_pretok_llama3 (refine mode) redoes only the tiles that have any.
// =================================================================================================
// CS: the case-split members of the family (o200k, tekken; tables.hpp split_rule_fast_cs) -- l3_window_starts_cs with the case classes ucc1 /
// ucc2 and bit 5 of every byte (the case of an ASCII letter); what it leaves undecided goes to the sequential matcher by sentence.
// LEAD: the lead-byte mask of the same text rides along (char offsets over a text read as it came: kernels/pretok_gpt2.hip)
template <bool CS, bool LEAD = false>
__global__ __launch_bounds__(256) void k_pretok_llama3_lane(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                            const int64_t* __restrict__ len_dev,
                                                            const unsigned long long* __restrict__ docmask,
                                                            const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                            unsigned long long* __restrict__ startmask,
                                                            unsigned long long* __restrict__ slowmask, SplitRule rule,
                                                            const uint16_t* __restrict__ ucc1, const uint8_t* __restrict__ ucc2,
                                                            unsigned long long* __restrict__ tileflags, unsigned long long* __restrict__ leadmask) {
    __shared__ uint2 lut[SQ_LUT_COPIES * 256];
    {
        const L3Flags f = l3_byte_flags(threadIdx.x);
#pragma unroll
        for (int c = 0; c < SQ_LUT_COPIES; ++c) lut[c * 256 + threadIdx.x] = make_uint2(f.x, f.y);
    }
    __syncthreads();
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const int64_t n_words_host = (n_bytes_host >> 6) + 1;
    const int64_t Lg = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t a = Lg * L3W_MAIN;                         // first byte this lane decides
    const int64_t base = a - L3W_HALO;                       // window = [base, base + 64), 8-byte aligned
    unsigned long long st = 0, un = 0, ld = 0;
    if (a < n_bytes) {
        uint32_t w[16];
        {
            // four 16-byte loads (8-byte aligned: gfx950 takes dwordx4 at any alignment); only lane 0's window starts before the text
            SqChunk c0{0u, 0u, 0u, 0u};
            if (base >= 0) c0 = *(const SqChunk*)(text + base);
            else { const uint2 t = *(const uint2*)text; c0.c = t.x; c0.d = t.y; }
            const SqChunk c1 = *(const SqChunk*)(text + base + 16), c2 = *(const SqChunk*)(text + base + 32), c3 = *(const SqChunk*)(text + base + 48);
            w[0] = c0.a; w[1] = c0.b; w[2] = c0.c; w[3] = c0.d; w[4] = c1.a; w[5] = c1.b; w[6] = c1.c; w[7] = c1.d;
            w[8] = c2.a; w[9] = c2.b; w[10] = c2.c; w[11] = c2.d; w[12] = c3.a; w[13] = c3.b; w[14] = c3.c; w[15] = c3.d;
        }
        L3Window m;
        const int vlo = base < 0 ? (int)-base : 0;
        const int64_t rem = n_bytes - base;
        m.V = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & (~0ull << vlo);
        if (base < 0) m.D = docmask[0] << L3W_HALO;
        else {
            const int64_t wi = base >> 6;
            const int sh = (int)(base & 63);
            m.D = docmask[wi] >> sh;
            if (sh && wi + 1 < n_words_host) m.D |= docmask[wi + 1] << (64 - sh);
        }
        const uint2* my_lut = lut + (threadIdx.x & (SQ_LUT_COPIES - 1)) * 256;
        m.L = m.N = m.W = m.R = m.SP = m.C = m.AP = m.MU = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            uint32_t accA = 0, accB = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * g + j;
                const uint2 e = my_lut[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu];
                accA |= e.x << j;
                accB |= e.y << j;
            }
            m.L |= (unsigned long long)(accA & 0xFFu) << (8 * g);
            m.N |= (unsigned long long)((accA >> 8) & 0xFFu) << (8 * g);
            m.W |= (unsigned long long)((accA >> 16) & 0xFFu) << (8 * g);
            m.R |= (unsigned long long)(accA >> 24) << (8 * g);
            m.SP |= (unsigned long long)(accB & 0xFFu) << (8 * g);
            m.C |= (unsigned long long)((accB >> 8) & 0xFFu) << (8 * g);
            m.AP |= (unsigned long long)((accB >> 16) & 0xFFu) << (8 * g);
            m.MU |= (unsigned long long)(accB >> 24) << (8 * g);
        }
        m.L &= m.V; m.N &= m.V; m.W &= m.V; m.R &= m.V; m.SP &= m.V; m.C &= m.V; m.AP &= m.V; m.MU &= m.V;
        if constexpr (LEAD) ld = ((m.V & ~m.C) >> L3W_HALO) & ((1ull << L3W_MAIN) - 1ull);
        uint64_t s64, u64;
        if constexpr (CS) {
            // bit 5 of the 64 bytes: four bytes a multiply ((x >> 5) & 0x01010101 gathers through 0x01020408 into bits 24..27)
            unsigned long long b5 = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) b5 |= (unsigned long long)((((w[k] >> 5) & 0x01010101u) * 0x01020408u) >> 24 & 0xFu) << (4 * k);
            m.B5 = b5;
            l3_window_starts_cs(m, text, base, uc1, uc2, ucc1, ucc2, &s64, &u64, rule);
        } else
        l3_window_starts(m, text, base, uc1, uc2, &s64, &u64, rule);
        st = (s64 >> L3W_HALO) & ((1ull << L3W_MAIN) - 1ull);
        un = (u64 >> L3W_HALO) & ((1ull << L3W_MAIN) - 1ull);
        if (tileflags && un) {                                   // (rare) the tile kernel's work list: the tiles of my first and last undecided byte
            const int64_t t_lo = (a + __builtin_ctzll(un)) / PT_TILE, t_hi = (a + 63 - __builtin_clzll(un)) / PT_TILE;
            atomicOr(&tileflags[t_lo >> 6], 1ull << (t_lo & 63));
            if (t_hi != t_lo) atomicOr(&tileflags[t_hi >> 6], 1ull << (t_hi & 63));
        }
    }
    // four lanes' 48-bit results are three 64-bit mask words
    const unsigned long long st_n = __shfl_down(st, 1, 64), un_n = __shfl_down(un, 1, 64);
    const int q = (int)(threadIdx.x & 3);
    if (q < 3) {
        const int64_t word = 3 * (Lg >> 2) + q;
        if (word < n_words_host) {
            startmask[word] = (st >> (16 * q)) | (st_n << (L3W_MAIN - 16 * q));
            slowmask[word] = (un >> (16 * q)) | (un_n << (L3W_MAIN - 16 * q));
        }
    }
    if constexpr (LEAD) {
        const unsigned long long ld_n = __shfl_down(ld, 1, 64);
        if (q < 3) {
            const int64_t word = 3 * (Lg >> 2) + q;
            if (word < n_words_host) leadmask[word] = (ld >> (16 * q)) | (ld_n << (L3W_MAIN - 16 * q));
        }
    }
}

// Slow path: documents containing a run the tile kernel could not resolve are matched sequentially,
// alternative by alternative, one lane per document; the document's bits of the start mask are rewritten.
__device__ __forceinline__ uint32_t l3_dec(const uint8_t* __restrict__ s, int64_t i, int64_t n, int* len) {
    uint32_t b = s[i];
    if (b < 0x80u) { *len = 1; return b; }
    if (b < 0xE0u && i + 1 < n) { *len = 2; return ((b & 0x1Fu) << 6) | (s[i + 1] & 0x3Fu); }
    if (b < 0xF0u && i + 2 < n) { *len = 3; return ((b & 0x0Fu) << 12) | ((s[i + 1] & 0x3Fu) << 6) | (s[i + 2] & 0x3Fu); }
    if (i + 3 < n) { *len = 4; return ((b & 0x07u) << 18) | ((s[i + 1] & 0x3Fu) << 12) | ((s[i + 2] & 0x3Fu) << 6) | (s[i + 3] & 0x3Fu); }
    *len = 1;
    return 0xFFFDu;
}
// What the sequential matcher needs besides the text: the class tables and the member of the family (tables.hpp SplitRule)
struct L3Seq {
    const uint16_t* uc1;
    const uint8_t* uc2;
    const uint16_t* ucc1;        // case classes (rule.letters == 2), else null
    const uint8_t* ucc2;
    SplitRule rule;
};
__device__ __forceinline__ uint32_t l3_case(const L3Seq& q, uint32_t cp) {
    return cp < 0x110000u ? (uint32_t)q.ucc2[((uint32_t)q.ucc1[cp >> 8] << 8) | (cp & 255u)] : 0u;
}
// 'S|'T|'RE|'VE|'M|'LL|'D at the apostrophe s[i] (fold: (?i:...) -- Unicode simple case folding: upper case, and U+017F for 's'):
// the end of the literal, or i when none starts there
__device__ int64_t l3_contraction(const uint8_t* __restrict__ s, int64_t i, int64_t n, bool fold) {
    if (i + 1 >= n || s[i] != '\'') return i;
    int l1, l2 = 0;
    const uint32_t a = l3_dec(s, i + 1, n, &l1);
    const uint32_t af = (fold && a == 0x17Fu) ? 's' : ((a < 0x80u && ((fold ? (a | 0x20u) : a) - 'a') < 26u) ? (fold ? (a | 0x20u) : a) : 0u);
    const int64_t p2 = i + 1 + l1;
    uint32_t bf = 0;
    if (p2 < n) { const uint32_t b = l3_dec(s, p2, n, &l2); bf = (b < 0x80u && ((fold ? (b | 0x20u) : b) - 'a') < 26u) ? (fold ? (b | 0x20u) : b) : 0u; }
    if (af == 's' || af == 't' || af == 'm' || af == 'd') return p2;
    if (p2 < n && (((af == 'r' || af == 'v') && bf == 'e') || (af == 'l' && bf == 'l'))) return p2 + l2;
    return i;
}
// The case-split letter alternatives of o200k / tekken from p on, without their optional prefix char:
//   A  [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+        B  [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*
// With u = upper only (Lu, Lt), l = lower only (Ll), b = both (Lm, Lo, M): the greedy U* takes the longest [ub]* run; if an l follows, A is
// that run + the longest [lb]+ behind it; else A backtracks to the LAST b of the run (its L+ is that one char: only u's follow it) --
// the trailing u's are left for the next match; a run of u's alone fails A and is B's.  *kind: 1 A matched, 2 only B, 0 neither.
__device__ int64_t l3_case_run(const L3Seq& q, const uint8_t* __restrict__ s, int64_t p, int64_t n, int* kind) {
    int64_t j = p, last_b_end = -1;
    int l = 1;
    uint32_t cc = 0;
    while (j < n) {
        cc = l3_case(q, l3_dec(s, j, n, &l));
        if (!(cc & UCC_UPPER)) break;
        if (cc & UCC_LOWER) last_b_end = j + l;
        j += l;
    }
    if (j < n && (cc & UCC_LOWER)) {                      // (not UPPER, so lower only) U* then L+
        j += l;
        while (j < n) { int lj; if (!(l3_case(q, l3_dec(s, j, n, &lj)) & UCC_LOWER)) break; j += lj; }
        *kind = 1;
        return j;
    }
    if (last_b_end >= 0) { *kind = 1; return last_b_end; }
    *kind = j > p ? 2 : 0;
    return j;
}
// One leftmost-first match of the family's pattern at s[i] (the alternatives in the pattern's order, each with the backtracking the
// regex engine would do); returns its end
__device__ int64_t l3_match_seq(const uint8_t* __restrict__ s, int64_t i, int64_t n, const L3Seq& q) {
    const uint16_t* const uc1 = q.uc1;
    const uint8_t* const uc2 = q.uc2;
    const SplitRule rule = q.rule;
    int l;
    uint32_t c = l3_dec(s, i, n, &l);
    uint32_t cc = cls_llama3(c, uc1, uc2);
    if (rule.contr == 1 || rule.contr == 2) {
        const int64_t e = l3_contraction(s, i, n, rule.contr == 1);
        if (e > i) return e;
    }
    if (rule.letters == 0) {   // [^\r\n\p{L}\p{N}]?\p{L}+
        int64_t k = (cc == 0 || cc == 3) ? i + l : i;
        if (k < n) {
            int lk;
            uint32_t ck = l3_dec(s, k, n, &lk);
            if (cls_llama3(ck, uc1, uc2) == 1) {
                int64_t j = k + lk;
                while (j < n) { int lj; uint32_t cj = l3_dec(s, j, n, &lj); if (cls_llama3(cj, uc1, uc2) != 1) break; j += lj; }
                return j;
            }
        }
    } else {                   // prefix? A | prefix? B, each with the contraction suffix: A with the prefix char, A without, B with, B without
        int64_t e = -1;
        int k0 = 0, k1 = 0;
        int64_t e1 = -1;
        if ((cc == 0 || cc == 3) && i + l < n) e1 = l3_case_run(q, s, i + l, n, &k1);
        if (k1 == 1) e = e1;
        else {
            const int64_t e0 = l3_case_run(q, s, i, n, &k0);
            if (k0 == 1) e = e0;
            else if (k1 == 2) e = e1;
            else if (k0 == 2) e = e0;
        }
        if (e >= 0) return rule.contr == 3 ? l3_contraction(s, e, n, true) : e;
    }
    if (cc == 2) {   // \p{N}{1,k} / \p{N}+
        int64_t j = i + l;
        int cnt = 1;
        while (j < n && (rule.digit_max == 0 || cnt < (int)rule.digit_max)) { int lj; uint32_t cj = l3_dec(s, j, n, &lj); if (cls_llama3(cj, uc1, uc2) != 2) break; j += lj; ++cnt; }
        return j;
    }
    {   // " ?[^\s\p{L}\p{N}]+[\r\n]*"   (o200k: [\r\n/]*)
        int64_t k = (s[i] == ' ') ? i + 1 : i;
        if (k < n) {
            int lk;
            uint32_t ck = l3_dec(s, k, n, &lk);
            if (cls_llama3(ck, uc1, uc2) == 0) {
                int64_t j = k + lk;
                while (j < n) { int lj; uint32_t cj = l3_dec(s, j, n, &lj); if (cls_llama3(cj, uc1, uc2) != 0) break; j += lj; }
                while (j < n && (s[j] == '\r' || s[j] == '\n' || (rule.other_tail == 2 && s[j] == '/'))) ++j;
                return j;
            }
        }
    }
    if (cc >= 3) {
        int64_t j = i, last = -1, cur = i;
        while (j < n) { int lj; uint32_t cj = l3_dec(s, j, n, &lj); uint32_t k = cls_llama3(cj, uc1, uc2); if (k < 3) break; if (k == 4) last = j; cur = j; j += lj; }
        if (last >= 0) return last + 1;          // \s*[\r\n]+
        if (j >= n) return j;                    // \s+(?!\S) at end of text
        if (cur > i) return cur;                 // \s+(?!\S): all but the last whitespace char
        return j;                                // \s+
    }
    return i + l;
}

// documents with at least one unresolved byte -> slow_docs list (one lane per document)
// (n_dev: the number of sentences when it only exists on the device -- pieces between added-token matches; n_docs is then its bound)
// (slowmask null: every document that is not empty -- the members of the family only the sequential matcher serves)
__global__ void k_l3_slow_docs(const unsigned long long* __restrict__ slowmask, const int64_t* __restrict__ doc_off, int64_t n_docs, const int64_t* __restrict__ n_dev,
                               uint32_t* __restrict__ slow_docs, uint32_t* __restrict__ n_slow_docs) {
    if (n_dev) n_docs = *n_dev;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = doc_off[d], b = doc_off[d + 1];
        if (b <= a) continue;
        bool any = slowmask == nullptr;
        for (int64_t w = a >> 6; w <= (b - 1) >> 6 && !any; ++w) {
            int64_t lo = w << 6, hi = lo + 64;
            unsigned long long m = slowmask[w];
            if (a > lo) m &= ~0ull << (a - lo);
            if (b < hi) m &= ~0ull >> (hi - b);
            any = m != 0ull;
        }
        if (any) slow_docs[atomicAdd(n_slow_docs, 1u)] = (uint32_t)d;
    }
}

__global__ void k_pretok_llama3_slow(const uint8_t* __restrict__ text, const int64_t* __restrict__ doc_off,
                                     const uint32_t* __restrict__ slow_docs, const uint32_t* __restrict__ n_slow_docs,
                                     L3Seq q, unsigned long long* __restrict__ startmask) {
    const uint32_t n = *n_slow_docs;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t d = slow_docs[i];
        const int64_t a = doc_off[d], b = doc_off[d + 1];
        const uint8_t* s = text + a;
        const int64_t len = b - a;
        // clear the document's bits, then set one bit per sequential match start
        for (int64_t w = a >> 6; w <= (b - 1) >> 6; ++w) {
            int64_t lo = w << 6, hi = lo + 64;
            unsigned long long m = ~0ull;
            if (a > lo) m &= ~0ull << (a - lo);
            if (b < hi) m &= ~0ull >> (hi - b);
            atomicAnd(&startmask[w], ~m);
        }
        int64_t p = 0;
        while (p < len) {
            int64_t g = a + p;
            atomicOr(&startmask[g >> 6], 1ull << (g & 63));
            int64_t e = l3_match_seq(s, p, len, q);
            if (e <= p) e = p + 1;
            p = e;
        }
    }
}
