// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Whitespace / WhitespaceSplit / BertPreTokenizer (the per-lane bit-parallel kernel; round 1's lane-per-byte tile kernel is gone).

// =================================================================================================
// The pre-tokenizers whose split rule only looks at the class of a code point and
// of its predecessor; matches are kept, everything else is REMOVED, so a second bitmask marks ends.
//   PT_WHITESPACE       \w+|[^\w\s]+ via Invert + Removed            (pre_tokenizers/whitespace.rs:20-29)
//   PT_WHITESPACE_SPLIT char::is_whitespace, Removed                  (whitespace.rs:35-41)
//   PT_BERT             whitespace Removed, then is_bert_punc Isolated (pre_tokenizers/bert.rs:5-17)
// class: 0 = removed (whitespace), 1 / 2 = run classes, 3 = isolated (each code point its own split).
//   start[i] = cls != 0 && (doc start || prev != cls || cls == 3)
//   end[i]   = prev != 0 && (doc start || text end || cls != prev || prev == 3)     (exclusive end)
// =================================================================================================
// =================================================================================================
// K_pretok_local_lane: the three pre-tokenizers, bit-parallel per lane (the scheme of k_pretok_gpt2_seq): a lane
// owns 48 bytes inside a 64-byte window, deposits one-hot class flags from a 1 KB LDS table into 64-bit masks and
// runs local_window_masks (pretok_local_core.hpp; checked on the CPU by tests/test_pretok_core.py) to get the start
// and end bits of its bytes.  Four lanes' 48-bit results are three mask words.
// =================================================================================================
template <int KIND>
__global__ __launch_bounds__(256) void k_pretok_local_lane(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                           const int64_t* __restrict__ len_dev,
                                                           const unsigned long long* __restrict__ docmask,
                                                           const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                           unsigned long long* __restrict__ startmask,
                                                           unsigned long long* __restrict__ endmask, int len_bound) {
    __shared__ uint32_t lut[SQ_LUT_COPIES * 256];
    // len_bound (with len_dev): the launch covers the host's bound of the text -- three times the input behind BertNormalizer -- but only
    // the mask words of the text's own length are written ((len >> 6) + 2 of them: what the scan, the lookup and the emit kernels read);
    // a workgroup wholly behind them is gone before it fills its table
    if (len_bound && (int64_t)blockIdx.x * 256 * PLW_MAIN > *len_dev + 192) return;
    {
        const uint32_t f = local_byte_flags<KIND>(threadIdx.x);
#pragma unroll
        for (int c = 0; c < SQ_LUT_COPIES; ++c) lut[c * 256 + threadIdx.x] = f;
    }
    __syncthreads();
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const int64_t n_words_host = (n_bytes_host >> 6) + 1;
    const int64_t Lg = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t a = Lg * PLW_MAIN;                         // first byte this lane decides
    const int64_t base = a - PLW_HALO;                       // window = [base, base + 64)
    unsigned long long st = 0, en = 0;
    if (a <= n_bytes) {                                      // "<=": the end bit of the last pre-token sits at byte n_bytes
        uint32_t w[16];
        {
            SqChunk c0{0, 0, 0, 0};
            if (base >= 0) c0 = *(const SqChunk*)(text + base);
            else { const uint2 t = *(const uint2*)text; c0.c = t.x; c0.d = t.y; }
            const SqChunk c1 = *(const SqChunk*)(text + base + 16), c2 = *(const SqChunk*)(text + base + 32),
                          c3 = *(const SqChunk*)(text + base + 48);
            w[0] = c0.a; w[1] = c0.b; w[2] = c0.c; w[3] = c0.d; w[4] = c1.a; w[5] = c1.b; w[6] = c1.c; w[7] = c1.d;
            w[8] = c2.a; w[9] = c2.b; w[10] = c2.c; w[11] = c2.d; w[12] = c3.a; w[13] = c3.b; w[14] = c3.c; w[15] = c3.d;
        }
        LocalWindow m;
        const int vlo = base < 0 ? (int)-base : 0;
        const int64_t rem = n_bytes - base;                  // >= PLW_HALO
        m.V = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & (~0ull << vlo);
        m.END = rem < 64 ? (1ull << rem) : 0ull;
        if (base < 0) m.D = docmask[0] << PLW_HALO;
        else {
            const int64_t wi = base >> 6;
            const int sh = (int)(base & 63);
            m.D = docmask[wi] >> sh;
            if (sh && wi + 1 < n_words_host) m.D |= docmask[wi + 1] << (64 - sh);
        }
        const uint32_t* my_lut = lut + (threadIdx.x & (SQ_LUT_COPIES - 1)) * 256;
        unsigned long long C1 = 0, C2 = 0, C3 = 0, CC = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * g + j;
                acc |= my_lut[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu] << j;
            }
            C1 |= (unsigned long long)(acc & 0xFFu) << (8 * g);
            C2 |= (unsigned long long)((acc >> 8) & 0xFFu) << (8 * g);
            C3 |= (unsigned long long)((acc >> 16) & 0xFFu) << (8 * g);
            CC |= (unsigned long long)(acc >> 24) << (8 * g);
        }
        m.MU = C1 & C2 & C3;                                 // a multi-byte lead carries all three class flags
        m.C1 = C1 & ~m.MU; m.C2 = C2 & ~m.MU; m.C3 = C3 & ~m.MU;
        m.C = CC;
        uint64_t s64, e64;
        local_window_masks<KIND>(m, text, base, uc1, uc2, &s64, &e64);
        st = (s64 >> PLW_HALO) & ((1ull << PLW_MAIN) - 1ull);
        en = (e64 >> PLW_HALO) & ((1ull << PLW_MAIN) - 1ull);
    }
    // four lanes' 48-bit results are three 64-bit mask words
    const unsigned long long st_n = __shfl_down(st, 1, 64), en_n = __shfl_down(en, 1, 64);
    const int q = (int)(threadIdx.x & 3);
    if (q < 3) {
        const int64_t word = 3 * (Lg >> 2) + q;
        if (word < (len_bound ? min(n_words_host, (n_bytes >> 6) + 2) : n_words_host)) {
            startmask[word] = (st >> (16 * q)) | (st_n << (PLW_MAIN - 16 * q));
            endmask[word] = (en >> (16 * q)) | (en_n << (PLW_MAIN - 16 * q));
        }
    }
}
