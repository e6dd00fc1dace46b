// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Token compaction, offsets / word ids, special tokens.

// =================================================================================================
// Token compaction: tok0[P] (+ rows) -> ids[T], the running token offset of every pre-token, the total.
// Replaces: PreTokenizedString::into_encoding + Encoding::from_iter (tokenizer/pre_tokenizer.rs:198-263,
// tokenizer/encoding.rs:541-562) for the whole batch at once.
// ONE pass over the data with decoupled look-back (results.hip), software-pipelined so that nobody waits for it:
//   front(c)   load the chunk's tok0 words and result rows, count, scan, PUBLISH the chunk's total, and assemble the chunk's
//              ids (and chunk-local token offsets) in LDS -- none of which needs the tokens in front of the chunk;
//   back(c)    resolve the look-back (by now the predecessors have long published), then LDS -> ids[] / pt_tokoff[] as whole
//              wavefront-wide stores.
// A workgroup runs front(c_next) before back(c): the look-back round trips hide behind a whole chunk of work (measured: the
// un-pipelined kernel spent 45 % of its time in them).  Two LDS buffers alternate.  A chunk whose tokens do not fit the buffer
// (> CP_STAGE: only text made of many-token pre-tokens) is scattered straight from its rows in back().
// Chunks go round robin over a grid of RESIDENT workgroups (launcher), so the owner of any earlier chunk is running or
// done: front() never waits, back(c) only needs totals that earlier front() calls publish -- no deadlock.
// =================================================================================================
constexpr int CP_NT = 256;
constexpr int CP_ITEMS = 8;                       // pre-tokens per thread
constexpr int CP_CHUNK = CP_NT * CP_ITEMS;        // 2048 (the host sizes the state array by COMPACT_CHUNK)
constexpr int CP_STAGE = 5120;                    // tokens of a chunk assembled in LDS (20 KB per buffer)

struct CpRows {
    uint4 row[CP_ITEMS];
    uint32_t cnt[CP_ITEMS];
};
__device__ __forceinline__ uint32_t cp_load(const uint32_t* __restrict__ tok0, const uint4* __restrict__ rows, const uint4* __restrict__ crows, int64_t p0, int64_t P, CpRows& r) {
    uint32_t first[CP_ITEMS];
    if (p0 + CP_ITEMS <= P) {
        const uint4 a = *(const uint4*)(tok0 + p0), b = *(const uint4*)(tok0 + p0 + 4);
        first[0] = a.x; first[1] = a.y; first[2] = a.z; first[3] = a.w; first[4] = b.x; first[5] = b.y; first[6] = b.z; first[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) first[k] = (p0 + k < P) ? tok0[p0 + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k)                     // all row loads in flight together
        r.row[k] = (first[k] & TOK_ROW) ? ((first[k] & CACHE_ROW_BIT) ? crows[first[k] & (CACHE_ROW_BIT - 1u)] : rows[first[k] & ~TOK_ROW]) : make_uint4((first[k] & TOK_ID_MASK) | (((first[k] & TOK_ONE) ? 1u : 0u) << ROW_CNT_SHIFT), 0u, 0u, 0u);
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) { r.cnt[k] = row_count(r.row[k]); v += r.cnt[k]; }
    return v;
}
#define TKAMD_CP_SCATTER(DST, R, O)                                                                           \
    _Pragma("unroll") for (int k = 0; k < CP_ITEMS; ++k) {                                                    \
        const uint32_t c = (R).cnt[k];                                                                        \
        if (c) {                                                                                              \
            const bool more = ((R).row[k].x >> ROW_CNT_SHIFT) == ROW_CNT_MORE;                                \
            (DST)[(O)] = (R).row[k].x & ROW_ID_MASK;                                                          \
            if (!more) {                                                                                      \
                if (c > 1) (DST)[(O) + 1] = (R).row[k].y;                                                     \
                if (c > 2) (DST)[(O) + 2] = (R).row[k].z;                                                     \
                if (c > 3) (DST)[(O) + 3] = (R).row[k].w;                                                     \
            } else {                                                                                          \
                /* ids 1.. of a longer run: tmp_ids[s + 1 ..], four to a load (16 bytes at any alignment; the  */ \
                /* buffer extends four words past the text, so the last load may overshoot the run)           */ \
                const uint32_t* const src_ = tmp_ids + (R).row[k].y;                                          \
                for (uint32_t j = 1; j < c; j += 4) {                                                         \
                    const Unaligned16 v_ = *(const Unaligned16*)(src_ + j);                                   \
                    (DST)[(O) + j] = v_.a;                                                                    \
                    if (j + 1 < c) (DST)[(O) + j + 1] = v_.b;                                                 \
                    if (j + 2 < c) (DST)[(O) + j + 2] = v_.c;                                                 \
                    if (j + 3 < c) (DST)[(O) + j + 3] = v_.d;                                                 \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        (O) += c;                                                                                             \
    }

__global__ __launch_bounds__(CP_NT) void k_compact(const uint32_t* __restrict__ tok0, const uint4* __restrict__ rows, const uint4* __restrict__ crows,
                                                   const uint32_t* __restrict__ tmp_ids, const int64_t* __restrict__ n_pretok,
                                                   unsigned long long* __restrict__ state,
                                                   int64_t* __restrict__ n_tok, uint32_t* __restrict__ pt_tokoff, uint32_t* __restrict__ ids) {
    __shared__ uint32_t sm[4];
    __shared__ uint32_t s_stage[2][CP_STAGE];
    __shared__ uint32_t s_loc[2][CP_CHUNK];              // chunk-local token offset of every pre-token
    __shared__ uint32_t s_tot[2];
    __shared__ unsigned long long s_base;
    const int64_t P = *n_pretok;
    const int64_t n_chunks = (P + CP_CHUNK - 1) / CP_CHUNK;
    const int tid = (int)threadIdx.x;
    // front half of a chunk into LDS buffer b
    auto front = [&](int64_t ch, int b) {
        const int64_t p0 = ch * CP_CHUNK + (int64_t)tid * CP_ITEMS;
        CpRows r;
        const uint32_t v = cp_load(tok0, rows, crows, p0, P, r);
        uint32_t tot;
        const uint32_t ex = block256_excl_scan(v, sm, &tot);
        if (tid == 0) { lb_publish(state, ch, (unsigned long long)tot); s_tot[b] = tot; }
        uint32_t acc = ex;
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) { s_loc[b][tid * CP_ITEMS + k] = acc; acc += r.cnt[k]; }
        if (tot <= (uint32_t)CP_STAGE) {
            uint32_t o = ex;
            uint32_t* const dst = s_stage[b];
            TKAMD_CP_SCATTER(dst, r, o)
        }
    };
    int b = 0;
    if ((int64_t)blockIdx.x < n_chunks) front(blockIdx.x, 0);
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x, b ^= 1) {
        const int64_t nxt = ch + gridDim.x;
        if (nxt < n_chunks) front(nxt, b ^ 1);            // (its two barriers also order this chunk's LDS writes before the reads below)
        else __syncthreads();
        const uint32_t tot = s_tot[b];
        if (tid < 64) {                                    // wavefront 0 resolves the chunk's place in the token stream
            const unsigned long long base = lb_resolve(state, ch, (unsigned long long)tot);
            if (tid == 0) s_base = base;
        }
        __syncthreads();
        const unsigned long long base = s_base;
        if (ch == n_chunks - 1 && tid == 0) *n_tok = (int64_t)(base + tot);
        if (pt_tokoff) {
            const int64_t pc = ch * CP_CHUNK;
#pragma unroll
            for (int i = 0; i < CP_ITEMS; ++i) {
                const int q = i * CP_NT + tid;
                if (pc + q < P) pt_tokoff[pc + q] = (uint32_t)base + s_loc[b][q];
            }
        }
        if (tot <= (uint32_t)CP_STAGE) {
            for (uint32_t i = (uint32_t)tid; i < tot; i += CP_NT) ids[base + i] = s_stage[b][i];
        } else {                                           // rare: too many tokens for the buffer -- scatter from the rows
            const int64_t p0 = ch * CP_CHUNK + (int64_t)tid * CP_ITEMS;
            CpRows r;
            cp_load(tok0, rows, crows, p0, P, r);
            uint32_t o = s_loc[b][tid * CP_ITEMS];
            uint32_t* const dst = ids + base;
            TKAMD_CP_SCATTER(dst, r, o)
        }
        __syncthreads();                                   // buffer b is free for front() of the chunk after next
    }
}
#undef TKAMD_CP_SCATTER

__global__ void k_doc_tok_offsets(const uint32_t* __restrict__ doc_pt, int64_t n_docs, const uint32_t* __restrict__ pt_tokoff,
                                  const int64_t* __restrict__ n_pretok, const int64_t* __restrict__ n_tok,
                                  int64_t* __restrict__ tok_offsets) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    uint32_t p = doc_pt[d];
    tok_offsets[d] = ((int64_t)p < *n_pretok) ? (int64_t)pt_tokoff[p] : *n_tok;
}

// is_pretokenized inputs (InputSequence::PreTokenized, tokenizer/mod.rs:782-795): every word of a sequence went through the pipeline
// as a document of its own -- the reference encodes each word separately and merges the encodings -- and sequence s owns the tokens
// of the words [seq_off[s], seq_off[s + 1]).  (seq_off is the validated copy: launch_validate_csr.)
__global__ void k_seq_tok_offsets(const int64_t* __restrict__ seq_off, int64_t n_seqs, const int64_t* __restrict__ word_tok_off,
                                  int64_t* __restrict__ seq_tok_off) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= n_seqs) seq_tok_off[s] = word_tok_off[seq_off[s]];
}
// Encoding.word_ids of a pre-tokenized sequence: the index of the word in its sequence (do_tokenize's word_idx, mod.rs:1178-1200)
__global__ void k_word_index(const int64_t* __restrict__ seq_off, int64_t n_seqs, int64_t n_words, uint32_t* __restrict__ widx) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    int64_t lo = 0, hi = n_seqs;                          // last s with seq_off[s] <= w
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (seq_off[mid] <= w) lo = mid; else hi = mid; }
    widx[w] = (uint32_t)(w - seq_off[lo]);
}

// =================================================================================================
// K_token_meta: per-token (start, end) offsets and word ids.
// Replaces the per-token half of PreTokenizedString::into_encoding (tokenizer/pre_tokenizer.rs:231-256):
//   offsets = split.offsets_original().0 + convert_offsets(Normalized(token.offsets))   (:237-241)
//   word    = index of the split inside the document                                     (:252-256)
// plus BytesToCharOffsetConverter for OffsetType::Char (:329-364) and the ByteLevel post-processor's
// process_offsets (pre_tokenizers/byte_level.rs:202-234) when trim_offsets is set.
// Byte-level rule (byte_level.rs:135-143, tests/offsets.rs:47-57): a token that covers only part of a
// multi-byte char reports the whole char, so starts snap back and ends snap forward to char boundaries.
// One lane per pre-token; the document of a pre-token is found by binary search over doc_pt.
// =================================================================================================
__device__ __forceinline__ uint32_t lead_rank(const unsigned long long* __restrict__ leadmask, const uint32_t* __restrict__ lprefix, uint32_t pos) {
    unsigned long long m = leadmask[pos >> 6];
    uint32_t b = pos & 63u;
    return lprefix[pos >> 6] + (uint32_t)__popcll(m & ((1ull << b) - 1ull));
}

__global__ __launch_bounds__(256) void k_leadmask(const uint8_t* __restrict__ text, int64_t n_bytes, unsigned long long* __restrict__ leadmask) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool lead = (i < n_bytes) && ((text[i] & 0xC0u) != 0x80u);
    uint64_t m = __ballot(lead);
    if ((threadIdx.x & 63) == 0 && i <= n_bytes) leadmask[i >> 6] = m;
}

__global__ __launch_bounds__(256) void k_token_meta(MetaArgs a) {
    const int64_t P = *a.n_pretok;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
        const uint32_t o = a.pt_tokoff[p];
        const uint32_t c = ((p + 1 < P) ? a.pt_tokoff[p + 1] : (uint32_t)*a.n_tok) - o;
        if (!c) continue;
        const uint32_t s = a.pt_start[p], e = a.pt_end ? a.pt_end[p] : a.pt_start[p + 1];
        // document of this pre-token: last d with doc_pt[d] <= p
        int64_t lo = 0, hi = a.n_docs;
        while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if ((int64_t)a.doc_pt[mid] <= p) lo = mid; else hi = mid; }
        const int64_t d = lo;
        const uint32_t word = a.word_of_doc ? a.word_of_doc[d] : (uint32_t)(p - a.doc_pt[d]);
        const uint32_t xdoc = (uint32_t)a.x_doc_off[d];
        const uint32_t odoc = (uint32_t)a.doc_off[d];
        uint32_t rel = 0;
        // an added-token match is one token whose own length is on record (k_scatter_matches): a later, overlapping match may have
        // cut it short in the start mask, and its text is the raw slice (trimmed by real whitespace chars)
        const bool is_match = a.matchmask && a.tmp_end && ((a.matchmask[s >> 6] >> (s & 63)) & 1ull);
        for (uint32_t j = 0; j < c; ++j) {
            uint32_t rel_end = (c == 1) ? (e - s) : a.tmp_end[s + j];
            if (a.want_words) a.word_ids[o + j] = word;
            if (a.want_offsets) {
                uint32_t ts = s + rel, te = s + rel_end;              // token bytes in x space
                uint32_t bs = ts, be = te;
                if (a.byte_level && !is_match) {                      // snap to char boundaries inside the pre-token
                    while (bs > s && (a.x_text[bs] & 0xC0u) == 0x80u) --bs;
                    while (be < e && (a.x_text[be] & 0xC0u) == 0x80u) ++be;
                }
                // x space -> original text
                uint32_t os, oe;
                const uint8_t* ttext = a.x_text;                      // text the trimming below reads, and the token's span in it
                uint32_t tts = ts, tte = te;
                if (is_match) {
                    const uint32_t ml = a.tmp_end[s];
                    os = a.norig ? a.norig[s] : s - xdoc + odoc;
                    if (ml & MATCH_LEN_ORIG) { oe = os + (ml & ~MATCH_LEN_ORIG); ttext = a.text; tts = os; tte = oe; }
                    else { tte = s + ml; oe = a.norig ? a.norig_e[tte - 1] : tte - xdoc + odoc; }
                }
                else if (a.norig) { os = a.norig[bs]; oe = a.norig_e[be - 1]; }
                else { os = bs - xdoc + odoc; oe = be - xdoc + odoc; }
                if (a.char_mode) {
                    uint32_t base = lead_rank(a.leadmask, a.lprefix, odoc);
                    os = lead_rank(a.leadmask, a.lprefix, os) - base;
                    oe = lead_rank(a.leadmask, a.lprefix, oe) - base;
                } else { os -= odoc; oe -= odoc; }
                if (a.trim_offsets) {                                 // process_offsets, byte_level.rs:202-234
                    uint32_t lead_sp = 0, trail_sp = 0;
                    if (is_match) {
                        // an added token's text is the raw slice: its leading / trailing chars are tested with char::is_whitespace
                        uint32_t q = tts;
                        while (q < tte) { uint32_t l; if (!(uc_flags(utf8_global(ttext, q, &l), a.uc1, a.uc2) & UC_RUST_WS)) break; ++lead_sp; q += l; }
                        q = tte;
                        while (q > tts) {
                            uint32_t r = q - 1;
                            while (r > tts && (ttext[r] & 0xC0u) == 0x80u) --r;
                            uint32_t l;
                            if (!(uc_flags(utf8_global(ttext, r, &l), a.uc1, a.uc2) & UC_RUST_WS)) break;
                            ++trail_sp;
                            q = r;
                        }
                    } else {
                        while (ts + lead_sp < te && a.x_text[ts + lead_sp] == 0x20u) ++lead_sp;
                        while (trail_sp < te - ts && a.x_text[te - 1 - trail_sp] == 0x20u) ++trail_sp;
                    }
                    if (lead_sp) {
                        bool is_first = (word == 0 && j == 0) || os == 0;
                        if (is_first && a.pp_add_prefix_space && lead_sp == 1) lead_sp = 0;
                        os = min(os + lead_sp, oe);
                    }
                    if (trail_sp && oe >= trail_sp) oe = max(oe - trail_sp, os);
                }
                a.offsets[2 * (size_t)(o + j)] = os;
                a.offsets[2 * (size_t)(o + j) + 1] = oe;
            }
            rel = rel_end;
        }
    }
}

// =================================================================================================
// K_add_specials: PostProcessor::process for a single sequence with add_special_tokens = true
// (BertProcessing processors/bert.rs:51-120, RobertaProcessing, TemplateProcessing template.rs:544-590):
// every document becomes  prefix ids | its tokens | suffix ids ; specials carry offsets (0,0) and no word id.
// One wavefront per document copies the document's tokens to their shifted place.
// =================================================================================================
__global__ __launch_bounds__(256) void k_add_specials(SpecialArgs a) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    const int64_t add = (int64_t)a.n_prefix + a.n_suffix;
    for (int64_t d = wave; d <= a.n_docs; d += n_waves) {
        const int64_t lo = a.tok_offsets[d];
        const int64_t nlo = lo + d * add;
        if (lane == 0) a.tok_offsets2[d] = nlo;
        if (d == a.n_docs) { if (lane == 0) *a.n_tok2 = nlo; break; }
        const int64_t n = a.tok_offsets[d + 1] - lo;
        for (int64_t q = lane; q < a.n_prefix; q += 64) {
            a.ids2[nlo + q] = a.prefix[q];
            if (a.offsets) { a.offsets2[2 * (nlo + q)] = 0; a.offsets2[2 * (nlo + q) + 1] = 0; }
            if (a.word_ids) a.word_ids2[nlo + q] = 0xFFFFFFFFu;
        }
        const int64_t body = nlo + a.n_prefix;
        for (int64_t q = lane; q < n; q += 64) {
            a.ids2[body + q] = a.ids[lo + q];
            if (a.offsets) { a.offsets2[2 * (body + q)] = a.offsets[2 * (lo + q)]; a.offsets2[2 * (body + q) + 1] = a.offsets[2 * (lo + q) + 1]; }
            if (a.word_ids) a.word_ids2[body + q] = a.word_ids[lo + q];
        }
        for (int64_t q = lane; q < a.n_suffix; q += 64) {
            a.ids2[body + n + q] = a.suffix[q];
            if (a.offsets) { a.offsets2[2 * (body + n + q)] = 0; a.offsets2[2 * (body + n + q) + 1] = 0; }
            if (a.word_ids) a.word_ids2[body + n + q] = 0xFFFFFFFFu;
        }
    }
}

// =================================================================================================
// Truncation -> special tokens -> padding of the finished encodings, for a single sequence per document:
//   truncate_encodings (utils/truncation.rs:70-160) with n_added_tokens taken off max_length (tokenizer/mod.rs:1270-1284),
//   Encoding::truncate (tokenizer/encoding.rs:307-400; direction Right keeps the beginning, Left the end; the overflowing
//   pieces are not materialised), PostProcessor::process, pad_encodings (utils/padding.rs:50-85: BatchLongest / Fixed,
//   pad_to_multiple_of, direction; an encoding already longer than the target is left alone).
// Three small kernels over the documents: lengths (+ batch maximum), new CSR, copy (one wavefront per document).
// =================================================================================================
__global__ __launch_bounds__(256) void k_final_lens(FinalArgs a) {
    __shared__ uint32_t smax[4];
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t l = 0;
    if (d < a.n_docs) {
        const uint64_t n = (uint64_t)(a.tok_offsets[d + 1] - a.tok_offsets[d]);
        if (n > a.trunc_len && a.trunc_needs_pair) atomicOr(a.err, ERR_TRUNC_SECOND);
        if (n > a.trunc_len && a.trunc_len > 0u && a.trunc_stride >= a.trunc_len) atomicOr(a.err, ERR_TRUNC_STRIDE);      // encoding.rs:319
        l = (uint32_t)min(n, (uint64_t)a.trunc_len) + (uint32_t)(a.n_prefix + a.n_suffix);
        a.len1[d] = l;
    }
    if (a.pad_on && !a.pad_fixed) {                         // BatchLongest: one atomic per workgroup
        uint32_t m = l;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, s, 64));
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(a.target, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
    }
}
__device__ __forceinline__ uint32_t final_target(const FinalArgs& a) {
    uint32_t t = a.pad_fixed ? a.pad_length : *a.target;
    if (a.pad_multiple > 0 && t % a.pad_multiple > 0) t += a.pad_multiple - t % a.pad_multiple;
    return t;
}
__global__ __launch_bounds__(256) void k_final_fin(FinalArgs a) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d > a.n_docs) return;
    uint32_t f = 0;
    if (d < a.n_docs) { f = a.len1[d]; if (a.pad_on) f = max(f, final_target(a)); }
    a.fin[d] = f;
}
__global__ __launch_bounds__(256) void k_final_down(FinalArgs a) {
    __shared__ uint32_t sm[4];
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t x = (d <= a.n_docs) ? a.fin[d] : 0u;
    uint32_t tot;
    const uint32_t ex = a.bsum[blockIdx.x] + block256_excl_scan(x, sm, &tot);
    if (d <= a.n_docs) a.tok_offsets2[d] = (int64_t)ex;
}
// ---- overflowing encodings (TKAMD_WANT_OVERFLOW) ----
// Encoding::truncate keeps what it cuts off: further windows of max_len tokens, each sharing `stride` tokens with its neighbour
// (tokenizer/encoding.rs:307-395), pushed to Encoding.overflowing; the post-processor then puts the same special tokens around
// every one of them (processors/bert.rs:88-125, Encoding::merge_with encoding.rs:408-432) and Encoding::pad pads them like the
// encoding itself (:466-469).  Here they are simply further encodings of the result, numbered right behind their document's own:
// parts per document -> scan -> (document, first token, count) per encoding; from there on the epilogue runs per encoding.
__global__ __launch_bounds__(256) void k_ovf_parts(FinalArgs a) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d > a.n_docs) return;
    uint32_t p = 0;
    if (d < a.n_docs) {
        const uint64_t n = (uint64_t)(a.tok_offsets[d + 1] - a.tok_offsets[d]);
        if (n > a.trunc_len && a.trunc_needs_pair) atomicOr(a.err, ERR_TRUNC_SECOND);
        p = ovf_parts(n, a.trunc_len, a.trunc_stride);
        if (p == 0u) { atomicOr(a.err, ERR_TRUNC_STRIDE); p = 1u; }
        if (p == 0xFFFFFFFFu) { atomicOr(a.err, ERR_TOO_MANY_TOKENS); p = 1u; }
    }
    a.ovf_parts[d] = p;
}
__global__ __launch_bounds__(256) void k_ovf_ranges(FinalArgs a) {
    __shared__ uint32_t smax[4];
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t l = 0;                                         // length of the truncated encoding itself (part 0)
    if (d < a.n_docs) {
        const uint64_t n = (uint64_t)(a.tok_offsets[d + 1] - a.tok_offsets[d]);
        const int64_t e0 = a.enc_base[d];
        const uint32_t parts = (uint32_t)(a.enc_base[d + 1] - e0);
        const uint32_t add = (uint32_t)(a.n_prefix + a.n_suffix);
        for (uint32_t p = 0; p < parts; ++p) {
            uint64_t s, c;
            ovf_part_range(n, a.trunc_len, a.trunc_stride, a.trunc_left != 0u, p, &s, &c);
            a.enc_doc[e0 + p] = (uint32_t)d;
            a.enc_start[e0 + p] = (uint32_t)s;
            a.enc_cnt[e0 + p] = (uint32_t)c;
            a.len1[e0 + p] = (uint32_t)c + add;
            if (p == 0u) l = (uint32_t)c + add;
        }
    }
    if (a.pad_on && !a.pad_fixed) {                         // BatchLongest looks at the encodings themselves, not at their overflowing pieces (utils/padding.rs:55-63)
        uint32_t m = l;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, s, 64));
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(a.target, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
    }
}
__global__ __launch_bounds__(256) void k_finalize(FinalArgs a) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave; d < a.n_docs; d += n_waves) {
        int64_t n, src;
        if (a.enc_doc) {                                    // d numbers the encodings: a document's own, then its overflowing ones
            n = a.enc_cnt[d];
            src = a.tok_offsets[a.enc_doc[d]] + a.enc_start[d];
        } else {
            const int64_t lo = a.tok_offsets[d], n_all = a.tok_offsets[d + 1] - lo;
            n = min(n_all, (int64_t)a.trunc_len);
            src = lo + (a.trunc_left ? n_all - n : 0);
        }
        const int64_t dst0 = a.tok_offsets2[d], total = a.tok_offsets2[d + 1] - dst0;
        const int64_t real = n + a.n_prefix + a.n_suffix, pads = total - real;
        const int64_t body = dst0 + (a.pad_left ? pads : 0);
        if (lane == 0 && a.pad_count) a.pad_count[d] = (uint32_t)pads;
        for (int64_t q = lane; q < pads; q += 64) {
            const int64_t o = a.pad_left ? dst0 + q : body + real + q;
            a.ids2[o] = a.pad_id;
            if (a.offsets) { a.offsets2[2 * o] = 0; a.offsets2[2 * o + 1] = 0; }
            if (a.word_ids) a.word_ids2[o] = 0xFFFFFFFFu;
        }
        for (int64_t q = lane; q < a.n_prefix; q += 64) {
            a.ids2[body + q] = a.prefix[q];
            if (a.offsets) { a.offsets2[2 * (body + q)] = 0; a.offsets2[2 * (body + q) + 1] = 0; }
            if (a.word_ids) a.word_ids2[body + q] = 0xFFFFFFFFu;
        }
        const int64_t seq = body + a.n_prefix;
        for (int64_t q = lane; q < n; q += 64) {
            a.ids2[seq + q] = a.ids[src + q];
            if (a.offsets) { a.offsets2[2 * (seq + q)] = a.offsets[2 * (src + q)]; a.offsets2[2 * (seq + q) + 1] = a.offsets[2 * (src + q) + 1]; }
            if (a.word_ids) a.word_ids2[seq + q] = a.word_ids[src + q];
        }
        for (int64_t q = lane; q < a.n_suffix; q += 64) {
            a.ids2[seq + n + q] = a.suffix[q];
            if (a.offsets) { a.offsets2[2 * (seq + n + q)] = 0; a.offsets2[2 * (seq + n + q) + 1] = 0; }
            if (a.word_ids) a.word_ids2[seq + n + q] = 0xFFFFFFFFu;
        }
    }
}

// =================================================================================================
// The same epilogue for PAIRS (EncodeInput::Dual, tokenizer/mod.rs:871-889): documents 2i and 2i+1 went through the pipeline as
// sequence A and B of encoding i; here they are cut together (truncate_encodings, utils/truncation.rs:70-160: LongestFirst /
// OnlyFirst / OnlySecond), laid out by the post-processor's pair template with its type ids (processors/bert.rs:121-150,
// roberta.rs, template.rs:544-590; without one: A then B, type ids 0 / 1, PostProcessor::default_process), and padded.
// Every output token also gets its sequence id (0 / 1; 2 special; 3 padding) -- Encoding::token_to_sequence, the masks.
// =================================================================================================
__global__ __launch_bounds__(256) void k_pair_lens(PairArgs a) {
    __shared__ uint32_t smax[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t l = 0;
    if (i < a.n_pairs) {
        uint64_t n1 = (uint64_t)(a.tok_offsets[2 * i + 1] - a.tok_offsets[2 * i]), n2 = (uint64_t)(a.tok_offsets[2 * i + 2] - a.tok_offsets[2 * i + 1]);
        if (a.trunc_on) {
            // max_length - n_added_tokens (mod.rs:1273-1279; the subtraction wraps in the reference's release build when max_length is smaller)
            const uint64_t maxl = (a.n_special && a.trunc_max < a.n_special) ? ~0ull : (uint64_t)(a.trunc_max - a.n_special);
            const uint64_t total = n1 + n2;
            if (maxl == 0) { n1 = 0; n2 = 0; }
            else if (total > maxl) {
                const uint64_t to_remove = total - maxl;
                if (a.trunc_strategy == 0) {                          // LongestFirst (truncation.rs:101-141)
                    uint64_t s1 = n1, s2 = n2;
                    const bool swap = s1 > s2;
                    if (swap) { const uint64_t x = s1; s1 = s2; s2 = x; }
                    if (s1 > maxl) s2 = s1; else s2 = max(s1, maxl - s1);
                    if (s1 + s2 > maxl) { s1 = maxl / 2; s2 = s1 + maxl % 2; }
                    if (swap) { const uint64_t x = s1; s1 = s2; s2 = x; }
                    n1 = min(n1, s1); n2 = min(n2, s2);
                } else {                                              // OnlyFirst / OnlySecond (:143-159)
                    uint64_t& tgt = a.trunc_strategy == 1 ? n1 : n2;
                    if (tgt > to_remove) tgt -= to_remove;
                    else atomicOr(a.err, ERR_TRUNC_SHORT);
                }
            }
        }
        // Encoding::truncate(kept, stride): a sequence that is cut to kept > 0 tokens asserts stride < kept (encoding.rs:319)
        {
            const uint64_t a1 = (uint64_t)(a.tok_offsets[2 * i + 1] - a.tok_offsets[2 * i]), a2 = (uint64_t)(a.tok_offsets[2 * i + 2] - a.tok_offsets[2 * i + 1]);
            if ((n1 < a1 && n1 > 0 && a.trunc_stride >= n1) || (n2 < a2 && n2 > 0 && a.trunc_stride >= n2)) atomicOr(a.err, ERR_TRUNC_STRIDE);
        }
        a.keep[2 * i] = (uint32_t)n1;
        a.keep[2 * i + 1] = (uint32_t)n2;
        l = (uint32_t)(n1 + n2) + a.n_special;
        a.len1[i] = l;
    }
    if (a.pad_on && !a.pad_fixed) {
        uint32_t m = l;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, s, 64));
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(a.target, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
    }
}
__global__ __launch_bounds__(256) void k_pair_finalize(PairArgs a) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t i = wave; i < a.n_pairs; i += n_waves) {
        const int64_t dst0 = a.tok_offsets2[i], total = a.tok_offsets2[i + 1] - dst0;
        const int64_t real = a.len1[i], pads = total - real;
        int64_t cur = dst0 + (a.pad_left ? pads : 0);
        if (lane == 0 && a.pad_count) a.pad_count[i] = (uint32_t)pads;
        for (int64_t q = lane; q < pads; q += 64) {
            const int64_t o = a.pad_left ? dst0 + q : cur + real + q;
            a.ids2[o] = a.pad_id;
            a.type_ids2[o] = (uint8_t)a.pad_type_id;
            a.seq_ids2[o] = 3;
            if (a.offsets) { a.offsets2[2 * o] = 0; a.offsets2[2 * o + 1] = 0; }
            if (a.word_ids) a.word_ids2[o] = 0xFFFFFFFFu;
        }
        for (int k = 0; k < a.n_tpl; ++k) {
            const uint32_t kind = a.tpl[3 * k], id = a.tpl[3 * k + 1], ty = a.tpl[3 * k + 2];
            if (kind == 2u) {
                if (lane == 0) {
                    a.ids2[cur] = id;
                    a.type_ids2[cur] = (uint8_t)ty;
                    a.seq_ids2[cur] = 2;
                    if (a.offsets) { a.offsets2[2 * cur] = 0; a.offsets2[2 * cur + 1] = 0; }
                    if (a.word_ids) a.word_ids2[cur] = 0xFFFFFFFFu;
                }
                cur += 1;
            } else {
                const int64_t d = 2 * i + kind;
                const int64_t lo = a.tok_offsets[d], n_all = a.tok_offsets[d + 1] - lo, n = a.keep[d];
                const int64_t src = lo + (a.trunc_left ? n_all - n : 0);
                for (int64_t q = lane; q < n; q += 64) {
                    a.ids2[cur + q] = a.ids[src + q];
                    a.type_ids2[cur + q] = (uint8_t)ty;
                    a.seq_ids2[cur + q] = (uint8_t)kind;
                    if (a.offsets) { a.offsets2[2 * (cur + q)] = a.offsets[2 * (src + q)]; a.offsets2[2 * (cur + q) + 1] = a.offsets[2 * (src + q) + 1]; }
                    if (a.word_ids) a.word_ids2[cur + q] = a.word_ids[src + q];
                }
                cur += n;
            }
        }
    }
}
