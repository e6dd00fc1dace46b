// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers and kernels/scan_util.hip; it is not compiled on its own).  The epilogues over the finished token CSR -- special tokens,
// truncation (with its overflowing encodings), padding, pairs -- and their launchers.  This slice and scan_util.hip use nothing but
// plain HIP (thread / block indices, __shared__, __syncthreads, wavefront shuffles, atomics): tests/test_epilogue_core.py compiles
// them for the host under a small SIMT shim and runs them against the wheel's vectors without a GPU.

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
        // (max_length 0 cuts everything before the strategy is looked at, utils/truncation.rs:75-81)
        if (n > a.trunc_len && a.trunc_len > 0u && a.trunc_needs_pair) atomicOr(a.err, ERR_TRUNC_SECOND);
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
        if (n > a.trunc_len && a.trunc_len > 0u && a.trunc_needs_pair) atomicOr(a.err, ERR_TRUNC_SECOND);
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
        if (a.type_ids2) {
            // the template's type id is put on the sequence's own encoding only (set_type_ids on encodings[0], template.rs:554-559);
            // the special tokens are fresh encodings in every window
            const bool own = !a.enc_doc || d == a.enc_base[a.enc_doc[d]];
            for (int64_t q = lane; q < total; q += 64) {
                const int64_t r = a.pad_left ? q - pads : q;       // position inside prefix + sequence + suffix
                uint32_t ty = a.pad_type_id, sq = 3u;
                if (r >= 0 && r < real) {
                    sq = 2u;
                    if (r < a.n_prefix) ty = a.prefix_ty[r];
                    else if (r < a.n_prefix + n) { ty = own ? a.seq_ty : 0u; sq = 0u; }
                    else ty = a.suffix_ty[r - a.n_prefix - n];
                }
                a.type_ids2[dst0 + q] = (uint8_t)ty;
                a.seq_ids2[dst0 + q] = (uint8_t)sq;
            }
        }
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
            // (process_offsets runs on the cut encoding: its token 0 keeps the one leading space that stands for the prefix space)
            if (a.offsets) {
                const uint32_t back = (q == 0 && a.trim1) ? a.trim1[src] : 0u;
                a.offsets2[2 * (seq + q)] = a.offsets[2 * (src + q)] - (back ? 1u : 0u);
                a.offsets2[2 * (seq + q) + 1] = a.offsets[2 * (src + q) + 1] - (back == 2u ? 1u : 0u);
            }
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
// A mixed batch (PairArgs::inp_off): input i is one sequence (EncodeInput::Single) or two (::Dual).  What the three kernels below need
// of an input: its first sequence, whether it has a second one, its token counts -- and, per kind, the template and its special tokens.
struct PairInput { int64_t d0; bool has_b; uint64_t n1, n2; };
__device__ __forceinline__ PairInput pair_input(const PairArgs& a, int64_t i) {
    PairInput p;
    p.d0 = 2 * i;
    p.has_b = true;
    int64_t c = 2;
    if (a.inp_off) {
        p.d0 = a.inp_off[i];
        c = a.inp_off[i + 1] - p.d0;
        if (c < 1 || c > 2) atomicOr(a.err, ERR_INPUT_KIND);      // (the batch fails; the sequences read below all exist: the CSR was validated)
        p.has_b = c >= 2;
    }
    p.n1 = c >= 1 ? (uint64_t)(a.tok_offsets[p.d0 + 1] - a.tok_offsets[p.d0]) : 0ull;
    p.n2 = p.has_b ? (uint64_t)(a.tok_offsets[p.d0 + 2] - a.tok_offsets[p.d0 + 1]) : 0ull;
    return p;
}
__global__ __launch_bounds__(256) void k_pair_lens(PairArgs a) {
    __shared__ uint32_t smax[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t l = 0;
    if (i < a.n_pairs) {
        const PairInput in = pair_input(a, i);
        const uint32_t n_special = in.has_b ? a.n_special : a.n_special1;      // get_n_added_tokens(is_pair), mod.rs:1270-1272
        uint64_t n1 = in.n1, n2 = in.n2;
        if (a.trunc_on) {
            // max_length - n_added_tokens (mod.rs:1273-1279; the subtraction wraps in the reference's release build when max_length is smaller)
            const uint64_t maxl = (n_special && a.trunc_max < n_special) ? ~0ull : (uint64_t)(a.trunc_max - n_special);
            const uint64_t total = n1 + n2;
            if (maxl == 0) { n1 = 0; n2 = 0; }
            else if (total > maxl && !in.has_b && a.trunc_strategy == 2) {
                atomicOr(a.err, ERR_TRUNC_SECOND);                    // OnlySecond without a second sequence (truncation.rs:147-151)
            } else if (total > maxl) {
                const uint64_t to_remove = total - maxl;
                if (a.trunc_strategy == 0) {                          // LongestFirst (truncation.rs:101-141; without a pair: total - to_remove, the same numbers)
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
            const uint64_t a1 = in.n1, a2 = in.n2;
            if ((n1 < a1 && n1 > 0 && a.trunc_stride >= n1) || (n2 < a2 && n2 > 0 && a.trunc_stride >= n2)) atomicOr(a.err, ERR_TRUNC_STRIDE);
        }
        a.keep[2 * i] = (uint32_t)n1;
        a.keep[2 * i + 1] = (uint32_t)n2;
        l = (uint32_t)(n1 + n2) + n_special;
        if (a.ovf_parts) {
            // Encoding::truncate keeps what it cuts off either sequence; the pair then leaves every combination of their windows
            const uint64_t a1 = in.n1, a2 = in.n2;
            uint64_t pa = ovf_parts(a1, (uint32_t)n1, a.trunc_stride), pb = ovf_parts(a2, (uint32_t)n2, a.trunc_stride);
            if (pa == 0u) pa = 1u;                          // (the stride assert: reported above)
            if (pb == 0u) pb = 1u;
            uint64_t p = pa * pb;
            if (p >= 0x7FFFFFFFull) { atomicOr(a.err, ERR_TOO_MANY_TOKENS); p = 1u; }
            a.ovf_parts[i] = (uint32_t)p;
        } else {
            a.len1[i] = l;
        }
    } else if (i == a.n_pairs && a.ovf_parts) {
        a.ovf_parts[i] = 0u;
    }
    if (a.ovf_parts) return;                                // lengths and the batch maximum: k_pair_ranges, per encoding
    if (a.pad_on && !a.pad_fixed) {
        uint32_t m = l;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, s, 64));
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(a.target, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
    }
}
// Encodings of pair i in the reference's order (Encoding::merge_with, tokenizer/encoding.rs:408-432, F = the sequence the template names
// first, S the other, f_1.. / s_1.. their overflowing windows):  F+S ;  for every f_x:  f_x+S , f_x+s_1 , f_x+s_2 .. ;  then F+s_1 , F+s_2 ..
// (the reference also hangs f_x+s_* below f_x+S, and f_*+s_y below F+s_y, as nested `overflowing` lists: the same encodings again --
// the host rebuilds those lists from the window indices written here).
__global__ __launch_bounds__(256) void k_pair_ranges(PairArgs a) {
    __shared__ uint32_t smax[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t l = 0;
    if (i < a.n_pairs) {
        const PairInput in = pair_input(a, i);
        const uint32_t n_special = in.has_b ? a.n_special : a.n_special1;
        const uint64_t n_all[2] = {in.n1, in.n2};
        const uint32_t keep[2] = {a.keep[2 * i], a.keep[2 * i + 1]};
        uint32_t parts[2] = {ovf_parts(n_all[0], keep[0], a.trunc_stride), ovf_parts(n_all[1], keep[1], a.trunc_stride)};
        if (parts[0] == 0u) parts[0] = 1u;
        if (parts[1] == 0u) parts[1] = 1u;
        const int64_t e0 = a.enc_base[i];
        const uint32_t total = (uint32_t)(a.enc_base[i + 1] - e0);
        if (total != parts[0] * parts[1]) { parts[0] = 1u; parts[1] = 1u; }      // (the count was clamped: an error is pending)
        const int F = (a.first_is_b && in.has_b) ? 1 : 0, S = 1 - F;      // (a single sequence: its own windows, in order -- the S side has one part)
        for (uint32_t q = 0; q < total; ++q) {
            // q -> (window of F, window of S) in the order above
            uint32_t wf, ws;
            const uint32_t rows = (parts[F] - 1u) * parts[S];          // the f_x rows
            if (q == 0u) { wf = 0u; ws = 0u; }
            else if (q <= rows) { wf = 1u + (q - 1u) / parts[S]; ws = (q - 1u) % parts[S]; }
            else { wf = 0u; ws = q - rows; }
            uint32_t w[2];
            w[F] = wf;
            w[S] = ws;
            uint64_t s0, c0, s1, c1;
            ovf_part_range(n_all[0], keep[0], a.trunc_stride, a.trunc_left != 0u, w[0], &s0, &c0);
            ovf_part_range(n_all[1], keep[1], a.trunc_stride, a.trunc_left != 0u, w[1], &s1, &c1);
            const int64_t e = e0 + q;
            a.enc_doc[e] = (uint32_t)i;
            a.enc_idx[2 * e] = w[0];
            a.enc_idx[2 * e + 1] = w[1];
            a.enc_win[4 * e] = (uint32_t)s0;
            a.enc_win[4 * e + 1] = (uint32_t)c0;
            a.enc_win[4 * e + 2] = (uint32_t)s1;
            a.enc_win[4 * e + 3] = (uint32_t)c1;
            a.len1[e] = (uint32_t)(c0 + c1) + n_special;
            if (q == 0u) l = (uint32_t)(c0 + c1) + n_special;
        }
    }
    if (a.pad_on && !a.pad_fixed) {                         // BatchLongest: the pairs' own encodings (utils/padding.rs:55-63)
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
        // (a mixed batch: the input behind encoding i, its first sequence, and the template of its kind)
        const int64_t inp = a.enc_doc ? (int64_t)a.enc_doc[i] : i;
        int64_t d0 = 2 * inp;
        const uint32_t* tpl = a.tpl;
        int n_tpl = a.n_tpl;
        if (a.inp_off) {
            d0 = a.inp_off[inp];
            if (a.inp_off[inp + 1] - d0 < 2) { tpl = a.tpl1; n_tpl = a.n_tpl1; }
        }
        for (int k = 0; k < n_tpl; ++k) {
            const uint32_t kind = tpl[3 * k], id = tpl[3 * k + 1], ty = tpl[3 * k + 2];
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
                int64_t n, src;
                uint32_t ty_here = ty;
                if (a.enc_doc) {                            // i numbers the encodings: this one's window of sequence `kind`
                    n = a.enc_win[4 * i + 2 * kind + 1];
                    src = a.tok_offsets[d0 + kind] + a.enc_win[4 * i + 2 * kind];
                    // the template's type id is put on the sequence's own encoding only (template.rs:554-559); an overflowing window
                    // keeps what encode gave its tokens: 0 for the first sequence, 1 for the second (mod.rs:879-884) -- RobertaProcessing
                    // writes zeros over its overflowing windows as well when it adds the special tokens (roberta.rs:121-126, 187-192)
                    if (a.enc_idx[2 * i + kind] != 0u && !a.ovf_ty_tpl) ty_here = kind;
                } else {
                    const int64_t d = d0 + kind;
                    const int64_t lo = a.tok_offsets[d], n_all = a.tok_offsets[d + 1] - lo;
                    n = a.keep[2 * i + kind];
                    src = lo + (a.trunc_left ? n_all - n : 0);
                }
                for (int64_t q = lane; q < n; q += 64) {
                    a.ids2[cur + q] = a.ids[src + q];
                    a.type_ids2[cur + q] = (uint8_t)ty_here;
                    a.seq_ids2[cur + q] = (uint8_t)kind;
                    if (a.offsets) {
                        const uint32_t back = (q == 0 && a.trim1) ? a.trim1[src] : 0u;
                        a.offsets2[2 * (cur + q)] = a.offsets[2 * (src + q)] - (back ? 1u : 0u);
                        a.offsets2[2 * (cur + q) + 1] = a.offsets[2 * (src + q) + 1] - (back == 2u ? 1u : 0u);
                    }
                    if (a.word_ids) a.word_ids2[cur + q] = a.word_ids[src + q];
                }
                cur += n;
            }
        }
    }
}

// ---- launchers ----
void launch_add_specials(hipStream_t st, int grid, const SpecialArgs& a) {
    hipLaunchKernelGGL(k_add_specials, dim3(grid), dim3(256), 0, st, a);
}
void launch_final_lens(hipStream_t st, const FinalArgs& a) {
    hipLaunchKernelGGL(k_final_lens, dim3(blocks_for(a.n_docs + 1, 256)), dim3(256), 0, st, a);
}
void launch_overflow_count(hipStream_t st, const FinalArgs& a, int64_t* n_enc) {
    const unsigned nb = blocks_for(a.n_docs + 1, 256);
    hipLaunchKernelGGL(k_ovf_parts, dim3(nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)a.ovf_parts, a.n_docs + 1, a.bsum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, a.bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, n_enc);
    FinalArgs s = a;                                        // exclusive scan of the parts = k_final_down over them
    s.fin = a.ovf_parts;
    s.tok_offsets2 = a.enc_base;
    hipLaunchKernelGGL(k_final_down, dim3(nb), dim3(256), 0, st, s);
}
void launch_overflow_ranges(hipStream_t st, const FinalArgs& a) {
    hipLaunchKernelGGL(k_ovf_ranges, dim3(blocks_for(a.n_docs + 1, 256)), dim3(256), 0, st, a);
}
void launch_final_offsets(hipStream_t st, const FinalArgs& a) {
    const unsigned nb = blocks_for(a.n_docs + 1, 256);
    hipLaunchKernelGGL(k_final_fin, dim3(nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)a.fin, a.n_docs + 1, a.bsum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, a.bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, a.n_tok2);
    hipLaunchKernelGGL(k_final_down, dim3(nb), dim3(256), 0, st, a);
}
void launch_finalize(hipStream_t st, int grid, const FinalArgs& a) {
    hipLaunchKernelGGL(k_finalize, dim3(grid), dim3(256), 0, st, a);
}
void launch_pair_lens(hipStream_t st, const PairArgs& a) {
    hipLaunchKernelGGL(k_pair_lens, dim3(blocks_for(a.n_pairs + 1, 256)), dim3(256), 0, st, a);
}
void launch_pair_overflow_scan(hipStream_t st, const PairArgs& a, int64_t* n_enc) {
    const unsigned nb = blocks_for(a.n_pairs + 1, 256);
    hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)a.ovf_parts, a.n_pairs + 1, a.bsum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, a.bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, n_enc);
    FinalArgs s{};                                          // exclusive scan of the counts = k_final_down over them
    s.n_docs = a.n_pairs;
    s.fin = a.ovf_parts;
    s.bsum = a.bsum;
    s.tok_offsets2 = a.enc_base;
    hipLaunchKernelGGL(k_final_down, dim3(nb), dim3(256), 0, st, s);
}
void launch_pair_ranges(hipStream_t st, const PairArgs& a) {
    hipLaunchKernelGGL(k_pair_ranges, dim3(blocks_for(a.n_pairs + 1, 256)), dim3(256), 0, st, a);
}
void launch_pair_finalize(hipStream_t st, int grid, const PairArgs& a) {
    hipLaunchKernelGGL(k_pair_finalize, dim3(grid), dim3(256), 0, st, a);
}
