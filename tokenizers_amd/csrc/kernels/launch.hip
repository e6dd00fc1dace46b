// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Host launchers.

// =================================================================================================
// host-side launchers (called from capi.cpp; plain C++ signatures, stream-ordered, no syncs)
// =================================================================================================
void launch_mark_doc_starts_n(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes, const int64_t* len_dev,
                              unsigned long long* docmask, int* err) {
    hipLaunchKernelGGL(k_mark_doc_starts, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, n_bytes, len_dev, docmask, err);
}
void launch_mark_doc_starts(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes,
                            unsigned long long* docmask, int* err) {
    launch_mark_doc_starts_n(st, doc_off, n_docs, n_bytes, nullptr, docmask, err);
}
void launch_validate_csr(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes, int* err, int64_t* san) {
    hipLaunchKernelGGL(k_mark_doc_starts, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, n_bytes, (const int64_t*)nullptr,
                       (unsigned long long*)nullptr, err);
    hipLaunchKernelGGL(k_sanitize_csr, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, n_bytes, (const int*)err, san);
}
void launch_pretok_gpt2(hipStream_t st, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                        const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* leadmask) {
    if (leadmask) hipLaunchKernelGGL((k_pretok_gpt2_seq<SQ_LUT_COPIES, true>), dim3(blocks_for(n_bytes + 1, 256 * SQ_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, leadmask);
    else hipLaunchKernelGGL((k_pretok_gpt2_seq<SQ_LUT_COPIES, false>), dim3(blocks_for(n_bytes + 1, 256 * SQ_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, (unsigned long long*)nullptr);
}
void launch_mask_scan(hipStream_t st, const unsigned long long* mask, int64_t n_words, uint32_t* bsum, uint32_t* wprefix,
                      int64_t* total, const int64_t* len_dev, uint32_t* tile_w) {
    unsigned nb = blocks_for(n_words, 256 * WS_PER);
    hipLaunchKernelGGL(k_words_reduce, dim3(nb), dim3(256), 0, st, mask, n_words, bsum, len_dev);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, total);
    hipLaunchKernelGGL(k_words_down, dim3(nb), dim3(256), 0, st, mask, n_words, (const uint32_t*)bsum, wprefix, len_dev, tile_w);
}
void launch_emit_pretok(hipStream_t st, const unsigned long long* startmask, const uint32_t* wprefix, int64_t n_bytes,
                        const int64_t* len_dev, const int64_t* n_pretok, uint32_t* pt_start) {
    // one wavefront per 64 mask words = 4096 bytes of text; 4 wavefronts per workgroup
    hipLaunchKernelGGL(k_emit_pretok, dim3(blocks_for(n_bytes + 1, 4 * 4096)), dim3(256), 0, st, startmask, wprefix, n_bytes, len_dev, n_pretok, pt_start);
}
void launch_doc_first_pretok(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes,
                             const unsigned long long* startmask, const uint32_t* wprefix, const int64_t* n_pretok, uint32_t* doc_pt, uint32_t* chunk_lo,
                             const int* err, int64_t* san) {
    hipLaunchKernelGGL(k_doc_first_pretok, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, n_bytes, startmask, wprefix, n_pretok, doc_pt, chunk_lo,
                       (uint32_t)COMPACT_CHUNK, err, san);
}

void launch_lookup(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev,
                   const unsigned long long* startmask, const unsigned long long* endmask, const uint32_t* wprefix, uint32_t* tok0,
                   const QueuePlan& plan, int* err, const unsigned long long* matchmask, const void* hot, const WordCache& wc,
                   uint32_t no_hits, uint32_t miss_is_unk, void* phases, uint32_t* counters) {
    LookupArgs a{};
    a.counters = counters;
    a.shortw = (const uint4*)t.shortw;
    a.shortw_mask = t.shortw_mask;
    a.shortw_bmask = t.shortw_bmask;
    a.shortw_disp = t.shortw_disp;
    a.shortw_k3 = t.shortw_k3;
    a.word_seed = t.word_seed;
    a.any_hit_final = t.ignore_merges;
    a.unk_id = t.unk_id;
    a.has_unk = t.has_unk;
    a.text = text;
    a.n_bytes_host = n_bytes;
    a.len_dev = len_dev;
    a.startmask = startmask;
    a.endmask = endmask;
    a.wprefix = wprefix;
    a.tok0 = tok0;
    for (int c = 0; c < 4; ++c) a.v[c] = plan.v[c];
    a.err = err;
    a.matchmask = matchmask;
    a.hot = (const uint4*)hot;
    a.cache_keys = wc.keys;
    a.claims = wc.claims;
    a.claim_mask = wc.claim_mask;
    a.no_hits = no_hits;
    a.miss_is_unk = miss_is_unk;
    a.phases = (unsigned long long*)phases;
    const int lds = lookup_lds_bytes();
    if (phases) { if (endmask) hipLaunchKernelGGL((k_lookup<true, true>), dim3(grid), dim3(LU_NT), lds, st, a); else hipLaunchKernelGGL((k_lookup<false, true>), dim3(grid), dim3(LU_NT), lds, st, a); }     // (the diagnostic instantiations: tkamd_debug_phases)
    else if (endmask) hipLaunchKernelGGL((k_lookup<true, false>), dim3(grid), dim3(LU_NT), lds, st, a);
    else hipLaunchKernelGGL((k_lookup<false, false>), dim3(grid), dim3(LU_NT), lds, st, a);
}
void launch_bpe_merge(hipStream_t st, int grid, int group, const DevTables& t, const uint8_t* text, const QView& v, void* rows,
                      uint32_t* tmp_ids, uint32_t* tmp_end, const QView* also) {
    uint4* r = (uint4*)rows;
    const QView none{};
    // LDS-resident keys (needs newid_affine; prepare_long_kernel() raised the LDS limit)
    // (the instantiation with the displacements in LDS only when they fit: the kernel must know at compile time where they are -- bpe.hip)
    const bool disp_fits = t.merge_bmask < (uint32_t)DISP_LDS_MAX;
    if (group == 5 && disp_fits) launch_lds_merge<16, 640, true, true>(st, grid * 2, t, text, v, none, r, tmp_ids, tmp_end);   // two 640-lane workgroups per CU (704 with 512 sub-queues: the prefix array is static LDS)
    else if (group == 5) launch_lds_merge<16, 640, false, true>(st, grid * 2, t, text, v, none, r, tmp_ids, tmp_end);
    else if (group == 6 && disp_fits) launch_lds_merge<32, 768, true, true>(st, grid, t, text, v, also ? *also : none, r, tmp_ids, tmp_end);
    else if (group == 6) launch_lds_merge<32, 768, false, true>(st, grid, t, text, v, also ? *also : none, r, tmp_ids, tmp_end);
    else if (group == 1)
        hipLaunchKernelGGL(k_bpe_merge_lane<16>, dim3(grid), dim3(256), 0, st, t, text, v, r, tmp_ids, tmp_end);
    else if (group == 2)
        hipLaunchKernelGGL(k_bpe_merge_lane<32>, dim3(grid), dim3(256), 0, st, t, text, v, r, tmp_ids, tmp_end);
    else
        hipLaunchKernelGGL(k_bpe_merge<64>, dim3(grid), dim3(256), 0, st, t, text, v, r, tmp_ids, tmp_end);
}
template <int KIND>
static void launch_pretok_local_t(hipStream_t st, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                                  const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* endmask, bool len_bound) {
    hipLaunchKernelGGL(k_pretok_local_lane<KIND>, dim3(blocks_for(n_bytes + 2, 256 * PLW_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask,
                       (len_bound && len_dev) ? 1 : 0);
}
void launch_pretok_local(hipStream_t st, int kind, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                         const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* endmask, bool len_bound) {
    if (kind == PT_WHITESPACE) launch_pretok_local_t<PT_WHITESPACE>(st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask, len_bound);
    else if (kind == PT_WHITESPACE_SPLIT) launch_pretok_local_t<PT_WHITESPACE_SPLIT>(st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask, len_bound);
    else launch_pretok_local_t<PT_BERT>(st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask, len_bound);
}
void launch_emit_pretok_end(hipStream_t st, const unsigned long long* startmask, const unsigned long long* endmask,
                            const uint32_t* wprefix, int64_t n_bytes, const int64_t* len_dev, uint32_t* pt_end) {
    hipLaunchKernelGGL(k_emit_pretok_end, dim3(blocks_for(n_bytes + 64, 4 * 4096)), dim3(256), 0, st, startmask, endmask, wprefix, n_bytes, len_dev, pt_end);
}
void launch_bert_normalize(hipStream_t st, const BnTables& bt, const uint8_t* text, int64_t n_bytes, const int64_t* doc_off, int64_t n_docs,
                           const unsigned long long* verbatim, uint8_t* olen, uint32_t* wsum, uint32_t* bsum, uint32_t* wbase, int64_t* x_len, uint8_t* ntext,
                           uint32_t* nos, uint32_t* noe, int64_t* ndoc_off, int* err) {
    const int64_t n_words = (n_bytes >> 6) + 1;
    uint8_t* const ltot = bn_ltot_of(olen, n_bytes);           // (one byte per 16-byte lane, behind the per-byte array: BnOlen, bert_norm_core.hpp)
    const BnOlen ol{olen, ltot};
    hipLaunchKernelGGL(k_bn_count, dim3(blocks_for(n_bytes + 1, 256 * BN_LANE)), dim3(256), 0, st, bt, text, n_bytes, verbatim, olen, ltot, wsum, err);

    unsigned nb = blocks_for(n_words, 256);
    hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)wsum, n_words, bsum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, x_len);
    hipLaunchKernelGGL(k_u32_down, dim3(nb), dim3(256), 0, st, (const uint32_t*)wsum, n_words, (const uint32_t*)bsum, wbase);
    hipLaunchKernelGGL(k_bn_write, dim3(blocks_for(n_bytes, 256 * BN_LANE)), dim3(256), 0, st, bt, text, n_bytes, verbatim, ol, (const uint32_t*)wbase, ntext, nos, noe);
    hipLaunchKernelGGL(k_bn_reorder_fix, dim3(std::min<unsigned>(blocks_for(n_bytes + 1, 256), 2048u)), dim3(256), 0, st, bt, text, n_bytes, verbatim, doc_off, n_docs,
                       ol, (const uint32_t*)wbase, ntext, nos, noe, err);
    hipLaunchKernelGGL(k_bn_doc_offsets, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, n_bytes, ol,
                       (const uint32_t*)wbase, (const int64_t*)x_len, ndoc_off);
}
void launch_long_vocab3(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v1, const QView& v2, const QView& v3, void* rows, uint32_t miss_is_unk,
                        int* err, const WordCache& wc) {
    hipLaunchKernelGGL(k_long_vocab3, dim3(3 * grid), dim3(256), 0, st, t, text, v1, v2, v3, (uint4*)rows, miss_is_unk, err, wc.claim_mask,
                       wc.claims ? (uint4*)wc.rows : (uint4*)nullptr, wc.claim_pos);
}
void launch_wordpiece_all(hipStream_t st, int grid_short, int grid_long, const DevTables& t, const uint8_t* text, const QueuePlan& plan, void* rows, uint32_t* tmp_ids,
                          uint32_t* tmp_end, int* err) {
    hipLaunchKernelGGL(k_wordpiece_all, dim3(3 * grid_long + grid_short), dim3(256), 0, st, t, text, plan.v[0], plan.v[1], plan.v[2], plan.v[3], (uint4*)rows, tmp_ids, tmp_end, err,
                       (uint32_t)(3 * grid_long));
}
void launch_pretok_llama3(hipStream_t st, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                          const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* slowmask,
                          const int64_t* doc_off, int64_t n_docs, const int64_t* n_docs_dev, uint32_t* slow_docs, uint32_t* n_slow_docs,
                          SplitRule rule, const uint16_t* ucc1, const uint8_t* ucc2, unsigned long long* tileflags, unsigned long long* leadmask) {
    // Three tiers: the per-lane bit-parallel kernel, the tile kernel on the tiles where that left bytes undecided, the sequential matcher
    // on the sentences (doc_off: documents, or the pieces between added-token matches) the tile kernel could not finish.  A member of the
    // family with the case-split letters (o200k, tekken: split_rule_fast_cs) runs the per-lane kernel with l3_window_starts_cs and then the
    // sequential matcher on the sentences that left a byte of undecided; what neither serves goes to the sequential matcher whole: every
    // sentence, one lane each.
    const L3Seq q{uc1, uc2, ucc1, ucc2, rule};
    const unsigned doc_blocks = std::min<unsigned>(blocks_for(n_docs, 256), 4096u);
    if (split_rule_fast(rule)) {
        // (tileflags: a bit per PT_TILE bytes, zeroed by the caller -- the lane kernel's note of the tiles it left something undecided in)
        if (leadmask) hipLaunchKernelGGL((k_pretok_llama3_lane<false, true>), dim3(blocks_for(n_bytes + 1, 256 * L3W_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, slowmask, rule,
                           (const uint16_t*)nullptr, (const uint8_t*)nullptr, tileflags, leadmask);
        else hipLaunchKernelGGL((k_pretok_llama3_lane<false, false>), dim3(blocks_for(n_bytes + 1, 256 * L3W_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, slowmask, rule,
                           (const uint16_t*)nullptr, (const uint8_t*)nullptr, tileflags, (unsigned long long*)nullptr);
        const unsigned n_tiles = blocks_for(n_bytes + 1, PT_TILE);      // (without a flag array: every tile, one a workgroup)
        hipLaunchKernelGGL(k_pretok_llama3, dim3(tileflags ? blocks_for(n_tiles, 64) : n_tiles), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, slowmask,
                           (const unsigned long long*)tileflags, rule);
        hipLaunchKernelGGL(k_l3_slow_docs, dim3(doc_blocks), dim3(256), 0, st, (const unsigned long long*)slowmask, doc_off, n_docs, n_docs_dev, slow_docs, n_slow_docs);
        hipLaunchKernelGGL(k_pretok_llama3_slow, dim3(1024), dim3(64), 0, st, text, doc_off, (const uint32_t*)slow_docs, (const uint32_t*)n_slow_docs, q, startmask);
    } else if (split_rule_fast_cs(rule) && ucc1) {
        if (leadmask) hipLaunchKernelGGL((k_pretok_llama3_lane<true, true>), dim3(blocks_for(n_bytes + 1, 256 * L3W_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, slowmask, rule, ucc1, ucc2,
                           (unsigned long long*)nullptr, leadmask);
        else hipLaunchKernelGGL((k_pretok_llama3_lane<true, false>), dim3(blocks_for(n_bytes + 1, 256 * L3W_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, slowmask, rule, ucc1, ucc2,
                           (unsigned long long*)nullptr, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(k_l3_slow_docs, dim3(doc_blocks), dim3(256), 0, st, (const unsigned long long*)slowmask, doc_off, n_docs, n_docs_dev, slow_docs, n_slow_docs);
        hipLaunchKernelGGL(k_pretok_llama3_slow, dim3(1024), dim3(64), 0, st, text, doc_off, (const uint32_t*)slow_docs, (const uint32_t*)n_slow_docs, q, startmask);
    } else {
        (void)hipMemsetAsync(startmask, 0, (size_t)((n_bytes >> 6) + 1) * 8, st);
        hipLaunchKernelGGL(k_l3_slow_docs, dim3(doc_blocks), dim3(256), 0, st, (const unsigned long long*)nullptr, doc_off, n_docs, n_docs_dev, slow_docs, n_slow_docs);
        hipLaunchKernelGGL(k_pretok_llama3_slow, dim3(8192), dim3(64), 0, st, text, doc_off, (const uint32_t*)slow_docs, (const uint32_t*)n_slow_docs, q, startmask);
    }
}
void launch_leadmask(hipStream_t st, const uint8_t* text, int64_t n_bytes, unsigned long long* leadmask) {
    hipLaunchKernelGGL(k_leadmask, dim3(blocks_for(n_bytes + 64, 256 * 16)), dim3(256), 0, st, text, n_bytes, leadmask);
}
void launch_seq_regroup(hipStream_t st, const int64_t* seq_off, int64_t n_seqs, int64_t n_words, const int64_t* word_tok_off, int64_t* seq_tok_off, uint32_t* widx,
                        int64_t* first_tok) {
    hipLaunchKernelGGL(k_seq_tok_offsets, dim3(blocks_for(n_seqs + 1, 256)), dim3(256), 0, st, seq_off, n_seqs, word_tok_off, seq_tok_off);
    if ((widx || first_tok) && n_words)
        hipLaunchKernelGGL(k_word_index, dim3(blocks_for(n_words, 256)), dim3(256), 0, st, seq_off, n_seqs, n_words, widx, (const int64_t*)seq_tok_off, first_tok);
}
// one instantiation of k_token_meta on a grid of what is RESIDENT at once -- a workgroup walks its tiles in a loop whose every step is a
// chain of dependent phases, so the kernel lasts as long as the workgroup with the most tiles: 2,048 workgroups on a chip that holds
// 1,536 of them ran a second, half-empty round (profiles/r6j_c2_sq_summary_byte.json).  The occupancy is the instantiation's own (five
// workgroups a CU for some, four for the others: kernels/output.hip)
template <bool E, bool S, bool C, bool M, bool N>
static void launch_tm(hipStream_t st, int grid, const MetaArgs& a) {
    static const int per_cu = [] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_token_meta<E, S, C, M, N>, 256, 0) != hipSuccess || n < 1) n = 4;
        return std::min(n, 8);
    }();
    hipLaunchKernelGGL((k_token_meta<E, S, C, M, N>), dim3(std::max(1, grid / 8) * per_cu), dim3(256), 0, st, a);
}
void launch_token_meta(hipStream_t st, int grid, const MetaArgs& a) {
    // (char_id: BPE over characters without an unk_token -- token edges depend on the tokens in front of them: the sequential shape)
    if (a.char_id) { hipLaunchKernelGGL(k_token_meta_seq, dim3(grid), dim3(256), 0, st, a); return; }
    // (SIMPLE also reads char offsets off the ORIGINAL text's lead-byte mask at x positions: the two texts must be one)
    const bool simple = !a.norig && !a.matchmask && !a.trim_offsets && !a.word_of_doc && !a.first_tok && a.x_doc_off == a.doc_off && a.x_text == a.text;
    const bool chars = simple && a.char_mode && a.want_offsets;
    const bool masks = !a.pt_start;                      // (the starts off the start mask: pipeline.cpp)
    const bool ends = a.pt_end || (masks && a.endmask);
    // (behind BertNormalizer: the alignment map without per-byte ends; matches by tile -- kernels/output.hip NORIG)
    const bool norig_simple = masks && a.norig && !a.norig_e && !a.trim_offsets && !a.word_of_doc && !a.first_tok;
#define TKAMD_TM(E, S, C, M) launch_tm<E, S, C, M, false>(st, grid, a)
    if (norig_simple) { if (ends) launch_tm<true, true, false, true, true>(st, grid, a); else launch_tm<false, true, false, true, true>(st, grid, a); }
    else if (ends && masks) { if (chars) TKAMD_TM(true, true, true, true); else if (simple) TKAMD_TM(true, true, false, true); else TKAMD_TM(true, false, false, true); }
    else if (ends) { if (chars) TKAMD_TM(true, true, true, false); else if (simple) TKAMD_TM(true, true, false, false); else TKAMD_TM(true, false, false, false); }
    else if (masks) { if (chars) TKAMD_TM(false, true, true, true); else if (simple) TKAMD_TM(false, true, false, true); else TKAMD_TM(false, false, false, true); }
    else { if (chars) TKAMD_TM(false, true, true, false); else if (simple) TKAMD_TM(false, true, false, false); else TKAMD_TM(false, false, false, false); }
#undef TKAMD_TM
}
void launch_prefix_space(hipStream_t st, const uint8_t* text, const int64_t* seg_off, int64_t n_bound, const int64_t* n_dev, const unsigned long long* matchmask,
                         uint32_t* need, uint32_t* bsum, int64_t* xseg_off, int64_t* x_len, uint8_t* xtext, uint32_t* nos, uint32_t* noe, int grid) {
    unsigned nb = blocks_for(n_bound + 1, 256);
    hipLaunchKernelGGL(k_prefix_need, dim3(nb), dim3(256), 0, st, text, seg_off, n_bound, n_dev, matchmask, need);
    hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)need, n_bound + 1, bsum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, x_len);
    hipLaunchKernelGGL(k_prefix_doc_offsets, dim3(nb), dim3(256), 0, st, (const uint32_t*)need, n_bound, n_dev, (const uint32_t*)bsum, seg_off, xseg_off, x_len);
    hipLaunchKernelGGL(k_prefix_copy, dim3(grid), dim3(256), 0, st, text, seg_off, (const int64_t*)xseg_off, n_bound, n_dev, xtext, nos, noe);
}
void launch_add_u32(hipStream_t st, uint32_t* data, int64_t n, uint32_t delta) {
    if (n > 0) hipLaunchKernelGGL(k_add_u32, dim3(blocks_for(n, 256)), dim3(256), 0, st, data, n, delta);
}
void launch_add_i64(hipStream_t st, int64_t* data, int64_t n, int64_t delta) {
    hipLaunchKernelGGL(k_add_i64, dim3(blocks_for(n, 256)), dim3(256), 0, st, data, n, delta);
}
// one matching pass of the AddedVocabulary over the sentences seg_off[0 .. n_segs]: appends (start, stop, id) to match_list
void launch_added_match(hipStream_t st, const AddedArgs& a, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const int64_t* seg_off, int64_t n_segs,
                        const int64_t* n_segs_dev, const unsigned long long* skipmask, const uint16_t* uc1, const uint8_t* uc2, unsigned long long* candmask,
                        uint32_t* sents, uint32_t* n_sents, uint32_t* match_list, uint32_t* n_match, uint32_t cap, uint32_t len_flag, int* err) {
    hipLaunchKernelGGL(k_added_candidates, dim3(blocks_for(n_bytes + 1, 256 * 16)), dim3(256), 0, st, a, text, n_bytes, len_dev, candmask, (int*)nullptr);
    hipLaunchKernelGGL(k_l3_slow_docs, dim3(std::min<unsigned>(blocks_for(n_segs, 256), 4096u)), dim3(256), 0, st, (const unsigned long long*)candmask, seg_off, n_segs, n_segs_dev, sents, n_sents);
    hipLaunchKernelGGL(k_added_resolve, dim3(1024), dim3(64), 0, st, a, text, seg_off, (const uint32_t*)sents, (const uint32_t*)n_sents,
                       (const unsigned long long*)candmask, skipmask, uc1, uc2, match_list, n_match, cap, len_flag, err);
}
void launch_added_detect(hipStream_t st, const AddedArgs& a, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, int* note) {
    hipLaunchKernelGGL(k_added_candidates, dim3(blocks_for(n_bytes + 1, 256 * 16)), dim3(256), 0, st, a, text, n_bytes, len_dev, (unsigned long long*)nullptr, note);
}
void launch_scatter_matches(hipStream_t st, const uint32_t* list, const uint32_t* n_list, int64_t n_bytes, const int64_t* len_dev, unsigned long long* matchmask,
                            unsigned long long* spanmask, unsigned long long* stopmask, unsigned long long* hardmask, uint32_t* tmp_end, uint32_t* dirty) {
    hipLaunchKernelGGL(k_scatter_matches, dim3(256), dim3(256), 0, st, list, n_list, n_bytes, len_dev, matchmask, spanmask, stopmask, hardmask, tmp_end, dirty);
}
void launch_mask_or2(hipStream_t st, unsigned long long* dst, const unsigned long long* a, const unsigned long long* b, int64_t n_words) {
    hipLaunchKernelGGL(k_mask_or2, dim3(blocks_for(n_words, 256)), dim3(256), 0, st, dst, a, b, n_words);
}
void launch_emit_boundaries(hipStream_t st, const unsigned long long* mask, const uint32_t* wprefix, int64_t n_bytes, const int64_t* len_dev, const int64_t* total, int64_t* out) {
    hipLaunchKernelGGL(k_emit_boundaries, dim3(blocks_for((n_bytes >> 6) + 2, 256)), dim3(256), 0, st, mask, wprefix, n_bytes, len_dev, total, out);
}
void launch_translate_matches_norm(hipStream_t st, uint32_t* list, const uint32_t* n_list, const uint8_t* olen, const uint32_t* wbase, int64_t n_bytes, const int64_t* x_len) {
    hipLaunchKernelGGL(k_translate_matches_norm, dim3(256), dim3(256), 0, st, list, n_list, BnOlen{olen, bn_ltot_of((uint8_t*)olen, n_bytes)}, wbase, n_bytes, x_len);
}
void launch_translate_matches_prefix(hipStream_t st, uint32_t* list, const uint32_t* n_list, const unsigned long long* bmask, const uint32_t* wprefix, int64_t n_bytes,
                                     const int64_t* len_dev, const int64_t* total, const int64_t* xseg_off) {
    hipLaunchKernelGGL(k_translate_matches_prefix, dim3(256), dim3(256), 0, st, list, n_list, bmask, wprefix, n_bytes, len_dev, total, xseg_off);
}
void launch_prefix_doc_csr(hipStream_t st, const int64_t* doc_off, int64_t n_docs, const unsigned long long* bmask, const uint32_t* wprefix, int64_t n_bytes,
                           const int64_t* len_dev, const int64_t* total, const int64_t* xseg_off, int64_t* xdoc_off) {
    hipLaunchKernelGGL(k_prefix_doc_csr, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, bmask, wprefix, n_bytes, len_dev, total, xseg_off, xdoc_off);
}
void launch_mask_or(hipStream_t st, unsigned long long* dst, const unsigned long long* src, int64_t n_words, const uint32_t* n_list) {
    hipLaunchKernelGGL(k_mask_or, dim3(blocks_for(n_words, 256)), dim3(256), 0, st, dst, src, n_words, n_list);
}
void launch_apply_matches(hipStream_t st, unsigned long long* startmask, unsigned long long* endmask, const unsigned long long* matchmask,
                          const unsigned long long* spanmask, const unsigned long long* stopmask, int64_t n_words, const uint32_t* n_list) {
    hipLaunchKernelGGL(k_apply_matches, dim3(blocks_for(n_words, 256)), dim3(256), 0, st, startmask, endmask, matchmask, spanmask, stopmask, n_words, n_list);
}
void launch_apply_match_ids(hipStream_t st, const uint32_t* match_list, const uint32_t* n_match, const unsigned long long* startmask,
                            const uint32_t* wprefix, uint32_t* tok0) {
    hipLaunchKernelGGL(k_apply_match_ids, dim3(64), dim3(256), 0, st, match_list, n_match, startmask, wprefix, tok0);
}
int long_kernel_lds_bytes() { return LONG_PT_MAX * (4 + 4 + 4 + 2 + 2); }
int prepare_long_kernel() {
    int rc = (int)hipFuncSetAttribute((const void*)k_bpe_merge_long, hipFuncAttributeMaxDynamicSharedMemorySize, long_kernel_lds_bytes());
#define TKAMD_LU_ATTR(E, P) if (rc == 0) rc = (int)hipFuncSetAttribute((const void*)k_lookup<E, P>, hipFuncAttributeMaxDynamicSharedMemorySize, lookup_lds_bytes())
    TKAMD_LU_ATTR(true, false); TKAMD_LU_ATTR(false, false); TKAMD_LU_ATTR(true, true); TKAMD_LU_ATTR(false, true);
#undef TKAMD_LU_ATTR
    if (rc == 0) rc = prepare_lds_merge<16, 640, true, true>();
    if (rc == 0) rc = prepare_lds_merge<32, 768, true, true>();
    if (rc == 0) rc = prepare_lds_merge<16, 640, false, true>();
    if (rc == 0) rc = prepare_lds_merge<32, 768, false, true>();
    return rc;
}
void launch_bpe_merge_long(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v, void* rows,
                           uint32_t* tmp_ids, uint32_t* tmp_end, uint32_t* list_huge, uint32_t* n_huge,
                           uint32_t* scratch, unsigned long long scratch_words, unsigned long long* scratch_used, int* err) {
    hipLaunchKernelGGL(k_bpe_merge_long, dim3(grid), dim3(256), long_kernel_lds_bytes(), st, t, text, v, (uint4*)rows, tmp_ids, tmp_end, list_huge, n_huge);
    hipLaunchKernelGGL(k_bpe_merge_huge, dim3(64), dim3(256), 0, st, t, text, (const QItem*)v.q, (uint4*)rows, v.row_base, (const uint32_t*)list_huge,
                       (const uint32_t*)n_huge, tmp_ids, tmp_end, scratch, scratch_words, scratch_used, err);
}
void launch_bpe_merge_long_only(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v, void* rows, uint32_t* tmp_ids, uint32_t* tmp_end,
                                uint32_t* list_huge, uint32_t* n_huge) {
    hipLaunchKernelGGL(k_bpe_merge_long, dim3(grid), dim3(256), long_kernel_lds_bytes(), st, t, text, v, (uint4*)rows, tmp_ids, tmp_end, list_huge, n_huge);
}
void launch_zero_tail(hipStream_t st, uint8_t* p, const int64_t* len_dev, int n, unsigned long long* mask, int64_t mask_words, int grid) {
    hipLaunchKernelGGL(k_zero_tail, dim3(mask ? std::max(1, grid) : 1), dim3(256), 0, st, p, len_dev, n, mask, mask_words);
}
void launch_zero_regions(hipStream_t st, int grid, const ZeroRegions& z) {
    if (z.n > 0) hipLaunchKernelGGL(k_zero_regions, dim3(grid), dim3(256), 0, st, z);
}
int compact_grid(int n_cu) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_compact<CP_ITEMS_PER_LANE>, CP_NT, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    // (a CU holds 32 wavefronts: eight of these workgroups, whatever LDS and registers would allow)
    per_cu = std::min(per_cu, 32 / (CP_NT / 64));
    return per_cu * n_cu;
}
void launch_word_cache_insert(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v, const void* rows, const WordCache& wc) {
    hipLaunchKernelGGL(k_word_cache_insert, dim3(grid), dim3(256), 0, st, t, text, v, (const uint4*)rows, wc);
}
void launch_compact(hipStream_t st, int grid, const uint32_t* tok0, const void* rows, const void* cache_rows, const uint32_t* tmp_ids, const int64_t* n_pretok,
                    unsigned long long* state, int64_t* n_tok, uint32_t* pt_tokoff, uint32_t* ids, const uint32_t* chunk_lo, const uint32_t* doc_pt,
                    int64_t n_docs, int64_t* tok_offsets, void* phases, uint8_t* tok_b8) {
    static_assert(COMPACT_CHUNK == CpShape<CP_ITEMS_PER_LANE>::CHUNK, "the host sizes the look-back state and chunk_lo by the chunk");
    unsigned long long* const ph = (unsigned long long*)phases;
    // polls before a look-back computes a missing total itself (kernels/output.hip, results.hip); the test hook TKAMD_LB_PATIENCE sets it
    // to a handful so that the helping path runs on every wait
    const char* const e = test_hook("TKAMD_LB_PATIENCE");
    const uint32_t patience = e ? (uint32_t)std::max(0, atoi(e)) : LB_PATIENCE;
    if (ph)                                                  // the diagnostic instantiation (tkamd_debug_phases)
        hipLaunchKernelGGL((k_compact<CP_ITEMS_PER_LANE, true>), dim3(grid), dim3(CP_NT), 0, st, tok0, (const uint4*)rows, (const uint4*)cache_rows, tmp_ids, n_pretok, state, n_tok, pt_tokoff, ids,
                           chunk_lo, doc_pt, n_docs, tok_offsets, ph, patience, tok_b8);
    else
        hipLaunchKernelGGL(k_compact<CP_ITEMS_PER_LANE>, dim3(grid), dim3(CP_NT), 0, st, tok0, (const uint4*)rows, (const uint4*)cache_rows, tmp_ids, n_pretok, state, n_tok, pt_tokoff, ids,
                           chunk_lo, doc_pt, n_docs, tok_offsets, ph, patience, tok_b8);
}
