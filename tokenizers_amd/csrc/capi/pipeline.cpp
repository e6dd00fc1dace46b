// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): the kernel sequence of one batch (run_pipeline), its queues and workspace sizes, the synchronisation and error mapping.

// Queue capacities for a text of N bytes.  Every queue is NSQ sub-queues (results.hip), one per lookup workgroup; a workgroup
// takes every grid-th tile of LOOKUP_TILE_BYTES.  A pre-token of class 1 / 2 / 3 is longer than 16 / 32 / 64 bytes, so those three
// are sized for the worst case outright; the <= 16-byte queue (worst case: half the bytes) starts at 1 / q16_div of them and
// the batch is run again with the worst-case size if it ever overflows (ERR_QUEUE_FULL; natural text queues 1/50 .. 1/6).
// the lookup's grid: what is resident at once (kernels/lookup.hip: three workgroups a CU), one private sub-queue per workgroup
int lookup_grid(const tkamd_tokenizer* t) { return std::min(3 * t->n_cu, (int)NSQ); }

struct QueueSizes {
    uint32_t sq_cap[4], row_base[4];
    size_t total;
};
// What the error word of a batch that has just been read back asks for besides failing: the batch again with a larger <= 16-byte queue
// (ERR_QUEUE_FULL alone), or again with the added tokens' matching passes (NOTE_ADDED_SEEN: it was run as if its text held none).
static bool rerun_wanted(tkamd_tokenizer* t, Workspace* w, int raw) {
    const int err = raw & ~NOTE_BITS;
    bool again = false;
    if (raw & NOTE_ADDED_SEEN) { t->added_spec_pause = t->added_spec_len; w->force_general = true; again = true; }      // (force_general: the run that follows, whoever else draws on the pause)
    if (err == ERR_QUEUE_FULL && t->q16_div > 1) { t->q16_div = t->q16_div > 2 ? 2 : 1; again = true; }
    return again;
}

QueueSizes queue_sizes(size_t N, uint32_t q16_div, int grid) {
    QueueSizes z{};
    const size_t n_tiles = N / LOOKUP_TILE_BYTES + 1;
    const size_t per_sq = ((n_tiles + grid - 1) / grid) * LOOKUP_TILE_BYTES;
    z.sq_cap[0] = (uint32_t)(per_sq / q16_div + 64);
    z.sq_cap[1] = (uint32_t)(per_sq / 17 + 16);
    z.sq_cap[2] = (uint32_t)(per_sq / 33 + 16);
    z.sq_cap[3] = (uint32_t)(per_sq / 65 + 16);
    size_t acc = 0;
    for (int c = 0; c < 4; ++c) { z.row_base[c] = (uint32_t)acc; acc += (size_t)z.sq_cap[c] * (size_t)grid; }      // (sub-queues grid .. NSQ - 1 stay empty)
    z.total = acc;
    return z;
}

void reserve_workspace(tkamd_tokenizer* t, Workspace* w, int64_t n_bytes, int64_t n_docs, uint32_t flags, bool want_meta) {
    int64_t W = (n_bytes >> 6) + 2;
    size_t N = (size_t)n_bytes;
    w->w_docmask.reserve(W * 8);
    w->w_startmask.reserve(W * 8);
    w->w_wprefix.reserve(W * 4);
    w->w_bsum.reserve((W / 256 + 2) * 4);
    w->w_tok0.reserve((N + 4) * 4);
    w->w_tmp_ids.reserve((N + 4) * 4);
    const QueueSizes z = queue_sizes(N, t->q16_div, lookup_grid(t));
    w->w_rows.reserve(z.total * 16);
    w->w_queues.reserve(z.total * 8);
    w->w_cstate.reserve((N / COMPACT_CHUNK + 4) * 8 + 16);
    w->w_chunk_lo.reserve((N / COMPACT_CHUNK + 4) * 4);
    w->w_qcount.reserve((size_t)QCNT_WORDS * 4);
    w->w_pt_tokoff.reserve((N + 4) * 4);
    w->w_ids.reserve((N + 4) * 4);
    w->w_doc_pt.reserve((n_docs + 2) * 4);
    w->w_tok_offsets.reserve((n_docs + 2) * 8);
    w->w_scalars.reserve(SC_SLOTS * 8);
    if (want_meta) w->w_pt_start.reserve((N + 4) * 4);      // pre-token offsets exist in memory only for the offsets / word-id pass
    if (flags & TKAMD_OFFSETS_MASK) {
        w->w_tmp_end.reserve((N + 4) * 4);
        w->w_offsets.reserve((N + 4) * 8);
    }
    if (flags & TKAMD_WANT_WORD_IDS) w->w_word_ids.reserve((N + 4) * 4);
}

// Enqueue the whole path on `st`.  Inputs and outputs are device pointers.
//
// Coordinate spaces: the ORIGINAL text (what the caller passed, what offsets refer to) and the X text
// (what the pre-tokenizer and the model read).  X == original unless a normalizer ran (BertNormalizer:
// bytes deleted/replaced, w_norig maps back) or ByteLevel add_prefix_space inserted leading spaces
// (documents shifted, mapped back per document).  When X is derived its length only exists on the
// device (x_len_dev); kernels are launched over the host-side bound n_x and read the effective length.
// d_seq_off / n_seqs: is_pretokenized inputs (InputSequence::PreTokenized, tokenizer/mod.rs:782-795) -- the documents are the WORDS and
// sequence s is the words [d_seq_off[s], d_seq_off[s + 1]); n_seqs < 0: plain documents.
// d_inp_off / n_inputs: a Vec<EncodeInput> that mixes Single and Dual items (tokenizer/mod.rs:225-290, 1337-1356) -- input i is the
// sequences (documents, or sequences of words) [d_inp_off[i], d_inp_off[i + 1]), one or two of them; n_inputs < 0: one kind, per flags.
void run_pipeline(tkamd_tokenizer* t, Workspace* w, const uint8_t* d_text, const int64_t* d_doc_off, int64_t n_docs, int64_t n_bytes,
                  const int64_t* d_seq_off, int64_t n_seqs, uint32_t flags, hipStream_t st, tkamd_device_result* out,
                  const int64_t* d_inp_off = nullptr, int64_t n_inputs = -1) {
    HostModel& hm = t->hm;
    const int64_t* const d_doc_off_in = d_doc_off;         // as the caller passed them (the pipeline below works on validated copies)
    const int64_t* const d_seq_off_in = d_seq_off;
    const int64_t* const d_inp_off_in = d_inp_off;
    const bool mixed = n_inputs >= 0;
    if (mixed && (flags & TKAMD_PAIRS)) throw Invalid("a mixed batch names the kind of every input itself: TKAMD_PAIRS must not be set");
    if (mixed && !d_inp_off) throw Invalid("null input offsets");
    bool rerun = false;                                    // set by the overflow epilogue: a work queue was too small, run the batch again
    const uint32_t off_mode = flags & TKAMD_OFFSETS_MASK;
    const bool want_words = (flags & TKAMD_WANT_WORD_IDS) != 0;
    const bool want_meta = off_mode != TKAMD_OFFSETS_NONE || want_words;
    if (off_mode == 3u) throw Invalid("bad offsets mode");
    const bool add_special = (flags & TKAMD_ADD_SPECIAL) != 0 && !(hm.pp_prefix.empty() && hm.pp_suffix.empty());
    if ((flags & TKAMD_ADD_SPECIAL) && !(flags & TKAMD_PAIRS) && !hm.pp_unsupported.empty()) throw Unsupported("add_special_tokens: " + hm.pp_unsupported);
    if (!(flags & TKAMD_PAIRS) && hm.pp_single_refused) throw Unsupported("post_processor: " + hm.pp_unsupported);
    if (mixed && (flags & TKAMD_ADD_SPECIAL) && !hm.pp_pair_unsupported.empty()) throw Unsupported("add_special_tokens on a pair: " + hm.pp_pair_unsupported);
    const bool prefix_space = hm.byte_level && hm.add_prefix_space;
    // host-side bound of the X text length: +1 per document for the virtual space; BertNormalizer can grow a
    // character (CJK spacing: 3 -> 5 bytes, NFD/lowercase expansions <= 3x) -- 3x the input covers every case
    // added-token matches of a batch: at most one per min_len bytes (the shortest pattern)
    size_t at_min_len = (size_t)-1;
    for (int c = 0; c < 2; ++c)
        for (size_t k = 0; k + 1 < hm.at[c].off.size(); ++k) at_min_len = std::min<size_t>(at_min_len, hm.at[c].off[k + 1] - hm.at[c].off[k]);
    const bool have_added_tokens = at_min_len != (size_t)-1;
    const uint32_t mcap = have_added_tokens ? (uint32_t)std::min<size_t>((size_t)n_bytes / std::max<size_t>(at_min_len, 1) + 16, 0x7FFFFFF0u) : 0u;
    // (a prefix space goes in front of every piece: every document, and what follows every match)
    const int64_t n_x = (hm.norm == NORM_BERT) ? 3 * n_bytes + 64 : n_bytes + (prefix_space ? n_docs + (int64_t)mcap : 0);
    if (n_x >= (int64_t)0xFFFFFF00ll) throw Invalid("batch larger than 4 GiB: split it (byte offsets are 32-bit on the device)");
    const bool bpe_path = hm.model == MODEL_BPE && !hm.char_bpe && (hm.pretok == PT_BYTELEVEL_GPT2 || hm.pretok == PT_LLAMA3 || hm.pretok == PT_BYTELEVEL_NOREGEX);
    const bool local_pretok = hm.pretok == PT_WHITESPACE || hm.pretok == PT_WHITESPACE_SPLIT || hm.pretok == PT_BERT;
    const bool word_models = (hm.model == MODEL_WORDLEVEL || hm.model == MODEL_WORDPIECE) && local_pretok;
    const bool char_bpe = hm.model == MODEL_BPE && hm.char_bpe && local_pretok;      // BPE over characters rides the word models' pre-tokenizers
    if (!bpe_path && !word_models && !char_bpe)
        throw Unsupported("this build covers {ByteLevel(GPT-2 regex), Llama-3 Split+ByteLevel, ByteLevel(no regex)}+BPE and "
                          "{Whitespace,WhitespaceSplit,BertPreTokenizer}+{WordLevel,WordPiece,BPE over characters}");
    if (prefix_space && hm.norm != NORM_NONE) throw Unsupported("ByteLevel add_prefix_space behind a normalizer");

    reserve_workspace(t, w, n_x, n_docs, flags, want_meta);
    int64_t* sc = w->w_scalars.as<int64_t>();
    int64_t* d_npretok = sc + SC_NPRETOK;
    int64_t* d_ntok_total = sc + SC_NTOK;
    int64_t* d_xlen = sc + SC_NKEPT;
    int* d_err = (int*)(sc + SC_ERR);
    uint32_t* d_counters = (uint32_t*)(sc + SC_COUNTERS);
    const int64_t W0 = (n_bytes >> 6) + 1;      // mask words over the original text
    const int64_t W = (n_x >> 6) + 1;           // mask words over the X text
    const int grid = t->n_cu * 8;
    Prof pf{t, w, st};
    using ull = unsigned long long;

    // Everything the batch needs zeroed, in one launch: the scalars, the document mask, the queues' fill counters, the look-back state
    // of the compaction and the claims table (in-batch word claims, kernels/lookup.hip: on by default -- the test hook TKAMD_CLAIMS=0
    // switches them off, every occurrence of a word then goes to the model kernels; the word cache -- tkamd_word_cache, across batches --
    // takes their place when it is switched on).
    const char* const claims_hook = test_hook("TKAMD_CLAIMS");
    const bool claims_on = !(claims_hook && !strcmp(claims_hook, "0"));
    // A text made on the device (the normaliser's) has its length there; its masks and prefix counts are launched over the host's bound
    // and stop at the text's own length.
    constexpr bool len_bound = true;
    bool use_claims = claims_on && !t->word_cache &&
                      (hm.model == MODEL_BPE || (hm.model == MODEL_WORDPIECE && hm.max_input_chars >= (uint32_t)WORD_MAX_KEY));
    if (use_claims) {                                      // paused by an earlier batch that shared nothing (read_scalars)? one batch less to go
        int p = t->claims_pause.load();
        while (p > 0 && !t->claims_pause.compare_exchange_weak(p, p - 1)) {}
        if (p > 0) use_claims = false;
    }
    w->last_used_claims = use_claims;
    size_t claim_slots = 0;
    const size_t cstate_bytes = (((size_t)n_x / (size_t)COMPACT_CHUNK + 4) * 8 + 15) & ~(size_t)15;
    {
        ZeroRegions z{};
        z.add(sc, SC_SLOTS * 8);
        // (behind BertNormalizer the mask covers the bound of the normalised text, three times the input: the words that text really has
        // are zeroed behind the normaliser, next to the slack of the text -- launch_zero_tail below)
        if (!(len_bound && hm.norm == NORM_BERT)) z.add(w->w_docmask.p, (size_t)(W + 1) * 8);
        z.add(w->w_qcount.p, (size_t)QCNT_WORDS * 4);
        z.add(w->w_cstate.p, cstate_bytes);
        if (hm.pretok == PT_LLAMA3 && split_rule_fast(hm.split_rule)) {
            w->w_l3_tiles.reserve(l3_tileflag_words(n_x) * 8);
            z.add(w->w_l3_tiles.p, l3_tileflag_words(n_x) * 8);
        }
        if (use_claims) {
            // one slot per 64 bytes of the INPUT text (a word is a few bytes, most are repeats), 2^18 .. 2^24 slots: 32 MB of claims (two
            // 64-bit words a slot) + 32 MB of rows for a 120 MB batch.  (Not of the normalised text's bound, three times that behind BertNormalizer: the
            // words are the input's, and a table four times the size is four times the zeroing and a quarter of the cache hits.)
            // (a smaller table is less to zero and more of it in the caches, and more words whose slot another word holds)
            constexpr size_t per_slot = 64;
            int bits = 18;
            while (bits < 24 && ((size_t)1 << bits) < (size_t)n_bytes / per_slot) ++bits;
            claim_slots = (size_t)1 << bits;
            w->w_claims.reserve(claim_slots * 16);
            w->w_claim_rows.reserve(claim_slots * 16);
            z.add(w->w_claims.p, claim_slots * 16);
        }
        launch_zero_regions(st, t->n_cu * 4, z);
    }
    out->d_ids = w->w_ids.as<uint32_t>();
    out->d_tok_offsets = w->w_tok_offsets.as<int64_t>();
    out->d_offsets = nullptr;
    out->d_word_ids = nullptr;
    out->d_n_tokens = d_ntok_total;
    out->d_n_pretokens = d_npretok;
    out->ids_capacity = n_x + 4;          // what w_ids holds (a token covers a byte of the X text); the epilogues below size theirs from the data
    w->last_n_docs = n_docs;
    w->cur_trim1 = nullptr;
    w->last_n_enc = -1;
    out->d_enc_docs = nullptr;
    out->d_n_encodings = nullptr;
    w->last_seq_off = d_seq_off;
    w->last_n_seqs = n_seqs;
    w->last_inp_off = d_inp_off;
    w->last_n_inputs = n_inputs;
    // the caller's CSR is validated once; everything below reads the validated copy
    w->w_doc_off.reserve((size_t)(n_docs + 2) * 8);
    // The plain GPT-2 path (no added tokens, no normalizer, no prefix space: BASELINE configs[1] / [4]) reads the document CSR in two
    // places only: the document bitmask, and the documents' first pre-tokens.  The first is built by the validating kernel itself
    // (a bit only from a document that is consistent on its own: always inside the text), the second kernel writes the validated
    // copy on its way (it runs behind the whole validation, so it knows the verdict) -- two launches instead of four
    // (every other tokenizer takes the general order).  A malformed CSR still never turns into an access outside the buffers; the batch
    // fails with TKAMD_ERR_INVALID as before.
    const bool lean = n_bytes > 0 && hm.at[0].size() == 0 && hm.at[1].size() == 0 && hm.norm == NORM_NONE && !prefix_space &&
                      (hm.pretok == PT_BYTELEVEL_GPT2 || hm.pretok == PT_BYTELEVEL_NOREGEX);
    const int64_t* const raw_doc_off = d_doc_off;
    if (!lean) {
        pf.begin("validate_csr");
        launch_validate_csr(st, d_doc_off, n_docs, n_bytes, d_err, w->w_doc_off.as<int64_t>());
        pf.end();
    }
    d_doc_off = w->w_doc_off.as<int64_t>();
    // what the epilogues below see: one encoding per document, or per sequence of words
    const bool words_in = n_seqs >= 0;
    if (words_in) {
        if (!d_seq_off) throw Invalid("null sequence offsets");
        w->w_seq_off.reserve((size_t)(n_seqs + 2) * 8);
        w->w_seq_tok_off.reserve((size_t)(n_seqs + 2) * 8);
        launch_validate_csr(st, d_seq_off, n_seqs, n_docs, d_err, w->w_seq_off.as<int64_t>());     // a CSR over [0, n_words]
        d_seq_off = w->w_seq_off.as<int64_t>();
        out->d_tok_offsets = w->w_seq_tok_off.as<int64_t>();
    }
    const int64_t* const e_tok_off = words_in ? w->w_seq_tok_off.as<int64_t>() : w->w_tok_offsets.as<int64_t>();
    const int64_t e_n = words_in ? n_seqs : n_docs;
    if (mixed) {
        w->w_inp_off.reserve((size_t)(n_inputs + 2) * 8);
        launch_validate_csr(st, d_inp_off, n_inputs, e_n, d_err, w->w_inp_off.as<int64_t>());      // a CSR over [0, sequences]
        d_inp_off = w->w_inp_off.as<int64_t>();
    }
    auto add_specials = [&]() {
        // PostProcessor::process for a single sequence (processors/bert.rs:51-120, template.rs:544-590): specials around every document
        const size_t T2 = (size_t)n_x + 4 + (size_t)(e_n + 1) * (hm.pp_prefix.size() + hm.pp_suffix.size());
        w->w_ids2.reserve(T2 * 4);
        w->w_tok_offsets2.reserve((size_t)(e_n + 2) * 8);
        if (out->d_offsets) w->w_offsets2.reserve(T2 * 8);
        if (out->d_word_ids) w->w_word_ids2.reserve(T2 * 4);
        SpecialArgs sa{};
        sa.tok_offsets = e_tok_off;
        sa.n_docs = e_n;
        sa.ids = w->w_ids.as<uint32_t>();
        sa.offsets = out->d_offsets;
        sa.word_ids = out->d_word_ids;
        sa.prefix = t->t_pp_prefix.as<uint32_t>();
        sa.suffix = t->t_pp_suffix.as<uint32_t>();
        sa.n_prefix = (int32_t)hm.pp_prefix.size();
        sa.n_suffix = (int32_t)hm.pp_suffix.size();
        sa.tok_offsets2 = w->w_tok_offsets2.as<int64_t>();
        sa.ids2 = w->w_ids2.as<uint32_t>();
        sa.offsets2 = w->w_offsets2.as<uint32_t>();
        sa.word_ids2 = w->w_word_ids2.as<uint32_t>();
        sa.n_tok2 = sc + SC_NTOK2;
        pf.begin("add_specials");
        launch_add_specials(st, grid, sa);
        pf.end();
        out->d_ids = sa.ids2;
        out->ids_capacity = (int64_t)T2;
        out->d_tok_offsets = sa.tok_offsets2;
        if (out->d_offsets) out->d_offsets = sa.offsets2;
        if (out->d_word_ids) out->d_word_ids = sa.word_ids2;
        out->d_n_tokens = sa.n_tok2;
    };
    out->d_pad_counts = nullptr;
    out->d_type_ids = nullptr;
    out->d_seq_ids = nullptr;
    const bool pairs = (flags & TKAMD_PAIRS) != 0 || mixed;       // (a mixed batch: the pair epilogue lays out both kinds of input)
    if (pairs && !mixed && (e_n & 1)) throw Invalid("TKAMD_PAIRS: an odd number of documents");
    if (pairs && (flags & TKAMD_ADD_SPECIAL) && !hm.pp_pair_unsupported.empty()) throw Unsupported("add_special_tokens on a pair: " + hm.pp_pair_unsupported);
    const bool typed_single = !pairs && hm.pp_single_typed;          // the single template's type ids: written by the epilogue, with or without special tokens
    const bool epilogue = hm.trunc_on || hm.pad_on || pairs || typed_single;
    // Encoding.overflowing: what a truncation cuts off, as further encodings of the result (a pair leaves every combination of its two
    // sequences' windows, Encoding::merge_with encoding.rs:408-432)
    const bool want_overflow = (flags & TKAMD_WANT_OVERFLOW) != 0 && hm.trunc_on;
    out->d_enc_parts = nullptr;
    // BatchLongest (utils/padding.rs:55-63): the batch's longest encoding, read back from the device -- and, in a call that is sharded
    // over several devices, exchanged with the other shards' (Workspace::pad_exchange), the batch's written back for the kernels behind.
    // *again: a sharded call found its work queue too small -- the batch is run again BEFORE the exchange (every shard takes part in
    // it exactly once; finish_batch's later re-run would hand in a second value the others no longer wait for).
    auto batch_longest = [&](uint32_t* d_target, bool* again) -> uint64_t {
        int64_t head[SC_PADMAX + 1];
        HIP_CHECK(hipMemcpyAsync(head, sc, sizeof(head), hipMemcpyDeviceToHost, st));
        uint32_t mx = 0;
        if (d_target != (uint32_t*)(sc + SC_PADMAX)) HIP_CHECK(hipMemcpyAsync(&mx, d_target, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (d_target == (uint32_t*)(sc + SC_PADMAX)) mx = *(const uint32_t*)&head[SC_PADMAX];
        if (!w->pad_exchange) return mx;
        if (rerun_wanted(t, w, *(const int*)&head[SC_ERR])) {
            *again = true;
            return 0;
        }
        const uint32_t all = w->pad_exchange(mx);
        if (all != mx) {
            w->h_padmax = all;
            HIP_CHECK(hipMemcpyAsync(d_target, &w->h_padmax, 4, hipMemcpyHostToDevice, st));
        }
        return all;
    };
    auto finalize_pairs = [&]() {
        // EncodeInput::Dual: the two sequences of a pair were encoded as two documents; cut, lay out and pad them together
        const int64_t n_pairs = mixed ? n_inputs : e_n / 2;
        const bool tpl_on = (flags & TKAMD_ADD_SPECIAL) && !hm.pp_pair.empty();
        uint32_t n_special = 0;
        if (tpl_on) for (const HostModel::TplPiece& q : hm.pp_pair) n_special += q.kind == 2u;
        PairArgs pa{};
        pa.tok_offsets = e_tok_off;
        pa.n_pairs = n_pairs;
        if (mixed) {
            pa.inp_off = d_inp_off;
            pa.tpl1 = add_special ? t->t_pp_single.as<uint32_t>() : t->t_pp_single_plain.as<uint32_t>();
            pa.n_tpl1 = add_special ? (int32_t)hm.pp_single.size() : (int32_t)hm.pp_single_plain.size();
            pa.n_special1 = add_special ? (uint32_t)(hm.pp_prefix.size() + hm.pp_suffix.size()) : 0u;
        }
        const uint32_t n_special_max = std::max(n_special, pa.n_special1);      // (the bound of the output's size)
        pa.ids = w->w_ids.as<uint32_t>();
        pa.offsets = out->d_offsets;
        pa.word_ids = out->d_word_ids;
        pa.trim1 = pa.offsets ? w->cur_trim1 : nullptr;
        w->w_keep.reserve((size_t)(std::max(e_n, 2 * n_pairs) + 2) * 4);
        pa.tpl = tpl_on ? t->t_pp_pair.as<uint32_t>() : t->t_pp_pair_plain.as<uint32_t>();
        pa.n_tpl = tpl_on ? (int32_t)hm.pp_pair.size() : (int32_t)hm.pp_pair_plain.size();
        pa.n_special = n_special;
        pa.ovf_ty_tpl = (tpl_on && hm.pp_roberta) ? 1u : 0u;
        pa.trunc_on = hm.trunc_on ? 1u : 0u;
        pa.trunc_max = hm.trunc_max_length;
        pa.trunc_left = hm.trunc_left ? 1u : 0u;
        pa.trunc_strategy = (uint32_t)hm.trunc_strategy;
        pa.trunc_stride = hm.trunc_stride;
        pa.pad_on = hm.pad_on ? 1u : 0u;
        pa.pad_fixed = hm.pad_fixed ? 1u : 0u;
        pa.pad_length = hm.pad_length;
        pa.pad_multiple = hm.pad_multiple;
        pa.pad_left = hm.pad_left ? 1u : 0u;
        pa.pad_id = hm.pad_id;
        pa.pad_type_id = hm.pad_type_id;
        w->w_fbsum.reserve((size_t)((n_pairs + 1) / 256 + 2) * 4);
        pa.keep = w->w_keep.as<uint32_t>();
        pa.bsum = w->w_fbsum.as<uint32_t>();
        pa.target = (uint32_t*)(sc + SC_PADMAX);
        pa.n_tok2 = sc + SC_NTOK2;
        pa.err = d_err;
        for (int32_t k = 0; k < pa.n_tpl; ++k) {            // which sequence the template names first (it is "self" in the merge of the overflowing windows)
            const uint32_t kind = (tpl_on ? hm.pp_pair : hm.pp_pair_plain)[(size_t)k].kind;
            if (kind < 2u) { pa.first_is_b = kind == 1u ? 1u : 0u; break; }
        }
        pf.begin("pair_epilogue");
        int64_t n_enc = n_pairs;
        bool overflow = want_overflow;
        if (overflow) {
            w->w_ovf_parts.reserve((size_t)(n_pairs + 2) * 4);
            w->w_enc_base.reserve((size_t)(n_pairs + 2) * 8);
            pa.ovf_parts = w->w_ovf_parts.as<uint32_t>();
            pa.enc_base = w->w_enc_base.as<int64_t>();
        } else {
            w->w_len1.reserve((size_t)(n_pairs + 2) * 4);
            pa.len1 = w->w_len1.as<uint32_t>();
        }
        launch_pair_lens(st, pa);
        if (overflow) {
            launch_pair_overflow_scan(st, pa, sc + SC_NENC);
            int64_t head[SC_NENC + 1];
            HIP_CHECK(hipMemcpyAsync(head, sc, sizeof(head), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            const int err_now = *(const int*)&head[SC_ERR] & ~NOTE_BITS;
            if (rerun_wanted(t, w, *(const int*)&head[SC_ERR])) {     // (see finalize(): the batch is run again right away)
                rerun = true;
                pf.end();
                return;
            }
            if (err_now) {                                  // the batch fails when it is synchronised: finish it without the overflowing encodings
                overflow = false;
                pa.ovf_parts = nullptr;
                pa.enc_base = nullptr;
                w->w_len1.reserve((size_t)(n_pairs + 2) * 4);
                pa.len1 = w->w_len1.as<uint32_t>();
                launch_pair_lens(st, pa);
            } else {
                n_enc = head[SC_NENC];
                if (n_enc < n_pairs || n_enc >= ((int64_t)1 << 31)) throw Invalid("the truncation leaves more than 2^31 overflowing encodings: raise max_length - stride or split the batch");
                w->w_enc_doc.reserve((size_t)(n_enc + 2) * 4);
                w->w_enc_idx.reserve((size_t)(n_enc + 2) * 8);
                w->w_enc_win.reserve((size_t)(n_enc + 2) * 16);
                w->w_len1.reserve((size_t)(n_enc + 2) * 4);
                w->w_fbsum.reserve((size_t)((n_enc + 1) / 256 + 2) * 4);
                pa.enc_doc = w->w_enc_doc.as<uint32_t>();
                pa.enc_idx = w->w_enc_idx.as<uint32_t>();
                pa.enc_win = w->w_enc_win.as<uint32_t>();
                pa.len1 = w->w_len1.as<uint32_t>();
                pa.bsum = w->w_fbsum.as<uint32_t>();
            }
        }
        w->w_fin.reserve((size_t)(n_enc + 2) * 4);
        w->w_tok_offsets2.reserve((size_t)(n_enc + 2) * 8);
        if (hm.pad_on) w->w_pad_count.reserve((size_t)(n_enc + 2) * 4);
        pa.fin = w->w_fin.as<uint32_t>();
        pa.tok_offsets2 = w->w_tok_offsets2.as<int64_t>();
        pa.pad_count = hm.pad_on ? w->w_pad_count.as<uint32_t>() : nullptr;
        if (overflow) launch_pair_ranges(st, pa);           // (pa.n_pairs still counts pairs)
        FinalArgs fa{};                                    // the CSR of the padded lengths: same three kernels as for single sequences
        fa.n_docs = n_enc;
        fa.len1 = pa.len1; fa.fin = pa.fin; fa.bsum = pa.bsum; fa.target = pa.target; fa.tok_offsets2 = pa.tok_offsets2; fa.n_tok2 = pa.n_tok2;
        fa.pad_on = pa.pad_on; fa.pad_fixed = pa.pad_fixed; fa.pad_length = pa.pad_length; fa.pad_multiple = pa.pad_multiple;
        size_t T2 = (size_t)n_x + 4 + (size_t)(n_pairs + 1) * n_special_max;
        if (overflow) {
            launch_final_offsets(st, fa);
            int64_t total = 0;
            HIP_CHECK(hipMemcpyAsync(&total, fa.n_tok2, 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (total < 0 || (uint64_t)total >= ((uint64_t)1 << 32)) throw Invalid("the batch with its overflowing encodings would hold more than 2^32 tokens: encode fewer pairs per call");
            T2 = (size_t)total + 4;
        } else if (hm.pad_on) {
            uint64_t target = hm.pad_length;
            if (!hm.pad_fixed) {
                target = batch_longest(pa.target, &rerun);
                if (rerun) { pf.end(); return; }
            }
            if (hm.pad_multiple > 0 && target % hm.pad_multiple > 0) target += hm.pad_multiple - target % hm.pad_multiple;
            T2 += (size_t)n_pairs * (size_t)target;
            if ((uint64_t)T2 >= ((uint64_t)1 << 32)) throw Invalid("the padded batch would hold more than 2^32 tokens: pad fewer documents per call");
        }
        w->w_ids2.reserve(T2 * 4);
        w->w_type_ids2.reserve(T2 + 64);
        w->w_seq_ids2.reserve(T2 + 64);
        if (out->d_offsets) w->w_offsets2.reserve(T2 * 8);
        if (out->d_word_ids) w->w_word_ids2.reserve(T2 * 4);
        pa.ids2 = w->w_ids2.as<uint32_t>();
        pa.offsets2 = w->w_offsets2.as<uint32_t>();
        pa.word_ids2 = w->w_word_ids2.as<uint32_t>();
        pa.type_ids2 = w->w_type_ids2.as<uint8_t>();
        pa.seq_ids2 = w->w_seq_ids2.as<uint8_t>();
        if (!overflow) launch_final_offsets(st, fa);
        else pa.n_pairs = n_enc;                            // the copy runs per encoding
        launch_pair_finalize(st, grid, pa);
        pf.end();
        if (overflow) {
            w->last_n_enc = n_enc;
            out->d_enc_docs = pa.enc_doc;
            out->d_enc_parts = pa.enc_idx;
            out->d_n_encodings = sc + SC_NENC;
        }
        out->d_ids = pa.ids2;
        out->ids_capacity = 0;
        out->d_tok_offsets = pa.tok_offsets2;
        if (out->d_offsets) out->d_offsets = pa.offsets2;
        if (out->d_word_ids) out->d_word_ids = pa.word_ids2;
        out->d_n_tokens = pa.n_tok2;
        out->d_pad_counts = pa.pad_count;
        out->d_type_ids = pa.type_ids2;
        out->d_seq_ids = pa.seq_ids2;
    };
    auto finalize = [&]() {
        // truncation -> special tokens -> padding (tokenizer/mod.rs:1265-1317) as one epilogue over the token CSR
        const uint32_t n_add = add_special ? (uint32_t)(hm.pp_prefix.size() + hm.pp_suffix.size()) : 0u;
        FinalArgs fa{};
        fa.tok_offsets = e_tok_off;
        fa.n_docs = e_n;
        fa.ids = w->w_ids.as<uint32_t>();
        fa.offsets = out->d_offsets;
        fa.word_ids = out->d_word_ids;
        fa.trim1 = fa.offsets ? w->cur_trim1 : nullptr;
        fa.prefix = t->t_pp_prefix.as<uint32_t>();
        fa.suffix = t->t_pp_suffix.as<uint32_t>();
        fa.n_prefix = add_special ? (int32_t)hm.pp_prefix.size() : 0;
        fa.n_suffix = add_special ? (int32_t)hm.pp_suffix.size() : 0;
        // max_length - n_added_tokens when specials are added (mod.rs:1273-1279; the subtraction wraps in the reference's
        // release build when max_length is smaller: nothing is then truncated)
        fa.trunc_len = 0xFFFFFFFFu;
        if (hm.trunc_on) fa.trunc_len = (n_add && hm.trunc_max_length < n_add) ? 0xFFFFFFFFu : hm.trunc_max_length - n_add;
        fa.trunc_left = hm.trunc_left ? 1u : 0u;
        fa.trunc_needs_pair = (hm.trunc_on && hm.trunc_strategy == 2) ? 1u : 0u;
        fa.trunc_stride = hm.trunc_stride;
        fa.pad_on = hm.pad_on ? 1u : 0u;
        fa.pad_fixed = hm.pad_fixed ? 1u : 0u;
        fa.pad_length = hm.pad_length;
        fa.pad_multiple = hm.pad_multiple;
        fa.pad_left = hm.pad_left ? 1u : 0u;
        fa.pad_id = hm.pad_id;
        w->w_fbsum.reserve((size_t)((e_n + 1) / 256 + 2) * 4);
        fa.bsum = w->w_fbsum.as<uint32_t>();
        fa.target = (uint32_t*)(sc + SC_PADMAX);
        fa.n_tok2 = sc + SC_NTOK2;
        fa.err = d_err;
        pf.begin("truncate_pad");
        int64_t n_enc = e_n;                                   // encodings of the result
        bool overflow = want_overflow;
        if (overflow) {
            // how many encodings every document leaves -> their numbering; the total is read back because everything below is
            // sized and launched per encoding
            w->w_ovf_parts.reserve((size_t)(e_n + 2) * 4);
            w->w_enc_base.reserve((size_t)(e_n + 2) * 8);
            fa.ovf_parts = w->w_ovf_parts.as<uint32_t>();
            fa.enc_base = w->w_enc_base.as<int64_t>();
            launch_overflow_count(st, fa, sc + SC_NENC);
            int64_t head[SC_NENC + 1];
            HIP_CHECK(hipMemcpyAsync(head, sc, sizeof(head), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            const int err_now = *(const int*)&head[SC_ERR] & ~NOTE_BITS;
            if (rerun_wanted(t, w, *(const int*)&head[SC_ERR])) {
                // the token CSR is incomplete: this call is synchronous here anyway, so the batch is run again right away with the
                // larger queue (what finish_batch does for the calls that never wait)
                rerun = true;
                pf.end();
                return;
            }
            // any other error: the batch fails when it is synchronised; finish it without the overflowing encodings
            if (err_now) overflow = false;
            else n_enc = head[SC_NENC];
        }
        if (overflow) {
            if (n_enc < e_n || n_enc >= ((int64_t)1 << 31)) throw Invalid("the truncation leaves more than 2^31 overflowing encodings: raise max_length - stride or split the batch");
            w->w_enc_doc.reserve((size_t)(n_enc + 2) * 4);
            w->w_enc_start.reserve((size_t)(n_enc + 2) * 4);
            w->w_enc_cnt.reserve((size_t)(n_enc + 2) * 4);
            fa.enc_doc = w->w_enc_doc.as<uint32_t>();
            fa.enc_start = w->w_enc_start.as<uint32_t>();
            fa.enc_cnt = w->w_enc_cnt.as<uint32_t>();
            w->w_fbsum.reserve((size_t)((n_enc + 1) / 256 + 2) * 4);
            fa.bsum = w->w_fbsum.as<uint32_t>();
        }
        w->w_len1.reserve((size_t)(n_enc + 2) * 4);
        w->w_fin.reserve((size_t)(n_enc + 2) * 4);
        w->w_tok_offsets2.reserve((size_t)(n_enc + 2) * 8);
        if (hm.pad_on) w->w_pad_count.reserve((size_t)(n_enc + 2) * 4);
        fa.len1 = w->w_len1.as<uint32_t>();
        fa.fin = w->w_fin.as<uint32_t>();
        fa.tok_offsets2 = w->w_tok_offsets2.as<int64_t>();
        fa.pad_count = hm.pad_on ? w->w_pad_count.as<uint32_t>() : nullptr;
        if (overflow) {
            launch_overflow_ranges(st, fa);                    // (fa.n_docs still counts documents)
            fa.n_docs = n_enc;
        } else {
            launch_final_lens(st, fa);
        }
        // capacity of the padded arrays: known up front for Fixed; BatchLongest needs the batch maximum (one 4-byte read-back)
        size_t T2 = (size_t)n_x + 4 + (size_t)(e_n + 1) * n_add;
        if (overflow) {
            // overlapping windows: the token total is whatever the new CSR says (read back once it is built)
            launch_final_offsets(st, fa);
            int64_t total = 0;
            HIP_CHECK(hipMemcpyAsync(&total, fa.n_tok2, 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (total < 0 || (uint64_t)total >= ((uint64_t)1 << 32)) throw Invalid("the batch with its overflowing encodings would hold more than 2^32 tokens: encode fewer documents per call");
            T2 = (size_t)total + 4;
        } else if (hm.pad_on) {
            uint64_t target = hm.pad_length;
            if (!hm.pad_fixed) {
                target = batch_longest(fa.target, &rerun);
                if (rerun) { pf.end(); return; }
            }
            if (hm.pad_multiple > 0 && target % hm.pad_multiple > 0) target += hm.pad_multiple - target % hm.pad_multiple;
            T2 += (size_t)e_n * (size_t)target;
            if ((uint64_t)T2 >= ((uint64_t)1 << 32)) throw Invalid("the padded batch would hold more than 2^32 tokens: pad fewer documents per call");
        }
        w->w_ids2.reserve(T2 * 4);
        if (out->d_offsets) w->w_offsets2.reserve(T2 * 8);
        if (out->d_word_ids) w->w_word_ids2.reserve(T2 * 4);
        fa.ids2 = w->w_ids2.as<uint32_t>();
        fa.offsets2 = w->w_offsets2.as<uint32_t>();
        fa.word_ids2 = w->w_word_ids2.as<uint32_t>();
        if (typed_single) {
            w->w_type_ids2.reserve(T2 + 64);
            w->w_seq_ids2.reserve(T2 + 64);
            fa.type_ids2 = w->w_type_ids2.as<uint8_t>();
            fa.seq_ids2 = w->w_seq_ids2.as<uint8_t>();
            fa.prefix_ty = t->t_pp_prefix_ty.as<uint8_t>();
            fa.suffix_ty = t->t_pp_suffix_ty.as<uint8_t>();
            fa.seq_ty = hm.pp_seq_ty;
            fa.pad_type_id = hm.pad_type_id;
            out->d_type_ids = fa.type_ids2;
            out->d_seq_ids = fa.seq_ids2;
        }
        if (!overflow) launch_final_offsets(st, fa);
        launch_finalize(st, grid, fa);
        pf.end();
        if (overflow) {
            w->last_n_enc = n_enc;
            out->d_enc_docs = fa.enc_doc;
            out->d_n_encodings = sc + SC_NENC;
        }
        out->d_ids = fa.ids2;
        out->ids_capacity = 0;
        out->d_tok_offsets = fa.tok_offsets2;
        if (out->d_offsets) out->d_offsets = fa.offsets2;
        if (out->d_word_ids) out->d_word_ids = fa.word_ids2;
        out->d_n_tokens = fa.n_tok2;
        out->d_pad_counts = fa.pad_count;
    };
    if (n_bytes == 0) {
        // only empty documents: no tokens, but the post-processor still puts its specials around every one of them
        HIP_CHECK(hipMemsetAsync(w->w_tok_offsets.p, 0, (size_t)(n_docs + 1) * 8, st));
        if (words_in) HIP_CHECK(hipMemsetAsync(w->w_seq_tok_off.p, 0, (size_t)(n_seqs + 1) * 8, st));
        if (off_mode != TKAMD_OFFSETS_NONE) out->d_offsets = w->w_offsets.as<uint32_t>();
        if (want_words) out->d_word_ids = w->w_word_ids.as<uint32_t>();
        if (pairs) finalize_pairs();
        else if (epilogue) finalize();
        else if (add_special) add_specials();
        w->last_ntok_slot = (add_special || epilogue) ? SC_NTOK2 : SC_NTOK;
        HIP_CHECK(hipGetLastError());
        return;
    }

    // ---- AddedVocabulary::extract_and_normalize (added_vocabulary.rs:523-564) + normalizer + ByteLevel add_prefix_space ----
    // Three texts at most: the ORIGINAL one, the X text the pre-tokenizer reads (normalised, or shifted behind prefix spaces), and
    // in between -- with a normalizer -- nothing else: add_prefix_space behind a normalizer is refused above.  Matches are kept as
    // a list (start, stop, id) that is moved from text to text; the bitmasks are scattered from it in the text they are used in.
    const HostModel::PatternSet &setA = hm.at[0], &setB = hm.at[1];
    // Natural text holds no added token: a tokenizer that has some runs the batch as if it had none -- one detection pass per pattern set
    // (k_added_candidates with a note instead of a mask) where match / resolve / scatter / piece launches would find nothing -- and a batch
    // whose text does hold the content of one is run again with the passes below when it is synchronised (NOTE_ADDED_SEEN, finish_batch;
    // the handle's next added_spec_len batches then do not speculate).
    bool spec = (setA.size() > 0 || setB.size() > 0) && t->added_spec_len > 0 && !w->force_general && !(flags & TKAMD_NO_SPECULATION);
    w->force_general = false;
    if (spec) {
        int p = t->added_spec_pause.load();
        while (p > 0 && !t->added_spec_pause.compare_exchange_weak(p, p - 1)) {}
        if (p > 0) spec = false;
    }
    const bool have_raw = setA.size() > 0 && !spec, have_norm = setB.size() > 0 && !spec, have_added = have_raw || have_norm;
    const ull* matchmask = nullptr;
    uint32_t* mlist = nullptr;
    uint32_t* n_match = d_counters + CNT_MATCHES;
    size_t seg_cap = 0;                                        // bound of the number of pieces between document / match edges
    int64_t* d_nseg = sc + SC_NSEG;
    const size_t WX = (size_t)std::max(W0, W) + 2;             // mask words covering either text
    if (have_added) {
        seg_cap = (size_t)n_docs + 2 * (size_t)mcap + 2;
        DevBuf* masks[6] = {&w->w_candmask, &w->w_matchmask, &w->w_spanmask, &w->w_stopmask, &w->w_hardmask, &w->w_boundmask};
        bool grew = !w->w_mask_dirty.p;
        for (DevBuf* b : masks) { const size_t before = b->cap; b->reserve(WX * 8); grew = grew || b->cap != before; }
        // The four match masks are kept CLEAN between their uses: k_scatter_matches leaves "bits were set" in w_mask_dirty, and the zeroing
        // in front of the next scatter runs only then (natural text holds no special token: 240 MB of zeroing per C3 step went this way).
        // Fresh allocations hold anything: flagged dirty.
        w->w_mask_dirty.reserve(16);
        if (grew) HIP_CHECK(hipMemsetAsync(w->w_mask_dirty.p, 0xFF, 8, st));      // (dirty, as far as the buffers go)
        w->w_match_docs.reserve((seg_cap + 1) * 4);
        w->w_match_list.reserve(((size_t)mcap + 4) * 16);
        mlist = w->w_match_list.as<uint32_t>();
    }
    auto args_of = [&](int c) {
        AddedArgs a{t->t_at_blob[c].as<uint8_t>(), t->t_at_off[c].as<uint32_t>(), t->t_at_first[c].as<uint32_t>(), t->t_at_id[c].as<uint32_t>(),
                    t->t_at_flags[c].as<uint32_t>(), {0ull, 0ull, 0ull, 0ull}, 0u, {0u, 0u, 0u, 0u}, t->encode_special ? 1u : 0u};
        const std::vector<uint32_t>& first = hm.at[c].first;
        for (uint32_t b = 0; b < 256u && first.size() == 257; ++b)
            if (first[b + 1] > first[b]) {
                a.first_set[b >> 6] |= 1ull << (b & 63);
                if (a.n_first < 4u) a.first_byte[a.n_first] = b;
                ++a.n_first;
            }
        return a;
    };
    auto scatter_masks = [&](int64_t n_text, const int64_t* len_dev, bool with_end) {
        ull* m4[4] = {w->w_matchmask.as<ull>(), w->w_spanmask.as<ull>(), w->w_stopmask.as<ull>(), w->w_hardmask.as<ull>()};
        // (one launch for the four; over the RAW text -- no device-side length -- only the words that text has: the masks are sized for the
        // normalised text's bound, three times that.  Over a text with a device-side length the kernels downstream run over the bound.)
        constexpr bool lazy = true;
        const size_t zero_bytes = len_dev ? WX * 8 : std::min(WX, (size_t)(n_text >> 6) + 2) * 8;
        ZeroRegions z{};
        // (lazily: the WHOLE buffers -- the bits may be an earlier, larger batch's)
        const size_t cap4[4] = {w->w_matchmask.cap, w->w_spanmask.cap, w->w_stopmask.cap, w->w_hardmask.cap};
        for (int q = 0; q < 4; ++q) z.add(m4[q], lazy ? (cap4[q] & ~(size_t)15) : zero_bytes);
        if (lazy) z.only_if = w->w_mask_dirty.as<uint32_t>();
        launch_zero_regions(st, t->n_cu * 4, z);
        launch_scatter_matches(st, mlist, n_match, n_text, len_dev, m4[0], m4[1], m4[2], m4[3], with_end ? w->w_tmp_end.as<uint32_t>() : nullptr,
                               lazy ? w->w_mask_dirty.as<uint32_t>() : nullptr);
    };
    // pieces of a text: what lies between document edges and match edges (boundary mask = docmask | hardmask), as an int64 CSR
    auto build_pieces = [&](const int64_t* doc_csr, int64_t n_text, const int64_t* len_dev) -> const int64_t* {
        const int64_t Wt = (n_text >> 6) + 1;
        HIP_CHECK(hipMemsetAsync(w->w_boundmask.p, 0, WX * 8, st));
        launch_mark_doc_starts_n(st, doc_csr, n_docs, n_text, len_dev, w->w_boundmask.as<ull>(), d_err);
        launch_mask_or(st, w->w_boundmask.as<ull>(), w->w_hardmask.as<ull>(), Wt, n_match);
        w->w_bprefix.reserve((size_t)(Wt + 2) * 4);
        w->w_seg_off.reserve((seg_cap + 2) * 8);
        launch_mask_scan(st, w->w_boundmask.as<ull>(), Wt, w->w_bsum.as<uint32_t>(), w->w_bprefix.as<uint32_t>(), d_nseg);
        launch_emit_boundaries(st, w->w_boundmask.as<ull>(), w->w_bprefix.as<uint32_t>(), n_text, len_dev, d_nseg, w->w_seg_off.as<int64_t>());
        return w->w_seg_off.as<int64_t>();
    };

    if (have_added) HIP_CHECK(hipMemsetAsync(n_match, 0, 4, st));
    if (spec && setA.size() > 0) {
        pf.begin("added_token_match");
        launch_added_detect(st, args_of(0), d_text, n_bytes, nullptr, d_err);
        pf.end();
    }
    if (have_raw) {
        // pass 1: the tokens with normalized = false, over the raw documents
        pf.begin("added_token_match");
        launch_added_match(st, args_of(0), d_text, n_bytes, nullptr, d_doc_off, n_docs, nullptr, nullptr, t->dt.uc1, t->dt.uc2, w->w_candmask.as<ull>(),
                           w->w_match_docs.as<uint32_t>(), d_counters + CNT_MATCH_DOCS, mlist, n_match, mcap, MATCH_LEN_ORIG, d_err);
        pf.end();
    }

    const uint8_t* x_text = d_text;
    bool lead_done = false;                                    // the pre-tokenizer wrote the lead-byte mask on its way (char offsets)
    const int64_t* x_doc_off = d_doc_off;
    const int64_t* x_len_dev = nullptr;
    const uint32_t* norig = nullptr;
    const uint32_t* norig_e = nullptr;
    if (hm.norm == NORM_BERT || prefix_space) {
        w->w_ntext.reserve((size_t)n_x + TKAMD_TEXT_PAD);
        w->w_ndoc_off.reserve((size_t)(n_docs + 2) * 8);
        // (the prefix-space copy leaves nothing unwritten either, but only the normaliser's path has been taken through the tests without
        // this memset: k_zero_tail behind launch_bert_normalize zeroes the slack behind the text it wrote)
        if (hm.norm != NORM_BERT) HIP_CHECK(hipMemsetAsync(w->w_ntext.p, 0, (size_t)n_x + TKAMD_TEXT_PAD, st));
        // test hook TKAMD_POISON_NTEXT (with TKAMD_TEST_HOOKS=1): the normaliser's output buffer starts every batch as 0xFF, so a kernel that
        // reads it beyond *x_len + TEXT_PAD -- bounded by the host's n_x instead of the device length -- changes a result instead of
        // meeting zeros an earlier batch or the allocator happened to leave (tests/test_parity_gpu.py runs the BertNormalizer fixtures so)
        else if (test_hook("TKAMD_POISON_NTEXT")) {
            HIP_CHECK(hipMemsetAsync(w->w_ntext.p, 0xFF, (size_t)n_x + TKAMD_TEXT_PAD, st));
            // ... and so do the masks and prefix counts over that text: with TKAMD_LEN_BOUND only the words of its own length are written
            // (the document mask's are zeroed behind the normaliser), every reader must stop there too
            if (len_bound) HIP_CHECK(hipMemsetAsync(w->w_docmask.p, 0xFF, w->w_docmask.cap, st));
            HIP_CHECK(hipMemsetAsync(w->w_startmask.p, 0xFF, w->w_startmask.cap, st));
            HIP_CHECK(hipMemsetAsync(w->w_wprefix.p, 0xFF, w->w_wprefix.cap, st));
            if (w->w_endmask.p) HIP_CHECK(hipMemsetAsync(w->w_endmask.p, 0xFF, w->w_endmask.cap, st));
        }
        if (off_mode != TKAMD_OFFSETS_NONE) {
            w->w_norig.reserve(((size_t)n_x + 4) * 4);
            norig = w->w_norig.as<uint32_t>();
            // (behind BertNormalizer the END of a byte's original range follows from its start and the original text -- kernels/output.hip
            // norig_end: 4 bytes per normalised byte less to write and to read; the prefix-space copy keeps per-byte ends)
            if (hm.norm != NORM_BERT) {
                w->w_norig_e.reserve(((size_t)n_x + 4) * 4);
                norig_e = w->w_norig_e.as<uint32_t>();
            }
        }
    }
    if (hm.norm == NORM_BERT) {
        // ---- BertNormalizer: text -> normalised text + original byte range of every normalised byte; the matches of pass 1 are
        // not text (their split carries the raw slice): copied verbatim ----
        w->w_keepmask.reserve(bn_olen_bytes(n_bytes));          // olen + the per-lane totals: output bytes per source byte (kernels.hpp bn_olen_bytes)
        w->w_kprefix.reserve((size_t)(W0 + 1) * 4);             // wsum
        w->w_wbase.reserve((size_t)(W0 + 1) * 4);
        BnTables bt{t->t_bn1.as<uint16_t>(), t->t_bn2.as<uint8_t>(), t->t_bn_map.as<MergeSlot>(), hm.bn_mask, hm.bn_seed,
                    hm.bn_clean_text, hm.bn_handle_chinese, hm.bn_strip_accents, hm.bn_lowercase};
        const ull* verbatim = nullptr;
        if (have_raw) {
            scatter_masks(n_bytes, nullptr, false);
            launch_mask_or2(st, w->w_boundmask.as<ull>(), w->w_matchmask.as<ull>(), w->w_spanmask.as<ull>(), W0 + 1);
            verbatim = w->w_boundmask.as<ull>();
        }
        pf.begin("bert_normalize");
        launch_bert_normalize(st, bt, d_text, n_bytes, d_doc_off, n_docs, verbatim, w->w_keepmask.as<uint8_t>(), w->w_kprefix.as<uint32_t>(),
                              w->w_bsum.as<uint32_t>(), w->w_wbase.as<uint32_t>(), d_xlen, w->w_ntext.as<uint8_t>(), (uint32_t*)norig, (uint32_t*)norig_e,
                              w->w_ndoc_off.as<int64_t>(), d_err);
        launch_zero_tail(st, w->w_ntext.as<uint8_t>(), d_xlen, TKAMD_TEXT_PAD, len_bound ? w->w_docmask.as<ull>() : nullptr, W + 1, t->n_cu * 4);
        pf.end();
        if (have_raw) launch_translate_matches_norm(st, mlist, n_match, w->w_keepmask.as<uint8_t>(), w->w_wbase.as<uint32_t>(), n_bytes, d_xlen);
        x_text = w->w_ntext.as<uint8_t>();
        x_doc_off = w->w_ndoc_off.as<int64_t>();
        x_len_dev = d_xlen;
    }
    // n_in / len_in: the text the second pass (and the prefix-space copy) reads
    const int64_t n_in = hm.norm == NORM_BERT ? n_x : n_bytes;
    if (spec && setB.size() > 0) {
        pf.begin("added_token_match2");
        launch_added_detect(st, args_of(1), x_text, n_in, x_len_dev, d_err);
        pf.end();
    }
    if (have_norm) {
        // pass 2: the tokens with normalized = true, by their normalised patterns, over every piece pass 1 left (the whole documents
        // when it found nothing or there is no such token)
        const int64_t* seg = x_doc_off;
        const int64_t* nseg_dev = nullptr;
        int64_t nseg_bound = n_docs;
        if (have_raw) {
            scatter_masks(n_in, x_len_dev, false);
            seg = build_pieces(x_doc_off, n_in, x_len_dev);
            nseg_dev = d_nseg;
            nseg_bound = (int64_t)seg_cap;
        }
        pf.begin("added_token_match2");
        launch_added_match(st, args_of(1), x_text, n_in, x_len_dev, seg, nseg_bound, nseg_dev, have_raw ? w->w_matchmask.as<ull>() : nullptr, t->dt.uc1, t->dt.uc2,
                           w->w_candmask.as<ull>(), w->w_match_docs.as<uint32_t>(), d_counters + CNT_MATCH_DOCS2, mlist, n_match, mcap,
                           hm.norm == NORM_NONE ? MATCH_LEN_ORIG : 0u, d_err);
        pf.end();
    }
    if (have_added && !prefix_space) {
        scatter_masks(n_in, x_len_dev, off_mode != TKAMD_OFFSETS_NONE);
        matchmask = w->w_matchmask.as<ull>();
    }
    const int64_t* piece_off = nullptr;                        // sentence CSR for the Llama-3 sequential matcher when matches cut the documents
    const int64_t* piece_n_dev = nullptr;
    if (prefix_space) {
        // ---- ByteLevel add_prefix_space: every piece shifted behind its virtual leading space (byte_level.rs:120-125) ----
        const int64_t* seg = d_doc_off;
        const int64_t* nseg_dev = nullptr;
        int64_t nseg_bound = n_docs;
        if (have_added) {
            scatter_masks(n_bytes, nullptr, false);
            seg = build_pieces(d_doc_off, n_bytes, nullptr);
            nseg_dev = d_nseg;
            nseg_bound = (int64_t)seg_cap;
            w->w_xseg_off.reserve((seg_cap + 2) * 8);
        }
        w->w_need.reserve((size_t)(nseg_bound + 2) * 4);
        w->w_need_bsum.reserve((size_t)((nseg_bound + 1) / 256 + 2) * 4);
        int64_t* xseg = have_added ? w->w_xseg_off.as<int64_t>() : w->w_ndoc_off.as<int64_t>();
        pf.begin("prefix_space");
        launch_prefix_space(st, d_text, seg, nseg_bound, nseg_dev, have_added ? w->w_matchmask.as<ull>() : nullptr, w->w_need.as<uint32_t>(),
                            w->w_need_bsum.as<uint32_t>(), xseg, d_xlen, w->w_ntext.as<uint8_t>(), (uint32_t*)norig, (uint32_t*)norig_e, grid);
        if (have_added) {
            // documents and matches in the shifted text: both start at piece boundaries
            launch_prefix_doc_csr(st, d_doc_off, n_docs, w->w_boundmask.as<ull>(), w->w_bprefix.as<uint32_t>(), n_bytes, nullptr, d_nseg, xseg, w->w_ndoc_off.as<int64_t>());
            launch_translate_matches_prefix(st, mlist, n_match, w->w_boundmask.as<ull>(), w->w_bprefix.as<uint32_t>(), n_bytes, nullptr, d_nseg, xseg);
        }
        pf.end();
        x_text = w->w_ntext.as<uint8_t>();
        x_doc_off = w->w_ndoc_off.as<int64_t>();
        x_len_dev = d_xlen;
        if (have_added) {
            scatter_masks(n_x, x_len_dev, off_mode != TKAMD_OFFSETS_NONE);
            matchmask = w->w_matchmask.as<ull>();
            piece_off = xseg;
            piece_n_dev = d_nseg;
        }
    } else if (have_added && hm.pretok == PT_LLAMA3) {
        piece_off = build_pieces(x_doc_off, n_in, x_len_dev);
        piece_n_dev = d_nseg;
    }

    pf.begin("mark_doc_starts");
    launch_mark_doc_starts_n(st, lean ? raw_doc_off : x_doc_off, n_docs, n_x, x_len_dev, w->w_docmask.as<ull>(), d_err);
    if (matchmask) launch_mask_or(st, w->w_docmask.as<ull>(), w->w_hardmask.as<ull>(), W, n_match);   // match edges are hard boundaries
    pf.end();

    uint32_t* pt_end = nullptr;       // explicit pre-token ends in memory (offsets pass of the "Removed" pre-tokenizers)
    bool has_end = false;             // the pre-tokenizer produced an end bitmask
    bool meta_masks = false;          // offsets / word ids: k_token_meta reads the pre-tokens' starts off the start mask (no pt_start array)
    auto after_masks = [&]() {        // what reads the start mask and its prefix counts: behind the pre-tokenizer + scan
        if (want_meta && !meta_masks) {
            // the pre-token offsets themselves are only materialised for the offsets / word-id pass of the pre-tokenizers WITH an end mask
            // (and of BPE over characters without an unk_token: k_token_meta_seq); the others' k_token_meta reads the start mask itself
            pf.begin("emit_pretok");
            launch_emit_pretok(st, w->w_startmask.as<ull>(), w->w_wprefix.as<uint32_t>(), n_x, x_len_dev, d_npretok, w->w_pt_start.as<uint32_t>());
            if (pt_end) launch_emit_pretok_end(st, w->w_startmask.as<ull>(), w->w_endmask.as<ull>(), w->w_wprefix.as<uint32_t>(), n_x, x_len_dev, pt_end);
            pf.end();
        }
        pf.begin("doc_first_pretok");
        launch_doc_first_pretok(st, lean ? raw_doc_off : x_doc_off, n_docs, n_x, w->w_startmask.as<ull>(), w->w_wprefix.as<uint32_t>(),
                                d_npretok, w->w_doc_pt.as<uint32_t>(), w->w_chunk_lo.as<uint32_t>(),
                                lean ? d_err : nullptr, lean ? w->w_doc_off.as<int64_t>() : nullptr);
        pf.end();
    };
    if (hm.pretok == PT_BYTELEVEL_GPT2) {
        pf.begin("pretok_gpt2_seq");
        // (char offsets over a text the pre-tokenizer reads as it came: the lead-byte mask rides along)
        if (off_mode == TKAMD_OFFSETS_CHAR && x_text == d_text && !x_len_dev && n_x == n_bytes) {
            w->w_leadmask.reserve((size_t)(W0 + 1) * 8);
            lead_done = true;
        }
        launch_pretok_gpt2(st, x_text, n_x, x_len_dev, w->w_docmask.as<ull>(), t->dt.uc1, t->dt.uc2, w->w_startmask.as<ull>(), lead_done ? w->w_leadmask.as<ull>() : nullptr);
        pf.end();
    } else if (hm.pretok == PT_LLAMA3) {
        w->w_endmask.reserve((size_t)(W + 1) * 8);          // reused as the "unresolved" mask
        w->w_slow_docs.reserve((size_t)(n_docs + 1) * 4);
        // (char offsets over a text the pre-tokenizer reads as it came: the lead-byte mask rides in the lane kernel of the bit-parallel members)
        if (off_mode == TKAMD_OFFSETS_CHAR && x_text == d_text && !x_len_dev && n_x == n_bytes && (split_rule_fast(hm.split_rule) || (split_rule_fast_cs(hm.split_rule) && t->t_ucc1.p))) {
            w->w_leadmask.reserve((size_t)(W0 + 1) * 8);
            lead_done = true;
        }
        pf.begin("pretok_llama3");
        w->w_slow_docs.reserve((size_t)((piece_off ? seg_cap : (size_t)n_docs) + 1) * 4);
        launch_pretok_llama3(st, x_text, n_x, x_len_dev, w->w_docmask.as<ull>(), t->dt.uc1, t->dt.uc2, w->w_startmask.as<ull>(),
                             w->w_endmask.as<ull>(), piece_off ? piece_off : x_doc_off, piece_off ? (int64_t)seg_cap : n_docs, piece_n_dev,
                             w->w_slow_docs.as<uint32_t>(), d_counters + CNT_SLOW_DOCS, hm.split_rule,
                             t->t_ucc1.p ? t->t_ucc1.as<uint16_t>() : nullptr, t->t_ucc2.p ? t->t_ucc2.as<uint8_t>() : nullptr,
                             w->w_l3_tiles.p ? w->w_l3_tiles.as<ull>() : nullptr, lead_done ? w->w_leadmask.as<ull>() : nullptr);
        pf.end();
    } else if (hm.pretok == PT_BYTELEVEL_NOREGEX) {
        // ByteLevel(use_regex=false): every document is one pre-token (byte_level.rs:128-130)
        HIP_CHECK(hipMemcpyAsync(w->w_startmask.p, w->w_docmask.p, (size_t)W * 8, hipMemcpyDeviceToDevice, st));
    } else {
        w->w_endmask.reserve((size_t)(W + 1) * 8);
        has_end = true;
        if (want_meta) {
            w->w_pt_end.reserve(((size_t)n_x + 4) * 4);
            pt_end = w->w_pt_end.as<uint32_t>();
        }
        pf.begin("pretok_local");
        launch_pretok_local(st, (int)hm.pretok, x_text, n_x, x_len_dev, w->w_docmask.as<ull>(), t->dt.uc1, t->dt.uc2,
                            w->w_startmask.as<ull>(), w->w_endmask.as<ull>(), len_bound);
        pf.end();
    }
    if (matchmask)
        launch_apply_matches(st, w->w_startmask.as<ull>(), has_end ? w->w_endmask.as<ull>() : nullptr, matchmask, w->w_spanmask.as<ull>(),
                             w->w_stopmask.as<ull>(), W, n_match);
    {
        pf.begin("mask_scan");
        // (three launches: reduce, a one-workgroup scan of the totals, down.  A single-pass ticket + look-back kernel in their place measured
        // 0.031 ms against 0.015: 917 tickets on one address and a look-back chain cost more than two launch gaps, profiles/r4a_*)
        meta_masks = want_meta && !(hm.char_bpe && !hm.unk_configured && !hm.byte_fallback);       // (that one: k_token_meta_seq, from pt_start / pt_end)
        if (meta_masks) w->w_tile_w.reserve(((size_t)n_x / META_TILE + 4) * 4);
        launch_mask_scan(st, w->w_startmask.as<ull>(), W, w->w_bsum.as<uint32_t>(), w->w_wprefix.as<uint32_t>(), d_npretok, len_bound ? x_len_dev : nullptr,
                         meta_masks ? w->w_tile_w.as<uint32_t>() : nullptr);
        pf.end();
        after_masks();
    }

    uint32_t* tmp_end = (off_mode != TKAMD_OFFSETS_NONE) ? w->w_tmp_end.as<uint32_t>() : nullptr;
    const size_t N = (size_t)n_x;
    const QueueSizes qz = queue_sizes(N, t->q16_div, lookup_grid(t));
    // (TKAMD_ROW_LIMIT_BITS: a test lowers the threshold -- never the 30 bits tok0 really has -- to see the refusal without a 3 GB batch)
    static const size_t row_limit = [] { const char* e = test_hook("TKAMD_ROW_LIMIT_BITS"); return e ? std::min<size_t>((size_t)1 << std::max(8, atoi(e)), ROW_INDEX_LIMIT) : (size_t)ROW_INDEX_LIMIT; }();
    if (qz.total >= row_limit) throw Invalid("batch too large for the work queues (row indices are 30-bit: about 3 GB of text): split it");
    QueuePlan plan{};
    for (int c = 0; c < 4; ++c) {
        plan.v[c].q = (QItem*)(w->w_queues.as<uint8_t>() + (size_t)qz.row_base[c] * 8);
        plan.v[c].counts = w->w_qcount.as<uint32_t>() + (size_t)c * NSQ * QCNT_STRIDE;
        plan.v[c].sq_cap = qz.sq_cap[c];
        plan.v[c].row_base = qz.row_base[c];
    }
    const ull* endmask = has_end ? w->w_endmask.as<ull>() : nullptr;
    // test hook TKAMD_PHASES: the lookup and the compaction run as their diagnostic instantiations, which add the shader-clock
    // ticks of their phases to a table of this workspace (tkamd_debug_phases reads and clears it); never in a measured run
    const bool phases_on = test_hook("TKAMD_PHASES") != nullptr;
    auto phases_of = [&](int which) -> void* {
        if (!phases_on) return nullptr;
        if (!w->w_phases.p) {
            w->w_phases.reserve(2 * PHASE_WGS * 64);
            HIP_CHECK(hipMemsetAsync(w->w_phases.p, 0, 2 * PHASE_WGS * 64, st));
        }
        return (uint8_t*)w->w_phases.p + (size_t)which * PHASE_WGS * 64;
    };
    WordCache wc{nullptr, nullptr, nullptr, 0u, nullptr};
    // (claims: see the top of this function; with offsets k_token_meta takes the token ends of a shared row from the claimant's slots of tmp_end)
    auto open_word_cache = [&]() {
        const size_t slots = (size_t)1 << WORD_CACHE_BITS;
        if (use_claims) {
            uint32_t* cpos = nullptr;                        // (the claimants' first bytes: only k_token_meta wants them)
            if (off_mode != TKAMD_OFFSETS_NONE) { w->w_claim_pos.reserve(claim_slots * 4); cpos = w->w_claim_pos.as<uint32_t>(); }
            wc = WordCache{nullptr, w->w_claim_rows.p, (unsigned long long*)w->w_claims.p, (uint32_t)(claim_slots - 1), cpos};
            return;
        }
        if (!t->word_cache || off_mode != TKAMD_OFFSETS_NONE) return;        // (a cached row carries no token ends)
        w->w_cache_keys.reserve(slots * sizeof(CacheKey));
        w->w_cache_rows.reserve(slots * 16);
        const uint64_t epoch = t->cache_epoch;
        if (w->cache_epoch != epoch) {
            HIP_CHECK(hipMemsetAsync(w->w_cache_keys.p, 0, slots * sizeof(CacheKey), st));
            w->cache_epoch = epoch;
        }
        wc = WordCache{(CacheKey*)w->w_cache_keys.p, w->w_cache_rows.p, nullptr, 0u, nullptr};
    };
    // the model kernels end an entry by publishing its row if it holds a claim (bpe.hip claim_publish_item)
    DevTables mdt = t->dt;
    mdt.err = d_err;
    mdt.probes = t->prof ? d_counters + CNT_MERGE_PROBES : nullptr;
    auto set_publish = [&]() {
        if (wc.claims) { mdt.pub_rows = wc.rows; mdt.pub_mask = wc.claim_mask; mdt.pub_pos = wc.claim_pos; }
    };
    if (hm.model == MODEL_BPE) {
        pf.begin("lookup");
        open_word_cache();
        set_publish();
        launch_lookup(st, lookup_grid(t), t->dt, x_text, n_x, x_len_dev, w->w_startmask.as<ull>(), endmask, w->w_wprefix.as<uint32_t>(),
                      w->w_tok0.as<uint32_t>(), plan, d_err, matchmask, t->t_hot.p, wc, 0u, 0u, phases_of(0), d_counters);
        pf.end();
        if (hm.ignore_merges)                              // vocab.get(sequence) for pre-tokens beyond the 16-byte keys (bpe/model.rs:559-567)
            launch_long_vocab3(st, t->n_cu, t->dt, x_text, plan.v[1], plan.v[2], plan.v[3], w->w_rows.p, 0u, d_err, wc);
        // the LDS kernels need new_id = rank + c (true of every trainer-made vocabulary); otherwise -- and under the test hook
        // TKAMD_FORCE_LANE_MERGE -- the register-resident lane kernels run
        const bool lds16 = t->dt.newid_affine && !test_hook("TKAMD_FORCE_LANE_MERGE");      // keys in LDS
        const bool lds32 = lds16;
        // With the claims on both queues hold the distinct words only, and a launch of the LDS kernels lasts as long as its longest word's
        // chain of dependent merge probes whatever it holds: the 32-symbol kernel takes both queues in one launch.  Thin or not is only
        // known on the device: while the handle has not seen a thin <= 16-byte queue (its first batch, or text that repeats nothing) BOTH
        // kernels are launched and pick the queue's owner from its fill themselves (thin_limit; an extra ~4 us launch); once a batch came
        // back thin the next ones launch the 32-symbol kernel alone, until a fat one is seen again.  (Test hook TKAMD_MERGE_TWO: always two
        // launches, each with its own queue.)
        const bool can_one = wc.claims && lds16 && lds32 && !test_hook("TKAMD_MERGE_TWO");
        const bool both = can_one && t->q16_fat_hint.load() != 0;
        const bool one = can_one && !both;
        if (both) mdt.thin_limit = MERGE_THIN_LIMIT;
        // BPE over characters: only the kernels that know its start (kernels/bpe.hip CHARS) -- the two LDS kernels, each on its own queue,
        // and the workgroup-per-pre-token kernel for everything beyond 32 bytes (or for everything, when the vocabulary's new ids are not
        // in merge order and the LDS kernels cannot run)
        if (hm.char_bpe) {
            mdt.thin_limit = 0u;                           // (each queue has its one kernel here)
            w->w_huge.reserve(64);
            w->w_list_huge.reserve(64);
            auto long_only = [&](const QView& q) {
                launch_bpe_merge_long_only(st, t->n_cu * 2, mdt, x_text, q, w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, w->w_list_huge.as<uint32_t>(), d_counters + CNT_LISTH);
            };
            pf.begin("bpe_merge_lds32");
            if (t->dt.newid_affine) launch_bpe_merge(st, t->n_cu, 6, mdt, x_text, plan.v[1], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, nullptr);
            else long_only(plan.v[1]);
            pf.end();
            pf.begin("bpe_merge_lds");
            if (t->dt.newid_affine) launch_bpe_merge(st, t->n_cu, 5, mdt, x_text, plan.v[0], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end);
            else long_only(plan.v[0]);
            pf.end();
            pf.begin("bpe_merge_long");
            long_only(plan.v[2]);
            long_only(plan.v[3]);
            pf.end();
        } else {
        pf.begin(lds32 ? "bpe_merge_lds32" : "bpe_merge_lane32");
        launch_bpe_merge(st, lds32 ? t->n_cu : grid, lds32 ? 6 : 2, mdt, x_text, plan.v[1], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, (one || both) ? &plan.v[0] : nullptr);
        pf.end();
        if (!one) {
            pf.begin(lds16 ? "bpe_merge_lds" : "bpe_merge_lane");
            launch_bpe_merge(st, lds16 ? t->n_cu : grid, lds16 ? 5 : 1, mdt, x_text, plan.v[0], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end);
            pf.end();
        }
        pf.begin("bpe_merge64");
        launch_bpe_merge(st, grid, 64, mdt, x_text, plan.v[2], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end);
        pf.end();
        pf.begin("bpe_merge_long");
        // pre-tokens beyond the LDS path (> 8192 B) run from a global scratch slab: 5 words per symbol, sized for the
        // worst case this batch can contain (the whole X text being such pre-tokens), capped at 1 GiB
        const size_t huge_words = std::min<size_t>((size_t)6 * N + 4096, (size_t)1 << 28);
        if (N > (size_t)LONG_PT_MAX) {
            w->w_huge.reserve(huge_words * 4);
            w->w_list_huge.reserve((N / LONG_PT_MAX + 16) * 4);
        } else {
            w->w_huge.reserve(64);
            w->w_list_huge.reserve(64);
        }
        launch_bpe_merge_long(st, t->n_cu, mdt, x_text, plan.v[3], w->w_rows.p,
                              w->w_tmp_ids.as<uint32_t>(), tmp_end, w->w_list_huge.as<uint32_t>(), d_counters + CNT_LISTH, w->w_huge.as<uint32_t>(),
                              (unsigned long long)(N > (size_t)LONG_PT_MAX ? huge_words : 0), (unsigned long long*)(sc + SC_HUGE_USED), d_err);
        pf.end();
        }
        if (wc.keys) {
            pf.begin("word_cache_insert");
            launch_word_cache_insert(st, grid, mdt, x_text, plan.v[0], w->w_rows.p, wc);
            pf.end();
        }
    } else if (hm.model == MODEL_WORDLEVEL) {
        // WordLevel::tokenize (wordlevel/mod.rs:162-178) is the lookup itself: every hit is final, a miss is the unk id
        DevTables wt = t->dt;
        wt.ignore_merges = 1;
        pf.begin("wordlevel_lookup");
        launch_lookup(st, lookup_grid(t), wt, x_text, n_x, x_len_dev, w->w_startmask.as<ull>(), endmask, w->w_wprefix.as<uint32_t>(),
                      w->w_tok0.as<uint32_t>(), plan, d_err, matchmask, t->t_hot.p, WordCache{nullptr, nullptr, nullptr, 0u, nullptr}, 0u, 1u, nullptr, nullptr);
        launch_long_vocab3(st, t->n_cu, wt, x_text, plan.v[1], plan.v[2], plan.v[3], w->w_rows.p, 1u, d_err, WordCache{nullptr, nullptr, nullptr, 0u, nullptr});      // words longer than 16 bytes
        pf.end();
    } else {
        // WordPiece's first candidate is the whole word (wordpiece/mod.rs:245-258 starts at end = len): the whole-word lookup
        // settles most words with one probe; only the rest walk the trie.  With max_input_chars_per_word < 16 a whole-word
        // hit could belong to a word over the limit, so every word takes the walk (which counts the chars).
        const bool shortcut = hm.max_input_chars >= (uint32_t)WORD_MAX_KEY;
        DevTables wt = t->dt;
        wt.ignore_merges = 1;                              // any whole-word hit is final
        wt.long_probe_max_len = hm.max_input_chars;        // len <= limit  =>  chars <= limit
        // (the reference keeps no cache for WordPiece; a word's pieces depend on nothing but the word, so the same table serves. With
        // every word taking the walk -- max_input_chars_per_word < 16 -- the lookup probes nothing, the cache included.)
        if (shortcut) open_word_cache();
        set_publish();
        pf.begin("wordpiece_word_lookup");
        launch_lookup(st, lookup_grid(t), wt, x_text, n_x, x_len_dev, w->w_startmask.as<ull>(), endmask, w->w_wprefix.as<uint32_t>(),
                      w->w_tok0.as<uint32_t>(), plan, d_err, matchmask, t->t_hot.p, wc, shortcut ? 0u : 1u, 0u, phases_of(0), d_counters);
        pf.end();
        pf.begin("wordpiece");
        launch_wordpiece_all(st, grid, t->n_cu, mdt, x_text, plan, w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, d_err);      // the <= 16-byte queue and the words longer than that, side by side
        pf.end();
        if (wc.keys) {
            pf.begin("word_cache_insert");
            launch_word_cache_insert(st, grid, t->dt, x_text, plan.v[0], w->w_rows.p, wc);
            pf.end();
        }
    }
    if (matchmask)
        launch_apply_match_ids(st, w->w_match_list.as<uint32_t>(), d_counters + CNT_MATCHES, w->w_startmask.as<ull>(),
                               w->w_wprefix.as<uint32_t>(), w->w_tok0.as<uint32_t>());
    // with offsets: one byte per token next to the ids -- the boundary in front of the token, where its row carried it (results.hip)
    uint8_t* tok_b8 = nullptr;
    if (tmp_end) { w->w_tok_b8.reserve((size_t)n_x + 64); tok_b8 = w->w_tok_b8.as<uint8_t>(); }
    pf.begin("compact");
    // (the token offsets of the pre-tokens are only materialised for the offsets / word-id pass; the documents' token CSR comes out of the compaction itself)
    launch_compact(st, t->cp_grid, w->w_tok0.as<uint32_t>(), w->w_rows.p, wc.rows, w->w_tmp_ids.as<uint32_t>(), d_npretok, w->w_cstate.as<ull>(),
                   d_ntok_total, want_meta ? w->w_pt_tokoff.as<uint32_t>() : nullptr, w->w_ids.as<uint32_t>(), w->w_chunk_lo.as<uint32_t>(),
                   w->w_doc_pt.as<uint32_t>(), n_docs, w->w_tok_offsets.as<int64_t>(), (size_t)t->cp_grid <= PHASE_WGS ? phases_of(1) : nullptr, tok_b8);
    pf.end();
    const uint32_t* word_of_doc = nullptr;
    const int64_t* first_tok = nullptr;
    if (words_in) {
        // the words' token CSR -> the sequences'; the word id of a token is its word's index in the sequence
        if (want_words) { w->w_word_idx.reserve((size_t)(n_docs + 2) * 4); word_of_doc = w->w_word_idx.as<uint32_t>(); }
        if (off_mode != TKAMD_OFFSETS_NONE && hm.trim_offsets) { w->w_first_tok.reserve((size_t)(n_docs + 2) * 8); first_tok = w->w_first_tok.as<int64_t>(); }
        launch_seq_regroup(st, d_seq_off, n_seqs, n_docs, w->w_tok_offsets.as<int64_t>(), w->w_seq_tok_off.as<int64_t>(), (uint32_t*)word_of_doc, (int64_t*)first_tok);
    }
    if (want_meta) {
        MetaArgs a{};
        a.word_of_doc = word_of_doc;
        a.first_tok = first_tok;
        a.x_text = x_text;
        a.text = d_text;
        a.pt_start = meta_masks ? nullptr : w->w_pt_start.as<uint32_t>();
        a.pt_end = meta_masks ? nullptr : pt_end;
        a.startmask = w->w_startmask.as<ull>();
        a.endmask = has_end ? w->w_endmask.as<ull>() : nullptr;
        a.wprefix = w->w_wprefix.as<uint32_t>();
        a.tile_w = meta_masks ? w->w_tile_w.as<uint32_t>() : nullptr;
        a.n_mask_words = W;
        a.x_len_dev = x_len_dev;
        a.x_len_host = n_x;
        a.n_tok = d_ntok_total;
        a.pt_tokoff = w->w_pt_tokoff.as<uint32_t>();
        a.tmp_end = tmp_end;
        a.tok_b8 = tok_b8;
        a.tok0 = wc.claims ? w->w_tok0.as<uint32_t>() : nullptr;
        a.claim_pos = wc.claims ? wc.claim_pos : nullptr;
        a.n_pretok = d_npretok;
        a.doc_pt = w->w_doc_pt.as<uint32_t>();
        a.chunk_lo = w->w_chunk_lo.as<uint32_t>();
        a.chunk = (uint32_t)COMPACT_CHUNK;
        a.n_docs = n_docs;
        a.x_doc_off = x_doc_off;
        a.doc_off = d_doc_off;
        a.norig = norig;                                   // normalised / shifted text: every byte's original byte range
        a.norig_e = norig_e;
        a.byte_level = hm.byte_level;
        a.snap_chars = hm.byte_level || hm.char_bpe;
        if (hm.char_bpe && !hm.unk_configured && !hm.byte_fallback) {      // (chars can be dropped: offsets are running sums)
            a.char_id = t->dt.char_id;
            a.cb = t->dt.cb;
            if (hm.ignore_merges) { a.ww_tok0 = w->w_tok0.as<uint32_t>(); a.ww_rows = w->w_rows.p; a.ww_crows = wc.rows; }      // (... but not on a whole-word hit)
        }
        a.trim_offsets = hm.trim_offsets;
        a.trim_matches_only = !hm.byte_level;            // (a model that is not byte-level: only an added token's slice can hold what is trimmed; the loader checked the vocabulary)
        a.pp_add_prefix_space = hm.pp_add_prefix_space;
        a.want_offsets = off_mode != TKAMD_OFFSETS_NONE;
        a.char_mode = off_mode == TKAMD_OFFSETS_CHAR;
        a.want_words = want_words;
        a.matchmask = matchmask;
        a.uc1 = t->dt.uc1;
        a.uc2 = t->dt.uc2;
        a.offsets = w->w_offsets.as<uint32_t>();
        a.word_ids = w->w_word_ids.as<uint32_t>();
        if (a.want_offsets && a.trim_offsets && a.pp_add_prefix_space && hm.trunc_on) {      // (see MetaArgs::trim1)
            w->w_trim1.reserve((size_t)n_x + 8);
            a.trim1 = w->w_trim1.as<uint8_t>();
            w->cur_trim1 = a.trim1;
        }
        if (a.char_mode) {
            w->w_leadmask.reserve((size_t)(W0 + 1) * 8);
            w->w_lprefix.reserve((size_t)(W0 + 1) * 4);
            pf.begin("leadmask_scan");
            if (!lead_done) launch_leadmask(st, d_text, n_bytes, w->w_leadmask.as<ull>());
            launch_mask_scan(st, w->w_leadmask.as<ull>(), W0, w->w_bsum.as<uint32_t>(), w->w_lprefix.as<uint32_t>(), sc + SC_NCHARS);
            pf.end();
            a.leadmask = w->w_leadmask.as<ull>();
            a.lprefix = w->w_lprefix.as<uint32_t>();
        }
        pf.begin("token_meta");
        launch_token_meta(st, grid, a);
        pf.end();
        if (a.want_offsets) out->d_offsets = a.offsets;
        if (a.want_words) out->d_word_ids = a.word_ids;
    }
    if (pairs) finalize_pairs();
    else if (epilogue) finalize();
    else if (add_special) add_specials();
    if (rerun) {
        run_pipeline(t, w, d_text, d_doc_off_in, n_docs, n_bytes, d_seq_off_in, n_seqs, flags, st, out, d_inp_off_in, n_inputs);
        return;
    }
    w->last_ntok_slot = (add_special || epilogue) ? SC_NTOK2 : SC_NTOK;
    HIP_CHECK(hipGetLastError());
}

int read_scalars(tkamd_tokenizer* t, Workspace* w, hipStream_t st, int64_t* n_tok, int64_t* n_pretok);

// Wait for the batch enqueued last; if its <= 16-byte work queue overflowed (ERR_QUEUE_FULL), grow the queue and run the
// same call again on the same stream (the output buffers are sized for the worst case, so the result pointers stay).
int finish_batch(tkamd_tokenizer* t, Workspace* w, hipStream_t st, int64_t* n_tok, int64_t* n_pretok) {
    int bits = read_scalars(t, w, st, n_tok, n_pretok);
    // (rerun_wanted: half the bytes covers every text whose queued pre-tokens have two bytes or more (a word and its separator); one entry
    // per byte covers the rest (runs of one-byte pre-tokens the vocabulary does not know, e.g. punctuation under WordPiece); a speculative
    // batch that met an added token's content is run again with the matching passes -- whatever else its error word says: that run decides)
    while (rerun_wanted(t, w, bits | (w->last_note_added ? NOTE_ADDED_SEEN : 0))) {
        tkamd_device_result again{};
        run_pipeline(t, w, w->last_text, w->last_doc_off, w->last_n_docs, w->last_n_bytes, w->last_seq_off, w->last_n_seqs, w->last_flags, st, &again,
                     w->last_inp_off, w->last_n_inputs);
        if (again.d_ids != w->last_result.d_ids || again.d_tok_offsets != w->last_result.d_tok_offsets ||
            again.d_offsets != w->last_result.d_offsets || again.d_word_ids != w->last_result.d_word_ids || again.d_pad_counts != w->last_result.d_pad_counts || again.d_type_ids != w->last_result.d_type_ids ||
            again.d_enc_docs != w->last_result.d_enc_docs) {
            // (buffers sized from the data -- the padded / overflowing encodings -- may have grown; a device-entry caller already
            // holds the old pointers, the host entry reads w->last_result after this)
            if (w->device_bound) throw HipError("result buffers moved while a batch was run again");
            w->last_result = again;
        }
        bits = read_scalars(t, w, st, n_tok, n_pretok);
    }
    return bits;
}

int read_scalars(tkamd_tokenizer* t, Workspace* w, hipStream_t st, int64_t* n_tok, int64_t* n_pretok) {
    int64_t host[SC_SLOTS];
    HIP_CHECK(hipMemcpyAsync(host, w->w_scalars.p, sizeof(host), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    int err = *(int*)&host[SC_ERR] & ~NOTE_BITS;             // (notes of the normalizer and of the added tokens' speculation, not errors)
    w->last_note_added = (*(int*)&host[SC_ERR] & NOTE_ADDED_SEEN) != 0;
    memcpy(w->last_counters, &host[SC_COUNTERS], sizeof(w->last_counters));
    if (w->last_used_claims && t->claims_pause_len > 0) {
        // the claims' yield, counted by the lookup itself: candidates it looked at and how many of them were another pre-token's word.
        // Fewer than one in four shared: the round trips cost more than the merges they save (tkamd_tokenizer::claims_pause)
        const uint64_t cands = w->last_counters[CNT_CLAIM_CANDS], shared = w->last_counters[CNT_CLAIM_SHARED];
        if (cands >= 32768 && shared * 4 < cands) t->claims_pause = t->claims_pause_len;
        t->q16_fat_hint = w->last_counters[CNT_CLAIM_CANDS] - w->last_counters[CNT_CLAIM_SHARED] >= MERGE_THIN_LIMIT ? 1 : 0;   // (survivors: an upper bound of the queue's fill)
    }
    if (n_tok) *n_tok = host[w->last_ntok_slot];
    if (n_pretok) *n_pretok = host[SC_NPRETOK];
    return err;
}

int error_from_bits(int bits) {
    if (bits & ERR_BAD_OFFSETS) return set_error(TKAMD_ERR_INVALID, "doc_offsets is not a monotone CSR over [0, n_bytes]");
    if (bits & ERR_PRETOKEN_TOO_LONG)
        return set_error(TKAMD_ERR_UNSUPPORTED, "pre-tokens longer than 8192 bytes exceed the 1 GiB scratch slab of the global-memory merge path");
    if (bits & ERR_NON_ASCII_NORM)
        return set_error(TKAMD_ERR_UNSUPPORTED, "BertNormalizer strip_accents: a character with a non-zero combining class that survives the Mn filter "
                                                "stands in a run of more than 48 combining characters; NFD's canonical ordering of such a run is not built "
                                                "on the device");
    if (bits & ERR_ADDED_SPLIT) return set_error(TKAMD_ERR_INVALID, "AddedVocabulary bad split");
    if (bits & ERR_INTERNAL) return set_error(TKAMD_ERR_DEVICE, "internal invariant violated");
    if (bits & ERR_QUEUE_FULL) return set_error(TKAMD_ERR_DEVICE, "work queues still too small after growing them");
    if (bits & ERR_TRUNC_SECOND) return set_error(TKAMD_ERR_INVALID, "Truncation error: Second sequence not provided");
    if (bits & ERR_TRUNC_SHORT) return set_error(TKAMD_ERR_INVALID, "Truncation error: Sequence to truncate too short to respect the provided max_length");
    if (bits & ERR_TRUNC_STRIDE)
        return set_error(TKAMD_ERR_INVALID, "`stride` must be strictly less than `max_len` (note that `max_len` may be shorter than the max length of the "
                                            "original model, as it subtracts the number of special characters");
    if (bits & ERR_TOO_MANY_TOKENS) return set_error(TKAMD_ERR_INVALID, "a truncation leaves more than 2^32 overflowing encodings of one sequence");
    if (bits & ERR_MISSING_UNK) return set_error(TKAMD_ERR_MODEL, "MissingUnkToken: the model needed an unknown token but the vocabulary has none");
    if (bits & ERR_UNK_OOV) return set_error(TKAMD_ERR_MODEL, "UnkTokenOutOfVocabulary: Unk token not found in the vocabulary");
    if (bits & ERR_INPUT_KIND) return set_error(TKAMD_ERR_INVALID, "input_offsets: every input of a mixed batch is one sequence or two");
    return TKAMD_OK;
}

template <class F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const Unsupported& e) {
        return set_error(TKAMD_ERR_UNSUPPORTED, e.what());
    } catch (const Invalid& e) {
        return set_error(TKAMD_ERR_INVALID, e.what());
    } catch (const HipError& e) {
        return set_error(TKAMD_ERR_DEVICE, e.what());
    } catch (const std::bad_alloc&) {
        return set_error(TKAMD_ERR_DEVICE, "out of host memory");
    } catch (const std::exception& e) {
        return set_error(TKAMD_ERR_INVALID, e.what());
    }
}
