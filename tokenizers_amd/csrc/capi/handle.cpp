// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): Workspace, tkamd_tokenizer, the pinned-block pool, tkamd_batch / tkamd_text.

// Everything one encode / decode call writes: intermediate and result buffers in HBM (grow-only), the stream of the host entry, the
// call's bookkeeping.  A tokenizer handle owns a small pool of them, so calls from different host threads run concurrently
// (TokenizerImpl::encode_batch is &self + Send + Sync, tokenizer/mod.rs:1328-1335); the tables stay shared and read-only.
struct Workspace {
    std::mutex mu;               // a workspace serves one call at a time
    // a sharded call with BatchLongest padding: the epilogue hands its shard's maximum to the call's MaxExchange here and pads to what
    // comes back (null: the batch is this workspace's alone)
    std::function<uint32_t(uint32_t)> pad_exchange;
    uint32_t h_padmax = 0;       // (the exchanged maximum on its way back to the device)
    bool busy = false;           // taken by a host-entry call
    bool device_bound = false;   // belongs to the device entry: keyed by the caller's stream, results stay valid in it
    hipStream_t bound_stream = nullptr;
    hipStream_t own_stream = nullptr;   // host entry: its own non-blocking stream
    // host entry, sliced: every H2D of a call goes down one stream in slice order and every D2H down another, so that the two
    // directions of the link run side by side and neither waits behind the other in a compute stream's order (encode_host)
    hipStream_t io_in = nullptr, io_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    // (sized by the largest batch seen)
    DevBuf w_l3_tiles;                          // a bit per 2048 bytes: tiles the Llama-3 family's lane kernel left bytes undecided in (zeroed with the batch's scratch)
    DevBuf w_docmask, w_startmask, w_wprefix, w_bsum, w_tile_w, w_pt_start, w_tok0, w_pt_tokoff, w_tmp_ids, w_tmp_end, w_rows;
    DevBuf w_len1, w_fin, w_fbsum, w_pad_count, w_keep, w_type_ids2, w_seq_ids2;   // truncation / padding / pair epilogue
    DevBuf w_ovf_parts, w_enc_base, w_enc_doc, w_enc_start, w_enc_cnt;             // overflowing encodings (TKAMD_WANT_OVERFLOW)
    DevBuf w_enc_idx, w_enc_win;                                                   // ... of pairs: window indices / token windows of A and B
    DevBuf w_queues, w_qcount, w_cstate;   // work queues (start, length) of the model kernels + their fill counters; look-back state of the compaction
    DevBuf w_doc_off;            // validated copy of the caller's document CSR
    DevBuf w_chunk_lo;                           // first document of every compaction chunk (k_doc_first_pretok -> k_compact)
    DevBuf w_ids, w_doc_pt, w_tok_offsets, w_scalars, w_offsets, w_word_ids;
    DevBuf dw_ids, dw_tok_off, dw_first, dw_bad, dw_dup, dw_len, dw_bsum, dw_pos, dw_out_off, dw_bytes, dw_total;   // decode_batch workspace
    DevBuf w_endmask, w_pt_end, w_keepmask, w_kprefix, w_ntext, w_norig, w_ndoc_off, w_slow_docs, w_leadmask, w_lprefix, w_need, w_need_bsum, w_huge, w_list_huge, w_wbase, w_norig_e, w_ids2, w_tok_offsets2, w_offsets2, w_word_ids2, w_candmask, w_matchmask, w_spanmask, w_stopmask, w_hardmask, w_boundmask, w_bprefix, w_seg_off, w_xseg_off,
        w_match_docs, w_match_list;
    // host entry staging
    DevBuf h_text, h_doc_off, h_seq_off, h_inp_off;
    DevBuf w_tok_b8;                             // with offsets: per token, the boundary byte its row carried (kernels/results.hip row_boundary)
    DevBuf w_trim1;                              // per token: process_offsets took one leading space off it (MetaArgs::trim1)
    const uint8_t* cur_trim1 = nullptr;          // ... of the batch being enqueued, or null
    DevBuf w_cache_keys, w_cache_rows;           // word cache of this workspace (kernels.hpp WordCache)
    DevBuf w_claims, w_claim_rows, w_claim_pos;  // in-batch word claims (kernels.hpp WordCache::claims), the rows of the claimed slots, the claimants' first bytes
    DevBuf w_phases;                             // TKAMD_PHASES: shader-clock ticks per phase of the lookup / compaction, [2][PHASE_WGS][8] u64 (tkamd_debug_phases)
    uint64_t cache_epoch = 0;                    // the tokenizer's cache_epoch these were last cleared at (0: never)
    DevBuf w_seq_off, w_seq_tok_off, w_word_idx, w_first_tok;      // is_pretokenized: validated sequence CSR over the words, the sequences' token CSR, word index of every word
    // profiling records of this workspace's launches, folded into the tokenizer's totals when drained
    std::vector<StageRec> pending;
    // last call (for tkamd_device_sync, which runs it again if a work queue overflowed)
    const uint8_t* last_text = nullptr;
    const int64_t* last_doc_off = nullptr;
    const int64_t* last_seq_off = nullptr;      // is_pretokenized call: the sequence CSR (else null)
    int64_t last_n_seqs = -1;
    const int64_t* last_inp_off = nullptr;      // mixed call: the inputs' CSR over the sequences (else null)
    int64_t last_n_inputs = -1;
    DevBuf w_inp_off;                           // ... its validated copy
    DevBuf w_mask_dirty;                        // one word: the four added-token match masks may hold bits (run_pipeline scatter_masks)
    int64_t last_n_bytes = 0;
    uint32_t last_flags = 0;
    tkamd_device_result last_result{};
    int64_t last_n_docs = 0;
    int64_t last_n_enc = -1;                    // encodings of the last call when it materialised overflowing ones, else -1
    int last_ntok_slot = 1;
    uint32_t last_counters[CNT_COUNT] = {0};
    bool force_general = false;                  // the next run of the pipeline does not speculate on the added tokens (it repeats a batch that met one)
    bool last_note_added = false;                // the batch synchronised last was speculative and met an added token's content (read_scalars)
    bool last_used_claims = false;               // the batch enqueued last ran with the in-batch claims
    ~Workspace() {
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (io_in) (void)hipStreamDestroy(io_in);
        if (io_out) (void)hipStreamDestroy(io_out);
        for (int i = 0; i < 2; ++i) {
            if (ev_in[i]) (void)hipEventDestroy(ev_in[i]);
            if (ev_out[i]) (void)hipEventDestroy(ev_out[i]);
        }
    }
};

struct tkamd_tokenizer {
    HostModel hm;
    int device = -1;
    DevTables dt{};
    std::mutex mu;                       // pool, profile totals
    std::condition_variable cv;
    std::vector<std::unique_ptr<Workspace>> pool;
    Workspace* last_used = nullptr;      // workspace of the most recent call (diagnostics: tkamd_profile_counters)
    // tables
    DevBuf t_ucc1, t_ucc2;               // case classes of a case-split Split pattern (HostModel::ucc_stage1 / 2), else empty
    DevBuf t_uc1, t_uc2, t_byte_id, t_merges, t_long_blob, t_long_off, t_long_id, t_long_table;
    DevBuf t_hot;                // hot-word table of the lookup kernel (copied into LDS)
    DevBuf t_shortw, t_shortw_k3, t_shortw_disp;   // the short-word table: 16-byte slots, key bytes 12..15, eight-bit displacements (tables.hpp SHORTW_*)
    DevBuf t_char_id;            // BPE over characters: HostModel::char_id
    DevBuf t_at_id[2], t_at_flags[2], t_at_blob[2], t_at_off[2], t_at_first[2];   // AddedVocabulary patterns of the two matching passes
    DevBuf t_pp_single, t_pp_single_plain;      // the single layout as pieces (the single inputs of a mixed batch)
    DevBuf t_pp_pair, t_pp_pair_plain;   // pair template of the post-processor with / without its special tokens: [pieces][3]
    DevBuf t_pp_prefix, t_pp_suffix, t_pp_prefix_ty, t_pp_suffix_ty, t_bn1, t_bn2, t_bn_map, t_merge_disp, t_dec_entry, t_dec_blob, t_trie;
    int n_cu = 256;
    int n_direct = 0;
    int n_hot = 0;
    int cp_grid = 0;             // grid of k_compact: what is resident at once (any grid makes progress -- its look-back helps itself --, TKAMD_CP_GRID)
    // In-batch claims on text that shares nothing (every candidate word distinct): the claim traffic then buys nothing and costs a third
    // of the step (the claims' worst case).  Inside a batch every lookup workgroup gives the claims up by itself once
    // it has seen that (kernels/lookup.hip CLAIM_ADAPT_MIN); across batches, a batch that ran with the claims and found fewer than a
    // quarter of its candidates shared pauses them for the next claims_pause_len batches of the handle; then they are tried again.
    std::atomic<int> q16_fat_hint{1};    // the last batch that ran with the claims left a fat <= 16-byte queue (or none has run yet): see run_pipeline's merge launches
    std::atomic<int> claims_pause{0};
    // A tokenizer with added tokens runs a batch as if its text held none (one detection pass per pattern set instead of match / resolve /
    // scatter / piece launches that find nothing in natural text); a batch that did hold one is run again with the matching passes and
    // the handle's next added_spec_len batches do not speculate (run_pipeline, finish_batch).
    std::atomic<int> added_spec_pause{0};
    int added_spec_len = 32;     // (test hook TKAMD_ADDED_SPEC: 0 never speculate; n: the pause behind a miss)
    int claims_pause_len = 32;   // (test hook TKAMD_CLAIMS_PAUSE; 0: never pause)
    std::atomic<uint32_t> q16_div{4};    // capacity of the <= 16-byte queue = n_bytes / q16_div (raised to the worst case when a batch overflows it)
    // profiling
    std::atomic<bool> prof{false};
    std::atomic<bool> encode_special{false};    // tkamd_encode_special_tokens (Tokenizer.encode_special_tokens): special tokens in the text are not extracted
    std::atomic<bool> word_cache{false};        // tkamd_word_cache: BPE words merged by earlier batches are looked up instead of merged again
    std::atomic<uint64_t> cache_epoch{1};       // bumped by a clear: every workspace zeroes its cache before its next batch
    std::vector<tkamd_stage_time> acc;
    // ---- multi-device handle (tkamd_tokenizer_from_json_devices): this object is the replica on devices[0]; replicas[r - 1] holds the
    // tables on devices[r].  One host-entry call then shards its documents over all of them (encode_host_sharded).
    std::vector<int> devices;
    std::vector<std::unique_ptr<tkamd_tokenizer>> replicas;
    std::mutex group_mu;                 // one sharded call at a time (it already uses every device)
    std::atomic<int> collect{0};         // TKAMD_COLLECT_*
    std::string collect_note;            // why the handle left TKAMD_COLLECT_ROOT_RCCL for the peer copies (written under group_mu)
    DevBuf g_root[8];                    // COLLECT_ROOT_*: the whole result on devices[0] before its one D2H (indexed like the descriptors of the call)
    std::vector<void*> rccl_comms;       // ncclComm_t per device of the handle (COLLECT_ROOT_RCCL, made at first use)
    int64_t shard_min_bytes = 1 << 20;   // a batch of less than this per device is not worth the threads: it runs on devices[0] (TKAMD_SHARD_MIN_KB, read at load)
    std::vector<double> shard_ms;        // last sharded call: wall milliseconds every device's thread was busy (H2D + kernels + collect)
    std::vector<int64_t> shard_bytes;
};

constexpr uint32_t MERGE_THIN_LIMIT = 393216;   // <= 16-byte queue entries up to which the 32-symbol merge launch takes them along (two rounds of its 768 lanes x 256 CUs)
constexpr size_t PHASE_WGS = 1 << 17;           // workgroups the phase table has rows for (per kernel)
constexpr size_t MAX_HOST_WORKSPACES = 4;       // concurrent host-entry calls per handle; further callers wait for a free one

// Host results live in pinned (page-locked) memory so the D2H copies run at PCIe speed; blocks are recycled
// through a small process-wide pool because pinning is expensive.
struct PinnedBlock {
    void* p = nullptr;
    size_t cap = 0;
};
static std::mutex g_pin_mu;
static std::vector<PinnedBlock> g_pin_free;

static PinnedBlock pinned_get(size_t bytes) {
    if (bytes < 64) bytes = 64;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_pin_free.size(); ++i)
            if (g_pin_free[i].cap >= bytes && (best == (size_t)-1 || g_pin_free[i].cap < g_pin_free[best].cap)) best = i;
        if (best != (size_t)-1 && g_pin_free[best].cap <= 2 * bytes + (1u << 20)) {
            PinnedBlock b = g_pin_free[best];
            g_pin_free.erase(g_pin_free.begin() + best);
            return b;
        }
    }
    PinnedBlock b;
    size_t want = bytes + bytes / 8;
    HIP_CHECK(hipHostMalloc(&b.p, want, hipHostMallocPortable));      // (one result buffer is written by every device of a multi-device handle)
    b.cap = want;
    return b;
}
static void pinned_put(PinnedBlock b) {
    if (!b.p || g_forked) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_free.size() >= 16) { (void)hipHostFree(b.p); return; }
    g_pin_free.push_back(b);
}

struct tkamd_batch {
    int64_t n_docs = 0, n_tokens = 0;
    PinnedBlock ids, tok_offsets, offsets, word_ids, pad_counts, type_ids, seq_ids, enc_docs, enc_parts;
    bool has_offsets = false, has_words = false, has_pads = false, has_types = false, has_enc_docs = false, has_enc_parts = false;
    ~tkamd_batch() { pinned_put(ids); pinned_put(tok_offsets); pinned_put(offsets); pinned_put(word_ids); pinned_put(pad_counts); pinned_put(type_ids); pinned_put(seq_ids); pinned_put(enc_docs); pinned_put(enc_parts); }
};

struct tkamd_text {
    int64_t n_docs = 0, n_bytes = 0;
    PinnedBlock bytes, doc_offsets;
    ~tkamd_text() { pinned_put(bytes); pinned_put(doc_offsets); }
};
