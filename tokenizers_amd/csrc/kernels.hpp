// Host-visible declarations of the HIP kernels' launchers (kernels.hip) and the POD structs they take.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>

#include "tables.hpp"

namespace tkamd {

// device-resident tables, passed by value to kernels
struct DevTables {
    const uint16_t* uc1;          // unicode stage 1
    const uint8_t* uc2;           // unicode stage 2
    const uint32_t* byte_id;      // [256]
    const MergeSlot* merges;          // perfect-hash table (one slot per key)
    const uint16_t* merge_disp;       // bucket displacements
    uint32_t merge_mask, merge_seed, merge_bmask;
    uint32_t newid_affine, newid_base;   // new_id == rank + newid_base for every merge (host-verified)
    const uint32_t* char_id;          // BPE over characters: (code point << 2 | affix variant) -> id of the char's one-symbol token or CHAR_NONE (tables.hpp), else null
    uint32_t cb;                      // CB_* flags of that model (0: byte-level BPE, or another model)
    int* err;                         // (per call, in the host's copy) the batch's error bits: the char start reports ERR_UNK_OOV
    uint32_t* probes;                 // (per call, profiling runs only, else null) counter of merge-table probes, (k - 1) + 2 m per word
    uint32_t thin_limit;              // != 0 (per call, in the host's copy): the LDS merge kernels pick the owner of the <= 16-byte queue by its fill (bpe.hip)
    // in-batch claims: set (per call, in the host's copy) when the model kernels publish the claimants' rows themselves (bpe.hip):
    // the rows of the claimed slots, the slot mask, and -- only when offsets are requested -- where the claimant's first byte goes
    void* pub_rows;
    uint32_t* pub_pos;
    uint32_t pub_mask;
    uint32_t word_seed;               // seed of the whole-word hashes (the two-choice table of record lives on the host, tables.hpp)
    const void* shortw;               // the short-word table (tables.hpp): every whole word of <= 16 bytes in 16-byte slots
    const uint32_t* shortw_k3;        // bytes 12..15 of the key in slot i
    const uint8_t* shortw_disp;       // [SHORTW_BUCKETS] eight-bit displacements (the lookup kernel keeps them in LDS)
    uint32_t shortw_mask;
    uint32_t shortw_bmask;            // buckets - 1 of the displacement array (SHORTW_BUCKETS of them; four times that for vocabularies beyond 65,536 words)
    uint32_t ignore_merges;
    uint32_t long_probe_max_len;      // whole-word probes of keys > 16 bytes only up to this length (WordPiece: max_input_chars)
    uint32_t unk_id, has_unk;
    // long (>16 byte) whole-word keys
    const uint8_t* long_blob;
    const uint32_t* long_off;
    const uint32_t* long_id;
    const uint32_t* long_table;
    uint32_t long_mask;
    // WordPiece trie as a pair table: (parent node, byte) -> (child, token id)
    const MergeSlot* trie;
    uint32_t trie_mask, trie_seed;
    uint32_t max_input_chars;
};

// BertNormalizer tables + options, passed by value
struct BnTables {
    const uint16_t* bn1;
    const uint8_t* bn2;
    const MergeSlot* map;             // (cp, kind) -> up to 3 code points, 21 bits each in (rank, new_id)
    uint32_t map_mask, map_seed;
    uint32_t clean, cjk, strip, lower;
};

// one queued pre-token (kernels/results.hip); a queue = NSQ sub-queues of sq_cap entries, one per lookup workgroup, each with its
// own fill counter
struct QItem;
constexpr int NSQ = 768;                                  // (three lookup workgroups per CU x 256 CUs)
constexpr int QCNT_STRIDE = 1;
constexpr int LOOKUP_TILE_BYTES = 16384;                  // text one lookup workgroup takes at a time (kernels/lookup.hip)
constexpr int QCNT_WORDS = 4 * NSQ * QCNT_STRIDE;       // fill counters of the four queues
struct QView {
    QItem* q;
    uint32_t* counts;             // [NSQ] at stride QCNT_STRIDE
    uint32_t sq_cap;              // entries per sub-queue
    uint32_t row_base;            // rows[row_base + position]
};
// ---- word cache (kernels/lookup.hip, output.hip): what BPE::tokenize_with_cache keeps per thread (models/bpe/model.rs:573-586,
// utils/cache.rs) -- pre-token bytes -> its tokens -- kept per workspace in HBM.  A direct-mapped table that only ever fills (the
// reference's cache does not evict either); an entry is a key slot + a result row in the same 16-byte format the merge kernels
// write, so a hit costs the lookup one probe and the compaction nothing extra.
constexpr int WORD_CACHE_BITS = 20;                       // 1 M entries: 32 MB of keys + 16 MB of rows per workspace
constexpr uint32_t ROW_INDEX_LIMIT = 1u << 30;            // tok0 carries 30 bits of row index (results.hip)
constexpr uint32_t CACHE_CLAIMED = 0x80000000u;
struct __attribute__((aligned(32))) CacheKey {
    uint32_t k[4];               // the pre-token's bytes, zero padded to 16
    uint32_t state;              // 0 empty, CACHE_CLAIMED while its writer fills it in, else the length (1..16)
    uint32_t pad[3];
};
struct WordCache {
    CacheKey* keys;              // null: no cache
    void* rows;                  // [1 << WORD_CACHE_BITS] 16-byte rows
    // In-batch word claims (kernels/lookup.hip "claims"): TWO 64-bit words per slot.  Word 0: 0 = free; a word of <= 15 bytes claims with
    // its own bytes 0..6 | length << 56 and leaves bytes 7..14 in word 1 -- the entry IS the key, a later occurrence settles on this one
    // line; a word of 16..32 bytes claims with 0xFF << 56 | length << 32 | first byte of the claimant, whose bytes in the text are the key.
    // Zeroed before every batch.  null: off.  (keys and claims are alternatives.)
    unsigned long long* claims;
    uint32_t claim_mask;         // slots - 1 (a power of two, sized from the batch by the host)
    uint32_t* claim_pos;         // [slots] first byte of the claimant (written with its row; k_token_meta reads it): only when offsets are requested, else null
};

// buffers zeroed by one launch (launch_zero_regions)
struct ZeroRegions {
    void* p[6];
    unsigned long long n16[6];       // 16-byte words
    int n;
    const uint32_t* only_if;         // not null: the launch returns at once when word 0 is zero (the match masks' "may hold bits" flag, capi.cpp scatter_masks);
                                     // word 1 bounds every region: the 16-byte words the last writer could have reached
    void add(void* ptr, size_t bytes) { if (n < 6 && ptr && bytes) { p[n] = ptr; n16[n] = (bytes + 15) / 16; ++n; } }
};
struct QueuePlan {
    QView v[4];                   // pre-tokens of <= 16 bytes, <= 32, <= 64, longer
};

constexpr int META_TILE = 1024;      // pre-tokens a tile of k_token_meta (kernels/output.hip TM_TILE)
// arguments of k_token_meta (offsets / word ids), passed by value
struct MetaArgs {
    const uint8_t* x_text;            // text the pre-tokenizer saw (normalised if a normalizer ran)
    const uint8_t* text;              // the original text
    const uint32_t* pt_start;
    const uint32_t* pt_end;           // null: pt_start[p+1]
    // pt_start null (pre-tokenizers without an end mask, round 6): k_token_meta reads the starts off the start mask itself
    const unsigned long long* startmask;
    const unsigned long long* endmask;    // "Removed" pre-tokenizers: explicit ends (pt_end null then too), else null
    const uint32_t* wprefix;          // starts in front of every 64-byte word
    const uint32_t* tile_w;           // [ceil(P / META_TILE)] the word that holds the start of pre-token k * META_TILE (launch_mask_scan)
    int64_t n_mask_words;
    const int64_t* x_len_dev;         // length of the x text: on the device if it was derived there, else x_len_host
    int64_t x_len_host;
    const int64_t* n_tok;             // total token count (device scalar)
    const uint32_t* pt_tokoff;
    const uint32_t* tmp_end;          // token ends relative to the pre-token start (multi-token pre-tokens)
    const uint8_t* tok_b8;            // per token: the boundary byte in front of it, if its row carried one (kernels/results.hip row_boundary; the
                                      // compaction wrote it next to the ids), else 0 -- token_meta then takes the ends from tmp_end; null without offsets
    const uint32_t* tok0;             // in-batch claims: a pre-token whose tok0 names a claimed slot shares the claimant's tokens, and its token
    const uint32_t* claim_pos;        // ends are the claimant's: tmp_end[claim_pos[slot] + j] (both null when the claims are off)
    const int64_t* n_pretok;
    const uint32_t* doc_pt;
    const uint32_t* chunk_lo;         // the compaction's: chunk_lo[c] = the first document d with doc_pt[d] >= c * chunk (k_doc_first_pretok)
    uint32_t chunk;
    const uint32_t* word_of_doc;      // is_pretokenized: word id of every token of document d (its index in the sequence); else null
    const int64_t* first_tok;         // is_pretokenized with trim_offsets: index of the first token of document d's sequence; else null
    int64_t n_docs;
    const int64_t* x_doc_off;         // document CSR in x space
    const int64_t* doc_off;           // document CSR in the original text
    const uint32_t* norig;            // x byte -> original byte range start of its source char, or null
    const uint32_t* norig_e;          //           ... range end
    const unsigned long long* leadmask;   // char mode: lead-byte bitmask of the original text + its prefix
    const uint32_t* lprefix;
    uint32_t byte_level, trim_offsets, pp_add_prefix_space, want_offsets, char_mode, want_words;
    uint32_t trim_matches_only;       // trim_offsets on a model that is not byte-level: only added-token matches are looked at
    uint32_t snap_chars;              // token edges snap outwards to char boundaries: byte-level tokens, and BPE over characters (its offsets are running
                                      // sums of symbol lengths, which cut chars behind a dropped char or inside byte_fallback's one-byte symbols)
    const uint32_t* char_id;          // BPE over characters WITHOUT an unk_token: chars the vocabulary lacks are dropped and every offset behind them
    uint32_t cb;                      // moves up -- k_token_meta subtracts the dropped bytes in front of every token edge (null / 0: nothing is ever dropped)
    // ... except on a WHOLE-WORD hit of ignore_merges, which reports (0, len) whatever the word holds (bpe/model.rs:559-567).  Set only when
    // both are on: the tok0 words (TOK_ONE: the lookup's hit) and the rows / claimed rows (a long word's hit carries ROW_WHOLE_WORD)
    const uint32_t* ww_tok0;
    const void* ww_rows;
    const void* ww_crows;
    const unsigned long long* matchmask;  // added-token matches (their offsets trim real whitespace chars), or null
    const uint16_t* uc1;
    const uint8_t* uc2;
    uint32_t* offsets;                // [T][2]
    uint32_t* word_ids;               // [T]
    uint8_t* trim1;                   // [T] or null: 1 where process_offsets took exactly one leading space off a token that is not the first of its
                                      // document -- the truncation epilogue gives it back to a token that BECOMES the first of an encoding
                                      // (byte_level.rs:213-222 keeps the one space add_prefix_space stands for on token 0 of whatever it processes)
};

// added-token patterns (AddedVocabulary), passed by value
struct AddedArgs {
    const uint8_t* blob;
    const uint32_t* off;
    const uint32_t* first;     // CSR over the first byte
    const uint32_t* id;
    const uint32_t* flags;     // 1 single_word, 2 lstrip, 4 rstrip, 8 special
    unsigned long long first_set[4];   // bit b: some pattern starts with byte b
    uint32_t n_first;          // distinct first bytes; the first four of them:
    uint32_t first_byte[4];
    uint32_t skip_special;     // Tokenizer.encode_special_tokens: a special token found in the text is left there as text (added_vocabulary.rs:450-453)
};

// arguments of k_add_specials, passed by value
struct SpecialArgs {
    const int64_t* tok_offsets;
    int64_t n_docs;
    const uint32_t* ids;
    const uint32_t* offsets;          // null if not produced
    const uint32_t* word_ids;         // null if not produced
    const uint32_t* prefix;
    const uint32_t* suffix;
    int32_t n_prefix, n_suffix;
    int64_t* tok_offsets2;
    uint32_t* ids2;
    uint32_t* offsets2;
    uint32_t* word_ids2;
    int64_t* n_tok2;
};

// arguments of the truncation / special-token / padding epilogue (k_final_*), passed by value
struct FinalArgs {
    const int64_t* tok_offsets;       // token CSR of the plain encodings
    int64_t n_docs;
    const uint32_t* ids;
    const uint32_t* offsets;          // null if not produced
    const uint32_t* word_ids;         // null if not produced
    const uint8_t* trim1;             // MetaArgs::trim1 or null
    const uint32_t* prefix;           // special ids around every sequence (n_prefix = n_suffix = 0 without add_special_tokens)
    const uint32_t* suffix;
    int32_t n_prefix, n_suffix;
    uint32_t trunc_len;               // tokens of the sequence itself that survive (0xFFFFFFFF: no truncation)
    uint32_t trunc_left;              // keep the end instead of the beginning
    uint32_t trunc_needs_pair;        // strategy OnlySecond: a single sequence that must be cut is an error
    uint32_t trunc_stride;            // tokens two neighbouring windows of a truncation share (shapes the overflowing encodings; must be < trunc_len)
    uint32_t pad_on, pad_fixed, pad_length, pad_multiple, pad_left, pad_id;
    // overflowing encodings (TKAMD_WANT_OVERFLOW; overflow_core.hpp): document d leaves ovf_parts[d] encodings, numbered from
    // enc_base[d]; encoding e is tokens [enc_start[e], enc_start[e] + enc_cnt[e]) of document enc_doc[e].  With these set, n_docs of
    // k_final_fin / k_final_down / k_finalize counts ENCODINGS and len1 .. pad_count are per encoding.  All null: one encoding per document.
    uint32_t* ovf_parts;              // [n_docs + 1]
    int64_t* enc_base;                // [n_docs + 1]
    uint32_t* enc_doc;
    uint32_t* enc_start;
    uint32_t* enc_cnt;
    uint32_t* len1;                   // [n_docs] tokens after truncation + specials
    uint32_t* fin;                    // [n_docs] tokens after padding
    uint32_t* target;                 // device scalar: longest len1 of the batch
    uint32_t* bsum;
    int64_t* tok_offsets2;
    uint32_t* ids2;
    uint32_t* offsets2;
    uint32_t* word_ids2;
    uint32_t* pad_count;              // [n_docs] padding tokens of each encoding (null without padding)
    int64_t* n_tok2;
    int* err;
    // a single template with type ids (TemplateProcessing, template.rs:554-575): every output token's type id and sequence id
    // (0; 2 special; 3 padding) are written like the pair epilogue writes them.  Null otherwise: single sequences are all type 0.
    uint8_t* type_ids2;
    uint8_t* seq_ids2;
    const uint8_t* prefix_ty;         // [n_prefix] / [n_suffix] type ids of the special tokens
    const uint8_t* suffix_ty;
    uint32_t seq_ty;                  // type id of the sequence's own tokens (an overflowing window keeps 0)
    uint32_t pad_type_id;
};

// arguments of the PAIR epilogue (k_pair_*): documents 2i / 2i+1 are sequence A / B of encoding i (tokenizer/mod.rs:871-889)
struct PairArgs {
    const int64_t* tok_offsets;       // token CSR over the 2 * n_pairs documents
    int64_t n_pairs;
    const uint32_t* ids;
    const uint32_t* offsets;          // null if not produced
    const uint32_t* word_ids;         // null if not produced
    const uint8_t* trim1;             // MetaArgs::trim1 or null
    const uint32_t* tpl;              // [n_tpl][3] kind (0 A, 1 B, 2 special), id, type id
    int32_t n_tpl;
    uint32_t n_special;               // special tokens of the template (taken off max_length)
    // A Vec<EncodeInput> that mixes EncodeInput::Single and ::Dual (tokenizer/mod.rs:225-290, 1337-1356): input i is the sequences
    // [inp_off[i], inp_off[i + 1]) -- one (Single) or two (Dual, A then B) -- instead of 2i / 2i + 1; a Single is laid out by the single
    // template tpl1 (n_special1 special tokens), cut like truncate_encodings cuts an encoding without a pair (utils/truncation.rs:70-160)
    // and leaves the windows Encoding::truncate leaves a single sequence.  Null: every input is a pair.
    const int64_t* inp_off;           // [n_pairs + 1] or null
    const uint32_t* tpl1;             // [n_tpl1][3] the single template (kind 0 and 2 only)
    int32_t n_tpl1;
    uint32_t n_special1;
    uint32_t trunc_on, trunc_max, trunc_left, trunc_strategy, trunc_stride;
    uint32_t pad_on, pad_fixed, pad_length, pad_multiple, pad_left, pad_id, pad_type_id;
    uint32_t* keep;                   // [2 * n_pairs] tokens of A / B that survive the truncation
    // overflowing encodings of pairs (TKAMD_WANT_OVERFLOW): sequence A leaves pa windows (its truncated self + its overflowing pieces),
    // B leaves pb; Encoding::merge_with (tokenizer/encoding.rs:408-432) combines every window of the sequence that comes FIRST in the
    // template with every window of the other one.  Pair i leaves ovf_parts[i] = pa * pb encodings numbered from enc_base[i], in the
    // reference's order (k_pair_ranges); encoding e is windows enc_idx[2e] / [2e + 1] of A / B of pair enc_doc[e], token ranges
    // enc_win[4e ..] = {first token of A's window, count, first token of B's window, count}.  With these set, n_pairs of
    // k_pair_finalize counts ENCODINGS and len1 .. pad_count are per encoding.  All null: one encoding per pair.
    uint32_t first_is_b;              // the template names sequence B before sequence A
    uint32_t ovf_ty_tpl;              // overflowing windows take the template's type id too (RobertaProcessing with special tokens: everything 0)
    uint32_t* ovf_parts;              // [n_pairs + 1]
    int64_t* enc_base;                // [n_pairs + 1]
    uint32_t* enc_doc;
    uint32_t* enc_idx;
    uint32_t* enc_win;
    uint32_t* len1;                   // [n_pairs] tokens of the pair encoding before padding
    uint32_t* fin;
    uint32_t* target;
    uint32_t* bsum;
    int64_t* tok_offsets2;            // [n_pairs + 1]
    uint32_t* ids2;
    uint32_t* offsets2;
    uint32_t* word_ids2;
    uint8_t* type_ids2;               // per token
    uint8_t* seq_ids2;                // per token: 0 sequence A, 1 sequence B, 2 special token, 3 padding
    uint32_t* pad_count;
    int64_t* n_tok2;
    int* err;
};

// error bits accumulated in a device int during a batch
enum : int {
    ERR_BAD_OFFSETS = 1,          // doc_offsets not a valid CSR over [0, n_bytes]
    ERR_PRETOKEN_TOO_LONG = 2,    // a pre-token exceeds LONG_PT_MAX symbols
    ERR_ADDED_SPLIT = 4,          // an lstrip added token whose start was pushed past its own end by the previous match (the reference panics: "AddedVocabulary bad split")
    ERR_NON_ASCII_NORM = 8,       // BertNormalizer on non-ASCII text (full-Unicode path not built yet)
    ERR_MISSING_UNK = 16,
    ERR_INTERNAL = 32,            // an internal invariant was violated (bug guard)
    ERR_QUEUE_FULL = 64,
    ERR_TRUNC_SECOND = 128,       // truncation strategy OnlySecond on a single sequence that has to be cut (TruncationError::SecondSequenceNotProvided)
    ERR_TOO_MANY_TOKENS = 256,
    ERR_TRUNC_SHORT = 512,        // OnlyFirst / OnlySecond: the sequence to cut is not longer than what must go (TruncationError::SequenceTooShort)    // the padded batch has more than 2^32 tokens
    ERR_TRUNC_STRIDE = 1024,      // a sequence has to be cut to max_len tokens and stride >= max_len (the assert of Encoding::truncate, encoding.rs:319)
    ERR_INPUT_KIND = 8192,        // an input of a mixed batch that is neither one sequence nor two (tkamd_encode_batch_mixed)
    ERR_UNK_OOV = 4096,           // BPE: a char the vocabulary lacks, and the unk_token that should stand for it is not in the vocabulary either (Error::UnkTokenOutOfVocabulary, bpe/model.rs:528-533)
    NOTE_REORDER_SEEN = 2048,     // not an error: the normalizer met a character NFD's canonical ordering could move (k_bn_reorder_fix then looks at its neighbours)
    NOTE_ADDED_SEEN = 16384,      // not an error: a batch that was run as if the text held no added token (run_pipeline's speculation) met the content of one: the host runs
                                  // it again with the matching passes (finish_batch) and stops speculating for a while
    NOTE_BITS = NOTE_REORDER_SEEN | NOTE_ADDED_SEEN,
    ERR_QUEUE_FULL_PAD = 0,          // a work queue / the row area was too small for this batch: the host grows it and runs the batch again
};

// indices into the per-batch device counter array
enum : int { CNT_LIST16 = 0, CNT_LIST64 = 1, CNT_LISTL = 2, CNT_LIST32 = 3, CNT_SLOW_DOCS = 4, CNT_CLAIM_CANDS = 5, CNT_CLAIM_SHARED = 6, CNT_LISTH = 7, CNT_MATCH_DOCS = 8,
              CNT_MATCHES = 9, CNT_MATCH_DOCS2 = 10, CNT_CLAIM_GAVE_UP = 11, CNT_MERGE_PROBES = 12, CNT_COUNT = 13 };
// (CNT_CLAIM_*: in-batch claims -- candidates the lookup looked at, how many were another pre-token's word, workgroups that stopped claiming)

constexpr uint32_t MATCH_LEN_ORIG = 0x80000000u;    // added-token match list, word 3: the length counts bytes of the ORIGINAL text
constexpr int TEXT_PAD = 64;        // = TKAMD_TEXT_PAD (include/tokenizers_amd.h): readable bytes past the end of every text buffer
constexpr int LONG_PT_MAX = 8192;  // symbols per pre-token on the workgroup path (LDS resident); longer ones use the global-scratch kernel

void launch_mark_doc_starts(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes,
                            unsigned long long* docmask, int* err);
// validates the caller's CSR (ERR_BAD_OFFSETS) and writes the copy every later kernel reads (a trivially valid one if it is malformed)
void launch_validate_csr(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes, int* err, int64_t* san);
void launch_pretok_gpt2(hipStream_t st, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                        const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* leadmask = nullptr);      // leadmask: the lead-byte mask of the same text too
void launch_mask_scan(hipStream_t st, const unsigned long long* mask, int64_t n_words, uint32_t* bsum, uint32_t* wprefix,
                      int64_t* total, const int64_t* len_dev = nullptr, uint32_t* tile_w = nullptr);      // len_dev: only the words of a text of that (device-side) length
void launch_emit_pretok(hipStream_t st, const unsigned long long* startmask, const uint32_t* wprefix, int64_t n_bytes,
                        const int64_t* len_dev, const int64_t* n_pretok, uint32_t* pt_start);
void launch_prefix_space(hipStream_t st, const uint8_t* text, const int64_t* seg_off, int64_t n_bound, const int64_t* n_dev, const unsigned long long* matchmask,
                         uint32_t* need, uint32_t* bsum, int64_t* xseg_off, int64_t* x_len, uint8_t* xtext, uint32_t* nos, uint32_t* noe, int grid);
void launch_doc_first_pretok(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes,
                             const unsigned long long* startmask, const uint32_t* wprefix, const int64_t* n_pretok, uint32_t* doc_pt, uint32_t* chunk_lo,
                             const int* err = nullptr, int64_t* san = nullptr);      // san: doc_off is the caller's array, the kernel also writes its validated copy
// whole-word lookup straight from the start (/ end) bitmasks: settles or queues every pre-token (kernels/lookup.hip)
void launch_lookup(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev,
                   const unsigned long long* startmask, const unsigned long long* endmask, const uint32_t* wprefix, uint32_t* tok0,
                   const QueuePlan& plan, int* err, const unsigned long long* matchmask, const void* hot, const WordCache& wc,
                   uint32_t no_hits, uint32_t miss_is_unk, void* phases = nullptr, uint32_t* counters = nullptr);      // phases: [grid][8] u64, diagnostic instantiation (lookup.hip)
// group: 64 = a wavefront per pre-token; 1 / 2 = one lane per pre-token, Word in registers (16 / 32 symbols: vocabularies whose new ids are not rank + c);
// 5 / 6 = one lane per pre-token, keys in LDS (16 / 32 symbols; needs new_id = rank + c)
// also (group 6 only): a second queue for the same launch
void launch_bpe_merge(hipStream_t st, int grid, int group, const DevTables& t, const uint8_t* text, const QView& v, void* rows,
                      uint32_t* tmp_ids, uint32_t* tmp_end, const QView* also = nullptr);
// the normaliser's count arrays share one buffer: n_bytes + 64 per-byte counts (written for the lanes that are not plain only), then
// one byte per 16-byte lane (bert_norm_core.hpp BnOlen); bn_olen_bytes(n) is what the buffer must hold
inline size_t bn_olen_bytes(int64_t n_bytes) { return (((size_t)n_bytes + 64 + 15) & ~(size_t)15) + ((size_t)n_bytes >> 4) + 64; }
inline uint8_t* bn_ltot_of(uint8_t* olen, int64_t n_bytes) { return olen + (((size_t)n_bytes + 64 + 15) & ~(size_t)15); }
void launch_mark_doc_starts_n(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes, const int64_t* len_dev,
                              unsigned long long* docmask, int* err);
void launch_pretok_local(hipStream_t st, int kind, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                         const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* endmask, bool len_bound = false);
void launch_emit_pretok_end(hipStream_t st, const unsigned long long* startmask, const unsigned long long* endmask,
                            const uint32_t* wprefix, int64_t n_bytes, const int64_t* len_dev, uint32_t* pt_end);
void launch_bert_normalize(hipStream_t st, const BnTables& bt, const uint8_t* text, int64_t n_bytes, const int64_t* doc_off, int64_t n_docs,
                           const unsigned long long* verbatim, uint8_t* olen, uint32_t* wsum, uint32_t* bsum, uint32_t* wbase, int64_t* x_len, uint8_t* ntext,
                           uint32_t* nos, uint32_t* noe, int64_t* ndoc_off, int* err);
// whole-word vocabulary hits of QUEUED pre-tokens longer than 16 bytes (ignore_merges, WordLevel): a hit becomes the result row and
// the queue entry is retired (length 0) so that the model kernels skip it
void launch_long_vocab3(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v1, const QView& v2, const QView& v3, void* rows, uint32_t miss_is_unk,
                        int* err, const WordCache& wc);
void launch_wordpiece_all(hipStream_t st, int grid_short, int grid_long, const DevTables& t, const uint8_t* text, const QueuePlan& plan, void* rows, uint32_t* tmp_ids,
                          uint32_t* tmp_end, int* err);      // the <= 16-byte queue and the three longer ones in one launch
// (rule: the member of the tiktoken family, tables.hpp SplitRule; ucc1 / ucc2: the case classes its case-split letter alternatives read, else null)
void launch_pretok_llama3(hipStream_t st, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                          const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* slowmask,
                          const int64_t* doc_off, int64_t n_docs, const int64_t* n_docs_dev, uint32_t* slow_docs, uint32_t* n_slow_docs,
                          SplitRule rule, const uint16_t* ucc1, const uint8_t* ucc2, unsigned long long* tileflags, unsigned long long* leadmask = nullptr);      // leadmask: as launch_pretok_gpt2's (the bit-parallel members only)
// (tileflags: L3_TILEFLAG_WORDS(n_bytes) zeroed 64-bit words -- a bit per 2048 bytes of text, the tile kernel's work list)
inline size_t l3_tileflag_words(int64_t n_bytes) { return (size_t)((n_bytes + 1) / 2048 + 1) / 64 + 2; }
void launch_leadmask(hipStream_t st, const uint8_t* text, int64_t n_bytes, unsigned long long* leadmask);
void launch_seq_regroup(hipStream_t st, const int64_t* seq_off, int64_t n_seqs, int64_t n_words, const int64_t* word_tok_off, int64_t* seq_tok_off, uint32_t* widx,
                        int64_t* first_tok);
void launch_token_meta(hipStream_t st, int grid, const MetaArgs& a);
void launch_add_specials(hipStream_t st, int grid, const SpecialArgs& a);
// truncation + specials + padding: lengths (and the batch maximum), then the new CSR (*n_tok2 = its total), then the copy
void launch_add_i64(hipStream_t st, int64_t* data, int64_t n, int64_t delta);
void launch_add_u32(hipStream_t st, uint32_t* data, int64_t n, uint32_t delta);
void launch_final_lens(hipStream_t st, const FinalArgs& a);
// overflowing encodings: encodings per document + their numbering (*n_enc = how many there are), then -- with a.n_docs still the
// number of DOCUMENTS -- every encoding's token range, its length with the specials, and the batch maximum over the truncated
// encodings themselves (what launch_final_lens computes when nothing overflows)
void launch_overflow_count(hipStream_t st, const FinalArgs& a, int64_t* n_enc);
void launch_overflow_ranges(hipStream_t st, const FinalArgs& a);
void launch_final_offsets(hipStream_t st, const FinalArgs& a);
void launch_finalize(hipStream_t st, int grid, const FinalArgs& a);
void launch_pair_lens(hipStream_t st, const PairArgs& a);
// overflowing encodings of pairs: launch_pair_lens (with ovf_parts set) counts them; this numbers them (*n_enc = how many), then
// launch_pair_ranges -- a.n_pairs still the number of PAIRS -- writes every encoding's windows, its length and the batch maximum
void launch_pair_overflow_scan(hipStream_t st, const PairArgs& a, int64_t* n_enc);
void launch_pair_ranges(hipStream_t st, const PairArgs& a);
void launch_pair_finalize(hipStream_t st, int grid, const PairArgs& a);
// AddedVocabulary (kernels/documents.hip): one matching pass over a sentence CSR, list -> masks, list coordinate changes
void launch_added_match(hipStream_t st, const AddedArgs& a, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const int64_t* seg_off, int64_t n_segs,
                        const int64_t* n_segs_dev, const unsigned long long* skipmask, const uint16_t* uc1, const uint8_t* uc2, unsigned long long* candmask,
                        uint32_t* sents, uint32_t* n_sents, uint32_t* match_list, uint32_t* n_match, uint32_t cap, uint32_t len_flag, int* err);
// the speculative form of a matching pass: does the content of any pattern occur at all?  (NOTE_ADDED_SEEN into *note)
void launch_added_detect(hipStream_t st, const AddedArgs& a, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, int* note);
void launch_scatter_matches(hipStream_t st, const uint32_t* list, const uint32_t* n_list, int64_t n_bytes, const int64_t* len_dev, unsigned long long* matchmask,
                            unsigned long long* spanmask, unsigned long long* stopmask, unsigned long long* hardmask, uint32_t* tmp_end, uint32_t* dirty);
void launch_mask_or2(hipStream_t st, unsigned long long* dst, const unsigned long long* a, const unsigned long long* b, int64_t n_words);
void launch_emit_boundaries(hipStream_t st, const unsigned long long* mask, const uint32_t* wprefix, int64_t n_bytes, const int64_t* len_dev, const int64_t* total, int64_t* out);
void launch_translate_matches_norm(hipStream_t st, uint32_t* list, const uint32_t* n_list, const uint8_t* olen, const uint32_t* wbase, int64_t n_bytes, const int64_t* x_len);
void launch_translate_matches_prefix(hipStream_t st, uint32_t* list, const uint32_t* n_list, const unsigned long long* bmask, const uint32_t* wprefix, int64_t n_bytes,
                                     const int64_t* len_dev, const int64_t* total, const int64_t* xseg_off);
void launch_prefix_doc_csr(hipStream_t st, const int64_t* doc_off, int64_t n_docs, const unsigned long long* bmask, const uint32_t* wprefix, int64_t n_bytes,
                           const int64_t* len_dev, const int64_t* total, const int64_t* xseg_off, int64_t* xdoc_off);
void launch_mask_or(hipStream_t st, unsigned long long* dst, const unsigned long long* src, int64_t n_words, const uint32_t* n_list);
void launch_apply_matches(hipStream_t st, unsigned long long* startmask, unsigned long long* endmask, const unsigned long long* matchmask,
                          const unsigned long long* spanmask, const unsigned long long* stopmask, int64_t n_words, const uint32_t* n_list);
void launch_apply_match_ids(hipStream_t st, const uint32_t* match_list, const uint32_t* n_match, const unsigned long long* startmask,
                            const uint32_t* wprefix, uint32_t* tok0);
// decode_batch: phase 1 (out_bytes_or_null == nullptr) computes lengths / positions / document offsets / *total,
// phase 2 gathers the bytes; firstmask is null when no id has a position-dependent form
void launch_decode(hipStream_t st, const uint32_t* ids, const int64_t* tok_off, int64_t n_docs, int64_t n_tok, const void* entry, uint32_t n_ids,
                   const uint8_t* blob, uint32_t skip_special, uint32_t* firstmask, uint32_t* len, uint32_t* bsum, uint32_t* pos, int64_t* total,
                   int64_t* out_off, uint8_t* out_bytes_or_null, uint32_t from_end = 0, uint32_t* badmask = nullptr, uint32_t* dupmask = nullptr);      // from_end: the position mask marks the LAST kept token (BPEDecoder); badmask: ByteFallback
int prepare_long_kernel();
void launch_bpe_merge_long(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v, void* rows,
                           uint32_t* tmp_ids, uint32_t* tmp_end, uint32_t* list_huge, uint32_t* n_huge,
                           uint32_t* scratch, unsigned long long scratch_words, unsigned long long* scratch_used, int* err);
// the workgroup-per-pre-token kernel alone, on any queue (BPE over characters: the 33..64-byte class, and every class when the LDS
// kernels cannot run)
void launch_bpe_merge_long_only(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v, void* rows, uint32_t* tmp_ids, uint32_t* tmp_end,
                                uint32_t* list_huge, uint32_t* n_huge);
// single-pass compaction; `state` (8 bytes per chunk of COMPACT_CHUNK pre-tokens) must be zero on entry; pt_tokoff may be null.
// Any grid makes progress (a look-back that runs out of patience computes the missing totals itself: kernels/output.hip);
// compact_grid(n_cu) -- what is resident at once -- is the one that never has to.
void launch_word_cache_insert(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const QView& v, const void* rows, const WordCache& wc);
void launch_compact(hipStream_t st, int grid, const uint32_t* tok0, const void* rows, const void* cache_rows, const uint32_t* tmp_ids, const int64_t* n_pretok,
                    unsigned long long* state, int64_t* n_tok, uint32_t* pt_tokoff, uint32_t* ids, const uint32_t* chunk_lo, const uint32_t* doc_pt,
                    int64_t n_docs, int64_t* tok_offsets, void* phases = nullptr, uint8_t* tok_b8 = nullptr);
int compact_grid(int n_cu);
void launch_zero_regions(hipStream_t st, int grid, const ZeroRegions& z);
void launch_zero_tail(hipStream_t st, uint8_t* p, const int64_t* len_dev, int n, unsigned long long* mask = nullptr, int64_t mask_words = 0, int grid = 1);      // n <= 256 zero bytes at p[*len_dev ..]; mask: its words below (*len_dev >> 6) + 3 zeroed too
constexpr int CP_ITEMS_PER_LANE = 4;                // pre-tokens per lane of k_compact (0.145 ms on C2 against 0.187 with eight; two: deleted in round 6, HISTORY.md)
constexpr int COMPACT_CHUNK = 256 * CP_ITEMS_PER_LANE;   // pre-tokens per compaction chunk (and per tile of k_token_meta)
// test hooks: environment switches the TESTS alone use (a compaction grid no launch would pick, a look-back without patience, ...), read
// only when TKAMD_TEST_HOOKS=1 is set as well (capi.cpp)
const char* test_hook(const char* name);

}  // namespace tkamd
