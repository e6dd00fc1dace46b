// Host-side model of a tokenizer.json restricted to the encode_batch hot path: parses the
// reference's serialization (tokenizer/serialization.rs:104-171, models/bpe/serialization.rs:72-152,
// models/wordpiece/serialization.rs, models/wordlevel/serialization.rs) and lays the model out as
// flat lookup tables that are copied verbatim into HBM.  No tokenisation happens on the host.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "tables.hpp"

namespace tkamd {

// thrown for components outside the hot path -> TKAMD_ERR_UNSUPPORTED
struct Unsupported : std::runtime_error {
    using std::runtime_error::runtime_error;
};
// thrown for malformed input -> TKAMD_ERR_INVALID
struct Invalid : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Byte trie for WordPiece longest-match (models/wordpiece/mod.rs:224-283), flattened into the same
// 2-choice cuckoo layout as the merge table: key (parent node, byte) -> (child node, token id ending
// at the child or 0xFFFFFFFF).  Node 0 = root for word-initial pieces, node 1 = root for continuation
// pieces (vocab keys starting with continuing_subword_prefix, stored without the prefix).
struct ByteTrie {
    std::vector<MergeSlot> table;   // a = parent, b = byte, rank = child, new_id = token id
    uint32_t mask = 0, seed = 0;
    uint32_t n_nodes = 0;
};

struct AddedToken {
    std::string content;
    uint32_t id = 0;
    bool special = false, single_word = false, lstrip = false, rstrip = false, normalized = false;
};

// decode_batch (tokenizer/mod.rs:935-953): what the `decoder` section does to the token strings
enum DecoderKind { DEC_JOIN_SPACE = 0 /* decoder: null -> tokens.join(" ") */, DEC_BYTELEVEL = 1, DEC_WORDPIECE = 2, DEC_UNSUPPORTED = 3,
                   DEC_BPE = 4 /* BPEDecoder: the end-of-word suffix becomes a space, nothing on the last token (decoders/bpe.rs:26-39) */,
                   DEC_BYTE_FALLBACK = 5 /* ByteFallback, alone or Sequence[ByteFallback, Fuse] (decoders/byte_fallback.rs:27-67, fuse.rs:24-29) */,
                   DEC_FUSE = 6 /* Fuse: tokens.join("") */,
                   DEC_CHAIN = 7 /* Replace / Strip, alone or as Sequence[Replace*, Strip*, ByteFallback?, Fuse?, Strip(c, <= 1, 0)?] (replace.rs:88-106, strip.rs:27-60, sequence.rs:26-33) */,
                   DEC_CTC = 8 /* decoders/ctc.rs:45-63 */ };

struct HostModel {
    ModelKind model = MODEL_NONE;
    PretokKind pretok = PT_NONE;
    NormKind norm = NORM_NONE;

    // ByteLevel options (pre_tokenizers/byte_level.rs:57-70)
    bool byte_level = false;      // model strings are in the GPT-2 byte alphabet
    bool add_prefix_space = false;
    bool trim_offsets = false;    // ByteLevel *post-processor* option (byte_level.rs:175-234)
    bool pp_add_prefix_space = true;  // the post-processor's own add_prefix_space (process_offsets argument)
    // BertNormalizer options (normalizers/bert.rs:62-90)
    bool bn_clean_text = true, bn_handle_chinese = true, bn_strip_accents = true, bn_lowercase = true;

    // model options
    bool ignore_merges = false;
    bool has_unk = false;
    uint32_t unk_id = 0;
    std::string unk_token;
    std::string cont_prefix;            // WordPiece continuing_subword_prefix ("##")
    uint32_t max_input_chars = 100;     // WordPiece max_input_chars_per_word

    // post-processor layout for a single sequence (processors/{bert,roberta,template,sequence}.rs)
    std::vector<uint32_t> pp_prefix, pp_suffix;   // special ids before / after sequence A
    std::string pp_unsupported;                    // non-empty: why add_special_tokens cannot be honoured
    // the same post-processor for a PAIR of sequences (processors/bert.rs:51-150, roberta.rs, template.rs `pair`): the pieces in
    // order -- kind 0 = sequence A, 1 = sequence B, 2 = one special token id -- each with the type id its tokens get
    struct TplPiece { uint32_t kind, id, type_id; };
    std::vector<TplPiece> pp_pair;                 // empty: no adding post-processor (a pair is then A followed by B, type ids 0 / 1)
    std::string pp_pair_unsupported;
    bool pp_roberta = false;           // RobertaProcessing: with special tokens every type id is 0, overflowing windows included
    // TemplateProcessing's single template may give its pieces type ids (template.rs:554-575; the sequence's applies with or without
    // special tokens): single sequences then carry a type id array like pairs do
    std::vector<uint8_t> pp_prefix_ty, pp_suffix_ty;   // aligned with pp_prefix / pp_suffix
    uint32_t pp_seq_ty = 0;
    bool pp_single_typed = false;      // any of the three is non-zero
    bool pp_single_refused = false;    // a single template outside the path (pp_unsupported says why): it shapes a single sequence with or without special tokens
    // the layout of a pair when NO special tokens are added: A : 0, B : 1 by default (bert.rs:56-58 returns the encodings as they are);
    // RobertaProcessing zeroes every type id first (roberta.rs); TemplateProcessing still applies its order and type ids
    std::vector<TplPiece> pp_pair_plain = {{0, 0, 0}, {1, 0, 1}};
    // the single layout as pieces, for the inputs of a mixed batch that are single sequences (tkamd_encode_batch_mixed: the pair kernels
    // lay out both kinds): pp_prefix | A | pp_suffix with their type ids, and A alone when no special tokens are added
    std::vector<TplPiece> pp_single, pp_single_plain;

    // truncation / padding of the finished encodings (utils/truncation.rs:70-160, utils/padding.rs:50-85; tokenizer/mod.rs:1265-1317)
    bool trunc_on = false;
    uint32_t trunc_max_length = 0;      // before the post-processor's special tokens are subtracted
    bool trunc_left = false;            // TruncationDirection::Left keeps the END of the sequence
    int trunc_strategy = 0;             // 0 LongestFirst, 1 OnlyFirst, 2 OnlySecond (a single sequence then fails when it must be cut)
    uint32_t trunc_stride = 0;          // only shapes the overflowing pieces, which this path does not materialise
    bool pad_on = false;
    bool pad_fixed = false;             // PaddingStrategy::Fixed(pad_length) vs BatchLongest
    uint32_t pad_length = 0;
    bool pad_left = false;
    uint32_t pad_multiple = 0;          // pad_to_multiple_of (0 = off)
    uint32_t pad_id = 0, pad_type_id = 0;
    std::string pad_token;

    uint32_t vocab_size = 0;            // number of vocab entries
    uint32_t n_merges = 0;
    std::vector<AddedToken> added_tokens;
    // AddedVocabulary patterns (added_vocabulary.rs:370-414), one set per matching pass of extract_and_normalize (:523-564):
    // [0] normalized = false, matched by their content on the raw text; [1] normalized = true, matched by the NORMALIZED form of
    // their content on the normalized pieces between the matches of pass 1.  Sorted, CSR over the first byte.
    struct PatternSet {
        std::vector<uint8_t> blob;
        std::vector<uint32_t> off;      // [n_patterns+1]
        std::vector<uint32_t> first;    // [257] CSR over the first byte
        std::vector<uint32_t> id;       // [n_patterns] token id
        std::vector<uint32_t> flags;    // [n_patterns] 1 single_word, 2 lstrip, 4 rstrip
        size_t size() const { return id.size(); }
    };
    PatternSet at[2];
    // BertNormalizer::normalize (normalizers/bert.rs:92-138) of one code point / a string from the generated tables (host side:
    // pattern normalization at load, tkamd_probe_bert_norm); *refused = a character whose NFD reordering is context dependent
    int bn_expand_cp(uint32_t cp, uint32_t* out, int* refused) const;
    std::string bert_normalize(const std::string& s, bool* refused) const;

    // ---- tables copied to the device ----
    uint32_t byte_id[256];              // byte -> id of its one-symbol token (BPE byte-level); BPE over chars with byte_fallback: id of "<0xXX>"
    // BPE over CHARACTERS (no ByteLevel pre-tokenizer): the options of BPE::merge_word (bpe/model.rs:465-550)
    bool char_bpe = false;
    bool unk_configured = false;        // an unk_token is set (has_unk: ... and the vocabulary holds it; else UnkTokenOutOfVocabulary when first needed)
    bool fuse_unk = false, byte_fallback = false;
    std::string bpe_prefix, bpe_suffix; // continuing_subword_prefix / end_of_word_suffix ("" = none)
    // char -> id of its one-symbol token, indexed by (code point << 2 | variant), variant bit 0: the prefix is glued on (not the word's first
    // char), bit 1: the suffix (its last char); CHAR_NONE: the vocabulary has no such entry.  Direct indexed: 0x110000 x 4 words (17 MB)
    std::vector<uint32_t> char_id;
    std::vector<MergeSlot> merge_table; // perfect hash (hash-and-displace), size = merge_mask+1 (power of two)
    std::vector<uint16_t> merge_disp;   // displacement per bucket, size = merge_bmask+1
    uint32_t merge_mask = 0, merge_seed = 0, merge_bmask = 0;
    bool merge_newid_affine = false;    // new_id == rank + merge_newid_base for every merge
    uint32_t merge_newid_base = 0;
    std::vector<WordSlot> word_table;   // two-choice table (tables.hpp word_slot_a / word_slot_b), size = word_mask+1
    uint32_t word_mask = 0, word_seed = 0;
    uint32_t n_words = 0;               // keys stored in word_table (<= 16 bytes)
    // vocab entries longer than 16 raw bytes, needed for ignore_merges / WordLevel whole-word probes:
    // sorted blob for a device-side hash (long_table: open addressing over (hash -> entry index))
    std::vector<uint8_t> long_blob;     // concatenated raw keys
    std::vector<uint32_t> long_off;     // [n_long+1]
    std::vector<uint32_t> long_id;      // [n_long]
    std::vector<uint32_t> long_table;   // open addressing: entry index+1, 0 = empty; size long_mask+1
    uint32_t long_mask = 0;
    ByteTrie trie;                      // WordPiece

    // BertNormalizer per-code-point data (bert_norm_tables.inc): 2-stage flag table + cuckoo map (cp, kind) -> 3 x 21-bit cps
    std::vector<uint16_t> bn_stage1;
    std::vector<uint8_t> bn_stage2;
    std::vector<MergeSlot> bn_map;
    uint32_t bn_mask = 0, bn_seed = 0;

    // ---- decode_batch tables: every decoder on the path is a per-token string function of (id, first kept token of the
    // document?), so decoding is a gather.  dec_entry[id] = {offset of the FIRST-position form, its length | flags,
    // offset of the other-position form, its length}; dec_blob holds the byte strings.
    DecoderKind decoder = DEC_UNSUPPORTED;
    std::string dec_unsupported;        // why decode_batch is refused (DEC_UNSUPPORTED)
    std::vector<uint32_t> dec_entry;    // [n_ids * 4]
    std::vector<uint8_t> dec_blob;
    bool dec_position_dependent = false; // first-position form differs from the other one for some id
    bool dec_special_is_last = false;    // BPEDecoder: the position with a form of its own is the LAST kept token of a sequence, not the first
    bool dec_dedup = false;              // CTC: a kept token equal to the kept token in front of it is dropped first
    bool dec_has_bytes = false;          // ByteFallback: some id is a <0xXX> token (runs of them are validated as UTF-8 on the device)

    std::vector<uint16_t> uc_stage1;    // [UC_STAGE1_LEN]
    std::vector<uint8_t> uc_stage2;     // [n_blocks*256]
    SplitRule split_rule = SPLIT_RULE_LLAMA3;   // PT_LLAMA3: which member of the tiktoken family the Split pattern is (tables.hpp)
    std::vector<uint16_t> ucc_stage1;   // case classes of the case-split letter alternatives (split_rule.letters == 2), same two stages; else empty
    std::vector<uint8_t> ucc_stage2;

    // raw-byte form of every vocab entry that is expressible in raw bytes (byte-level: inverse of
    // the GPT-2 alphabet; others: the UTF-8 string itself) -- used to build the tables above and by
    // the load-time merge-stability check.
    std::vector<std::string> raw_tokens;   // indexed by position in `ids`
    std::vector<uint32_t> raw_ids;

    static HostModel from_json(const char* json, size_t len);
};

uint32_t fnv1a(const uint8_t* p, size_t n);
// two-choice placement of keys with hashes h1 (tables.hpp word_slot_a / word_slot_b): tenant[slot] = key index or 0xFFFFFFFF; false: unlucky seed
bool cuckoo_place(const std::vector<uint32_t>& h1, uint32_t mask, std::vector<uint32_t>* tenant);

}  // namespace tkamd
