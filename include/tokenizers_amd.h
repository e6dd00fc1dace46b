/*
 * tokenizers_amd.h -- C ABI of the MI355X-native encode_batch path.
 *
 * This is the drop-in boundary for ONE hot path of huggingface/tokenizers:
 *   TokenizerImpl::encode_batch / encode_batch_char_offsets / encode_batch_fast
 *   (reference: tokenizers/src/tokenizer/mod.rs:1337-1401), i.e. per document
 *   extract_and_normalize -> PreTokenizer::pre_tokenize -> Model::tokenize ->
 *   PreTokenizedString::into_encoding (mod.rs:762-805, pre_tokenizer.rs:198-263).
 *
 * The reference has no FFI for this path (its plugin API is Rust traits,
 * tokenizer/mod.rs:56-207), so the entry points below are what a Rust
 * `extern "C"` block / a ctypes stub would bind.  Plain
 * pointers and sizes only; no C++ / torch types; nothing throws across the ABI.
 *
 * Data model (replaces Vec<EncodeInput> in, Vec<Encoding> out):
 *   in : one contiguous UTF-8 buffer `text` + CSR `doc_offsets[n_docs+1]`
 *        (document d = text[doc_offsets[d] .. doc_offsets[d+1]) ).
 *   out: CSR token arrays: ids[n_tokens] (u32) + tok_offsets[n_docs+1] (i64),
 *        and, when requested, per-token (start,end) offsets relative to the
 *        document and per-token word ids (= pre-token index in the document,
 *        pre_tokenizer.rs:252-256).
 *   These are the fields of `Encoding` (tokenizer/encoding.rs:11-31) that
 *   carry information; type_ids / attention_mask / special_tokens_mask follow
 *   from the special-token layout (tkamd_tokenizer_specials) and the padding
 *   counts (tkamd_batch_pad_counts) and are synthesised by the host shim.
 *   A `truncation` / `padding` section of tokenizer.json is honoured
 *   (utils/truncation.rs, utils/padding.rs); with TKAMD_WANT_OVERFLOW the
 *   `overflowing` encodings a truncation leaves behind (Encoding::truncate,
 *   tokenizer/encoding.rs:307-395) are further encodings of the result.
 *   EncodeInput::Dual (pairs), InputSequence::PreTokenized (lists of words) and
 *   batches that mix single sequences and pairs (tokenizer/mod.rs:225-290) are
 *   the same buffers with one more CSR: TKAMD_PAIRS, tkamd_encode_batch_words,
 *   tkamd_encode_batch_mixed below.
 */
#ifndef TOKENIZERS_AMD_H
#define TOKENIZERS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (Result<_, Box<dyn Error>> at tokenizer/mod.rs:51-52) ---- */
#define TKAMD_OK              0
#define TKAMD_ERR_INVALID    -1   /* bad argument / malformed tokenizer.json                    */
#define TKAMD_ERR_UNSUPPORTED -2  /* tokenizer.json uses a component outside this hot path     */
#define TKAMD_ERR_DEVICE     -3   /* HIP runtime error (no device, OOM, launch failure)        */
#define TKAMD_ERR_MODEL      -4   /* model error the reference raises too (e.g. MissingUnkToken,
                                     models/wordlevel/mod.rs:175-177, wordpiece/mod.rs:229-232) */

/* ---- encode flags (OffsetType at tokenizer/pre_tokenizer.rs:10-17) ---- */
#define TKAMD_OFFSETS_NONE   0u   /* encode_batch_fast          (mod.rs:1382-1401)             */
#define TKAMD_OFFSETS_BYTE   1u   /* Rust encode_batch          (mod.rs:1337-1356)             */
#define TKAMD_OFFSETS_CHAR   2u   /* encode_batch_char_offsets  (mod.rs:1360-1379; Python)     */
#define TKAMD_OFFSETS_MASK   3u
#define TKAMD_WANT_WORD_IDS  4u   /* also produce Encoding.words                               */
#define TKAMD_ADD_SPECIAL    8u   /* add_special_tokens=true: PostProcessor::process for a single sequence
                                     (processors/bert.rs:51-120, roberta.rs, template.rs:544-590): special ids
                                     around every document, offsets (0,0), word id 0xFFFFFFFF (None)          */

#define TKAMD_PAIRS          16u   /* EncodeInput::Dual (tokenizer/mod.rs:871-889): documents 2i and 2i+1 are sequence A and B of
                                     encoding i (n_docs must be even); the result holds n_docs / 2 encodings: the pair is
                                     truncated together, laid out by the post-processor's pair template (type ids), padded  */

#define TKAMD_WANT_OVERFLOW   32u   /* Encoding.overflowing: with a `truncation` section, what a single sequence loses to the cut comes back as
                                     further encodings -- windows of max_length tokens (less the special tokens) that share `stride` tokens
                                     with their neighbour (tokenizer/encoding.rs:307-395), each with the same special tokens and padding as
                                     the truncated encoding (processors/bert.rs:88-125, encoding.rs:466-469).  The result then holds
                                     n_encodings >= n_docs encodings: every document's own followed by its overflowing ones in the
                                     reference's order; tkamd_batch_encoding_docs / d_enc_docs name the document of each.  A PAIR
                                     leaves every combination of its two sequences' windows (Encoding::merge_with, encoding.rs:408-432):
                                     with F the sequence the template names first, S the other, f_1.. / s_1.. their overflowing
                                     windows, the pair's encodings are F+S; then for every f_x: f_x+S, f_x+s_1, f_x+s_2 ..; then
                                     F+s_1, F+s_2 .. -- the reference's flat `overflowing` list (it also hangs f_x+s_* below f_x+S and
                                     f_*+s_y below F+s_y as nested lists: the same encodings; tkamd_batch_encoding_parts names the
                                     windows so a binding can rebuild them).  Ignored without a `truncation` section.                */

/* Readable slack the caller must leave after text[n_bytes] for the device entry
 * points (kernels read whole 16-byte words).  The host entry pads internally. */
#define TKAMD_NO_SPECULATION  64u   /* A tokenizer with added tokens runs a batch as if its text held none (natural text holds no special token: one
                                     detection pass instead of the matching passes of AddedVocabulary::extract_and_normalize,
                                     added_vocabulary.rs:523-564); a batch whose text does hold the content of one is run again with the matching
                                     passes when the call synchronises (the host entries before they return, the device entry in
                                     tkamd_device_sync) -- results are the reference's either way.  A caller of the device entry that
                                     consumes the results stream-ordered behind the call, WITHOUT tkamd_device_sync in between, sets this
                                     flag: the matching passes run outright.                                                            */
#define TKAMD_TEXT_PAD 64

typedef struct tkamd_tokenizer tkamd_tokenizer;  /* immutable after creation; owns device tables */
typedef struct tkamd_batch     tkamd_batch;      /* one encode_batch result (host copies)        */

/* What was recognised in tokenizer.json (for the host shim / diagnostics). */
typedef struct tkamd_info {
    int32_t model;          /* 1 BPE, 2 WordPiece, 3 WordLevel                                  */
    int32_t pre_tokenizer;  /* 1 ByteLevel(GPT-2 regex; also that regex spelled as a Split), 2 Split(a pattern of the tiktoken family:
                               Llama-3 / cl100k, Qwen2, o200k, tekken ...; pre_tokenizers/split.rs:76-105) + ByteLevel, 3 Whitespace,
                               4 WhitespaceSplit, 5 BertPreTokenizer, 6 ByteLevel(use_regex=false) */
    int32_t normalizer;     /* 0 none, 1 BertNormalizer                                         */
    int32_t vocab_size;
    int32_t n_merges;
    int32_t add_prefix_space;
    int32_t ignore_merges;
    int32_t n_added_tokens; /* special/added tokens registered in the JSON                       */
    int32_t device;         /* HIP device ordinal, -1 = host-only handle (no kernels)           */
    int32_t n_direct_words; /* entries of the whole-word table proven merge-stable (see DESIGN)  */
    int32_t truncation;         /* max_length of the `truncation` section, -1 = none (utils/truncation.rs)          */
    int32_t padding;            /* 0 none, 1 pad on the right, 2 on the left (utils/padding.rs)                       */
    int32_t pad_id;
    int32_t pad_type_id;
    int32_t word_disp_entries;  /* always 0 since round 4 (the whole-word table is a two-choice table without displacements);
                                   the field keeps the struct's layout                                                */
    int32_t merge_disp_entries; /* displacement entries of the merge table's perfect hash (> 16384: the kernels read them
                                   from global memory instead of their LDS copy)                                      */
} tkamd_info;

/* Replaces Tokenizer::from_file / from_str (tokenizer/mod.rs:468-472, serialization.rs:104-171).
 * `device` >= 0: build HBM-resident tables on that HIP device.  `device` == -1: parse and build
 * host tables only (no GPU needed; encode calls then fail with TKAMD_ERR_DEVICE). */
int tkamd_tokenizer_from_json(const char* json, size_t json_len, int device, tkamd_tokenizer** out);
void tkamd_tokenizer_free(tkamd_tokenizer* tok);
int tkamd_tokenizer_info(const tkamd_tokenizer* tok, tkamd_info* info);

/* ---- one call, every GPU ------------------------------------------------------------------------
 * TokenizerImpl::encode_batch is ONE call that uses every parallel resource of the machine: it maps `encode` over the inputs on the
 * Rayon pool (tokenizer/mod.rs:1345-1348, utils/parallelism.rs:85-106).  The counterpart here: a handle made over a LIST of devices
 * replicates the tables on each (<= 5 MB) and the host entries (tkamd_encode_batch, tkamd_encode_batch_words) then cut every batch
 * into one contiguous, byte-balanced run of documents per device -- one host thread and one stream per device, each shard's text
 * going up its own PCIe link -- and return ONE result in document order.  How the shards' results meet is the handle's collect mode:
 *   TKAMD_COLLECT_HOST       every device copies its ids straight into its slice of the one pinned result (no collective): the
 *                            default; N links carry the result.
 *   TKAMD_COLLECT_ROOT_P2P   the peers push their shards into one buffer on devices[0] over xGMI (hipMemcpyPeerAsync), one D2H.
 *   TKAMD_COLLECT_ROOT_RCCL  the same gather with RCCL: ncclSend on every peer, ncclRecv at the displacement on devices[0]
 *                            (RCCL has no gatherv; librccl.so is opened at first use) -- the reference's `collect()` of the
 *                            Vec<Encoding> as a collective, for consumers that want the result on one GPU.
 * A device may be named more than once (two shards then share a GPU: how the sharded path is tested on a one-GPU box; RCCL needs
 * distinct devices).  n_devices == 0: the list comes from the environment, TOKENIZERS_GPU_DEVICES = "all" | "0,2,3" (unset: device
 * 0).  BatchLongest padding couples the documents of a batch through one number -- the longest encoding: the shards exchange theirs and
 * pad to the batch's (utils/padding.rs:55-63).  TKAMD_WANT_OVERFLOW (a shard's encodings are counted when its kernels are done, its
 * document indices rebased) and mixed batches (cut between inputs) are sharded like the rest; a batch of
 * less than 1 MB per device runs on devices[0] alone (TKAMD_SHARD_MIN_KB, read when the handle is made).  The device-buffer entries and decode_batch run on devices[0]. */
#define TKAMD_COLLECT_HOST      0
#define TKAMD_COLLECT_ROOT_P2P  1
#define TKAMD_COLLECT_ROOT_RCCL 2
int tkamd_tokenizer_from_json_devices(const char* json, size_t json_len, const int* devices, int n_devices, tkamd_tokenizer** out);
int tkamd_tokenizer_set_collect(tkamd_tokenizer* tok, int mode);
/* devices[0 .. *n) of the handle (1 entry for a single-device handle, 0 for a host-only one). */
int tkamd_tokenizer_devices(const tkamd_tokenizer* tok, int* devices, int cap, int* n);
/* The last sharded host-entry call: per device of the handle, the bytes of its shard and the wall milliseconds its host thread was
 * busy (H2D + kernels + its part of the collect) -- BASELINE configs[4]'s "per-GPU busy time imbalance (max / mean)". */
int tkamd_shard_stats(const tkamd_tokenizer* tok, int64_t* shard_bytes, double* busy_ms, int cap, int* n);

/* Special-token layout of the post-processor for a single sequence: ids inserted before / after every document
 * when TKAMD_ADD_SPECIAL is set.  Returns TKAMD_ERR_UNSUPPORTED if the post-processor is outside the path. */
int tkamd_tokenizer_specials(const tkamd_tokenizer* tok, uint32_t* prefix_ids, int32_t* n_prefix, uint32_t* suffix_ids,
                             int32_t* n_suffix, int32_t cap);

/* The same for a pair (TKAMD_PAIRS): the post-processor's pair template as pieces of three values each -- kind (0 sequence A,
 * 1 sequence B, 2 special token), token id (kind 2), type id -- in output order; with_specials = 0: the layout without
 * add_special_tokens (A then B, PostProcessor::default_process).  *n_pieces = number of pieces (written even beyond cap). */
int tkamd_tokenizer_pair_template(const tkamd_tokenizer* tok, int with_specials, uint32_t* pieces, int32_t cap, int32_t* n_pieces);

/* Thread-local message of the last failing call on this thread (error.rs:26-31 maps
 * the reference's error to `Exception(str(e))`; the shim does the same with this). */
const char* tkamd_last_error(void);

/* ---- host-buffer entry: H2D + kernels + D2H -------------------------------------------------
 * Replaces TokenizerImpl::encode_batch{,_char_offsets,_fast}(inputs, add_special_tokens=false)
 * for Single(Raw) inputs.  `text`/`doc_offsets` are borrowed for the call.  The result is owned
 * by the library until tkamd_batch_free. */
int tkamd_encode_batch(tkamd_tokenizer* tok, const uint8_t* text, const int64_t* doc_offsets,
                       int64_t n_docs, uint32_t flags, tkamd_batch** out);

/* is_pretokenized = true (InputSequence::PreTokenized, tokenizer/mod.rs:225-290; encode_single_sequence :782-795): every input
 * sequence is a list of words.  The reference encodes each word on its own -- AddedVocabulary, normalizer, pre-tokenizer, model:
 * a ByteLevel add_prefix_space therefore applies to every word, and offsets are relative to the WORD -- and merges the
 * encodings with word_ids = the word's index in its sequence; truncation, special tokens and padding then see one encoding per
 * sequence.  `text` is the concatenation of all words, word_offsets[n_words + 1] their byte CSR, seq_offsets[n_seqs + 1] the CSR of
 * the sequences over the words.  With TKAMD_PAIRS sequences 2i and 2i + 1 are sequence A and B of encoding i.  The result holds one
 * encoding per sequence (or pair). */
int tkamd_encode_batch_words(tkamd_tokenizer* tok, const uint8_t* text, const int64_t* word_offsets, int64_t n_words,
                             const int64_t* seq_offsets, int64_t n_seqs, uint32_t flags, tkamd_batch** out);

/* One call for a Vec<EncodeInput> that mixes EncodeInput::Single and EncodeInput::Dual items (tokenizer/mod.rs:225-290; encode_batch
 * maps `encode` over them, :1337-1356, and pads the lot together, utils/padding.rs:50-81).  The sequences of the batch are laid out as
 * for tkamd_encode_batch (seq_offsets == NULL: `text` + doc_offsets[n_docs + 1], a sequence is a document) or as for
 * tkamd_encode_batch_words (seq_offsets[n_seqs + 1] over the words word_offsets = doc_offsets[n_docs + 1]); input i is the sequences
 * [input_offsets[i], input_offsets[i + 1]): one -- a single sequence -- or two -- sequence A and B of a pair.  Every input is cut,
 * laid out by the post-processor's template of ITS kind (n_added_tokens(is_pair) taken off max_length, mod.rs:1270-1284) and the batch
 * is padded as one (BatchLongest over every input).  TKAMD_PAIRS must not be set.  The result holds one encoding per input (plus the
 * overflowing ones with TKAMD_WANT_OVERFLOW; tkamd_batch_encoding_parts then gives (window, 0) for a single sequence); type ids and
 * sequence ids are written for every token.  A batch that turns out to hold one kind only is handed to the entry of that kind; a
 * mixed one runs as one slice (a multi-device handle: one shard per device, cut between inputs).  TKAMD_ERR_INVALID for an input of no or more than two sequences. */
int tkamd_encode_batch_mixed(tkamd_tokenizer* tok, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs,
                             const int64_t* seq_offsets, int64_t n_seqs, const int64_t* input_offsets, int64_t n_inputs,
                             uint32_t flags, tkamd_batch** out);

/* tkamd_encode_batch for a caller that is still FILLING `text` while the call runs.  The reference's Python binding turns every
 * input into an owned Rust String first and encodes afterwards (bindings/python/src/tokenizer.rs:1312-1338: the extraction loop,
 * then py.allow_threads around encode_batch); a binding of this library packs its strings into one buffer anyway -- with this entry
 * the packing of the batch's tail overlaps the H2D copy and the kernels of its head.  doc_offsets[0 .. n_docs] is complete on entry;
 * bytes [0, *ready_bytes) of `text` are valid, *ready_bytes only grows (written with release semantics by the caller's packing
 * threads, read with acquire semantics here) and reaches doc_offsets[n_docs].  The call never reads a byte it has not seen announced.
 * `consumed(user)`, if not NULL, is called once from the calling thread as soon as the call has seen *ready_bytes at its final value
 * (the caller's source objects are no longer needed: a Python binding releases the GIL there) -- also on every error path that
 * waited that long; a call that fails before has not called it.  A producer that cannot finish stores a NEGATIVE value in
 * *ready_bytes: the call stops waiting, calls `consumed` and fails with TKAMD_ERR_INVALID (without that signal a producer that
 * dies leaves the call waiting for ever).  pace == NULL: tkamd_encode_batch. */
typedef struct tkamd_pace {
    const int64_t* ready_bytes;
    void (*consumed)(void* user);
    void* user;
} tkamd_pace;
int tkamd_encode_batch_paced(tkamd_tokenizer* tok, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, uint32_t flags,
                             const tkamd_pace* pace, tkamd_batch** out);

int64_t         tkamd_batch_n_docs(const tkamd_batch* b);       /* encodings in the result (= documents / sequences / pairs, plus the
                                                                   overflowing encodings with TKAMD_WANT_OVERFLOW)              */
int64_t         tkamd_batch_n_tokens(const tkamd_batch* b);
const uint32_t* tkamd_batch_ids(const tkamd_batch* b);          /* [n_tokens]                   */
const int64_t*  tkamd_batch_tok_offsets(const tkamd_batch* b);  /* [n_docs+1] CSR into ids      */
const uint32_t* tkamd_batch_offsets(const tkamd_batch* b);      /* [n_tokens][2] or NULL        */
const uint32_t* tkamd_batch_word_ids(const tkamd_batch* b);     /* [n_tokens] or NULL           */
const uint8_t*  tkamd_batch_type_ids(const tkamd_batch* b);     /* [n_tokens] Encoding.type_ids (pairs; single sequences under a TemplateProcessing with type ids), or NULL: 0, pad_type_id on padding */
const uint8_t*  tkamd_batch_sequence_ids(const tkamd_batch* b); /* [n_tokens] 0 / 1 = token of sequence A / B, 2 = special token, 3 = padding, or NULL
                                                                   (single sequences: derived from tkamd_tokenizer_specials and the pad counts)      */
const uint32_t* tkamd_batch_pad_counts(const tkamd_batch* b);   /* [n_docs] padding tokens of each encoding (at the side
                                                                   tkamd_info.padding names), NULL without a `padding` section:
                                                                   attention_mask = 0, special_tokens_mask = 1 on them      */
const uint32_t* tkamd_batch_encoding_docs(const tkamd_batch* b);/* [n_docs] TKAMD_WANT_OVERFLOW with a `truncation` section: the document
                                                                   (sequence) every encoding belongs to, ascending -- a document's
                                                                   first encoding is the truncated one, the rest are its
                                                                   Encoding.overflowing in order; NULL otherwise                */
const uint32_t* tkamd_batch_encoding_parts(const tkamd_batch* b);/* [n_docs][2] TKAMD_WANT_OVERFLOW | TKAMD_PAIRS: which window of sequence A
                                                                   and of sequence B every encoding combines (0 = the truncated
                                                                   sequence itself, 1.. = its overflowing windows); NULL otherwise */
void            tkamd_batch_free(tkamd_batch* b);

/* ---- page-locked memory for the caller's side of the host entry ---------------------------------
 * tkamd_encode_batch takes `text` / `doc_offsets` from any host memory; from pageable memory the runtime stages the copy through
 * its own bounce buffers (about half the link's rate, and the calling thread does the copying).  The binding has to pack the
 * documents into one buffer anyway: packing them into a block from here costs nothing extra and lets the H2D
 * copies run as plain DMA.  Blocks are portable across the devices of a multi-device handle; free them with tkamd_pinned_free
 * (never free()).  Fails with TKAMD_ERR_DEVICE where no HIP device exists. */
int  tkamd_pinned_alloc(size_t bytes, void** out);
void tkamd_pinned_free(void* p);

/* ---- device-buffer entry: inputs already resident in HBM, outputs stay in HBM ---------------
 * Enqueues the whole path on `hip_stream` (a hipStream_t, NULL = default stream) and returns
 * without synchronising.  The output pointers refer to the handle's workspace and stay valid
 * until the next encode call on the same handle.  d_n_tokens[0] holds the token count once the
 * stream has drained; tkamd_device_sync() waits for it and returns it. */
typedef struct tkamd_device_result {
    const uint32_t* d_ids;          /* [n_tokens]                                               */
    const int64_t*  d_tok_offsets;  /* [n_docs+1]                                               */
    const uint32_t* d_offsets;      /* [n_tokens][2] or NULL                                    */
    const uint32_t* d_word_ids;     /* [n_tokens] or NULL                                       */
    const int64_t*  d_n_tokens;     /* [1]                                                      */
    const int64_t*  d_n_pretokens;  /* [1] number of pre-tokens (splits) in the batch           */
    const uint32_t* d_pad_counts;   /* [n_docs] padding tokens per encoding, or NULL            */
    const uint8_t*  d_type_ids;     /* TKAMD_PAIRS, or a single template with type ids: [n_tokens], else NULL */
    const uint8_t*  d_seq_ids;      /* TKAMD_PAIRS: [n_tokens] 0 / 1 / 2 special / 3 padding    */
    const uint32_t* d_enc_docs;     /* TKAMD_WANT_OVERFLOW (with a `truncation` section): [n_encodings] document of every encoding, else NULL;
                                       d_tok_offsets / d_pad_counts then have n_encodings (+ 1) entries                                        */
    const int64_t*  d_n_encodings;  /* [1] with d_enc_docs, else NULL (the call itself waits for this count: it sizes the result)              */
    const uint32_t* d_enc_parts;    /* TKAMD_WANT_OVERFLOW | TKAMD_PAIRS: [n_encodings][2] window of A / of B, else NULL                       */
    int64_t         ids_capacity;   /* an upper bound of n_tokens known when the call returns (before the stream has drained): a consumer on
                                       the same stream may treat d_ids (and the other per-token arrays) as arrays of this many elements and
                                       read the count from d_n_tokens on the device.  0: no such bound (the padded / overflowing encodings of
                                       a `truncation` / `padding` section are sized from the data)                                            */
} tkamd_device_result;

int tkamd_encode_batch_device(tkamd_tokenizer* tok, const uint8_t* d_text, const int64_t* d_doc_offsets,
                              int64_t n_docs, int64_t n_bytes, uint32_t flags, void* hip_stream,
                              tkamd_device_result* out);
/* tkamd_encode_batch_words on device-resident buffers (d_seq_offsets: [n_seqs + 1] int64 in HBM as well). */
int tkamd_encode_batch_words_device(tkamd_tokenizer* tok, const uint8_t* d_text, const int64_t* d_word_offsets, int64_t n_words,
                                    int64_t n_bytes, const int64_t* d_seq_offsets, int64_t n_seqs, uint32_t flags, void* hip_stream,
                                    tkamd_device_result* out);
int tkamd_device_sync(tkamd_tokenizer* tok, void* hip_stream, int64_t* n_tokens, int64_t* n_pretokens);

/* ---- decode_batch: token ids -> text -----------------------------------------------------------
 * Replaces Tokenizer::decode_batch (tokenizer/mod.rs:1404-1416) = map of Tokenizer::decode (mod.rs:935-953):
 * id -> token string (added vocabulary first, added_vocabulary.rs:239-246; ids without a token are dropped),
 * specials dropped when TKAMD_SKIP_SPECIAL (skip_special_tokens), then the `decoder` section:
 * ByteLevel (pre_tokenizers/byte_level.rs:155-171), WordPiece (decoders/wordpiece.rs:46-64), BPEDecoder (decoders/bpe.rs:26-39),
 * ByteFallback (byte_fallback.rs:27-67), Fuse (fuse.rs:24-29), Replace with a literal pattern (normalizers/replace.rs:88-106), Strip
 * (strip.rs:27-60), a Sequence (sequence.rs:26-33) of the shape [Replace*, Strip*, ByteFallback?, Fuse?, Strip(c, <= 1, 0)?], CTC
 * (ctc.rs:45-63) or none (join with " ").  Other decoders / shapes -> TKAMD_ERR_UNSUPPORTED.  The result is the raw byte string per sequence; the
 * ByteLevel decoder's String::from_utf8_lossy (byte_level.rs:170) is left to the caller (bytes that do not
 * form valid UTF-8 can only come from id sequences that split a character), e.g. Python's
 * bytes.decode("utf-8", "replace"), which substitutes the same maximal invalid subparts. */
#define TKAMD_SKIP_SPECIAL 1u
typedef struct tkamd_text tkamd_text;
int tkamd_decode_batch(tkamd_tokenizer* tok, const uint32_t* ids, const int64_t* tok_offsets, int64_t n_docs,
                       uint32_t flags, tkamd_text** out);
int64_t         tkamd_text_n_docs(const tkamd_text* b);
int64_t         tkamd_text_n_bytes(const tkamd_text* b);
const uint8_t*  tkamd_text_bytes(const tkamd_text* b);         /* [n_bytes]                    */
const int64_t*  tkamd_text_doc_offsets(const tkamd_text* b);   /* [n_docs+1] CSR into bytes    */
void            tkamd_text_free(tkamd_text* b);
/* The byte string token `id` contributes (first_position != 0: as the first kept token of a sequence), straight from
 * the load-time decode tables; *flags = 0 ordinary, 1 special, 2 no token has this id.  Works on host-only handles. */
int tkamd_decode_token(const tkamd_tokenizer* tok, uint32_t id, int first_position, uint8_t* out, int32_t cap,
                       int32_t* len, int32_t* flags);

/* Host-side probes of the load-time tables -- the lookups the kernels perform, on the host copy (test hooks; work on
 * host-only handles).  Return 1 on a hit, 0 on a miss, < 0 on a bad argument.
 *   tkamd_probe_word : raw pre-token bytes -> token id (vocab.get of bpe/model.rs:559-567, wordlevel/mod.rs:162-178);
 *                      *flags bit 0 = WORD_DIRECT (byte-level BPE: the merges of these bytes yield exactly [id])
 *   tkamd_probe_merge: (left id, right id) -> (rank, new id)   (the `merges` map of bpe/model.rs:252-275)          */
int tkamd_probe_word(const tkamd_tokenizer* tok, const uint8_t* bytes, int32_t len, uint32_t* id, uint32_t* flags);
int tkamd_probe_merge(const tkamd_tokenizer* tok, uint32_t left, uint32_t right, uint32_t* rank, uint32_t* new_id);
/* One edge of the WordPiece byte trie (the longest-match walk of wordpiece/mod.rs:245-258 as a trie): (node, byte) ->
 * (child, id of the piece ending there or 0xFFFFFFFF); node 0 = word-initial pieces, 1 = continuation pieces. */
int tkamd_probe_trie(const tkamd_tokenizer* tok, uint32_t node, uint32_t byte, uint32_t* child, uint32_t* id);
/* Encoding::truncate (tokenizer/encoding.rs:307-395) of a sequence of n_tokens to max_len with `stride`, direction Right (left = 0) or
 * Left, straight from the function the epilogue kernels call (csrc/overflow_core.hpp): returns the number of encodings it leaves
 * (1 = nothing is cut; 0 = the reference's assert stride < max_len) and, for part < that number, its token range (part 0 = the
 * truncated encoding, 1.. = Encoding.overflowing in order). */
int tkamd_probe_truncation(uint64_t n_tokens, uint32_t max_len, uint32_t stride, int left, uint32_t part, uint64_t* start, uint64_t* count);
/* BertNormalizer::normalize (normalizers/bert.rs:92-138) of ONE code point from the host copy of the generated tables:
 * out[0..*n) (at most 12 code points; 0 = the char is removed), *refused = 1 for the characters NFD's canonical ordering could
 * move (they survive the Mn filter with a non-zero combining class): whether it does depends on their neighbours, see
 * tkamd_probe_bert_alone. */
int tkamd_probe_bert_norm(const tkamd_tokenizer* tok, uint32_t cp, uint32_t* out, int32_t* n, int32_t* refused);
/* BertNormalizer strip_accents on the character whose lead byte is text[pos] (text[0 .. n) = one piece handed to the normalizer: a
 * document, or what lies between two added-token matches): *reorder = 1 if it survives the Mn filter with a non-zero combining
 * class (NFD's canonical ordering could move it), *alone = 1 if it is alone in its run of non-starters -- nothing moves; 0: the
 * device puts that run into canonical order (csrc/bert_norm_core.hpp bn_fix_run).  The very function the kernels call, on the host
 * copy of the tables. */
int tkamd_probe_bert_alone(const tkamd_tokenizer* tok, const uint8_t* text, int64_t n, int64_t pos, int32_t* reorder, int32_t* alone);
/* The NFD form of one code point as bn_fix_run sees it, from the host copy of the tables: *packed = n | per piece q, at bit 3 + 7q,
 * (rank of its canonical combining class among the classes in use, 0 = starter) | (survives the Mn filter) << 6; 0: no piece of
 * the form is a non-starter.  *flags = the per-character table flags (1 dropped by clean_text, 2 whitespace, 4 CJK, 8 reorderable,
 * 16 / 32 has an NFD / lowercase expansion, 64 / 128 first / last piece is a non-starter). */
int tkamd_probe_bert_nfd(const tkamd_tokenizer* tok, uint32_t cp, uint32_t* packed, uint32_t* flags);
/* Class flags of one code point from the host copy of the generated Unicode table: bit 0 \p{L}, 1 \p{N}, 2 \s (as
 * Oniguruma sees them, byte_level.rs:43-46), 3 \w, 4 \s (regex crate, whitespace.rs:22), 5 char::is_whitespace, 6 is_bert_punc. */
int tkamd_probe_unicode_flags(const tkamd_tokenizer* tok, uint32_t cp, uint32_t* flags);

/* Tokenizer.encode_special_tokens (tokenizer/mod.rs:752-759, AddedVocabulary::set_encode_special_tokens added_vocabulary.rs:249-255):
 * value != 0 -> the special tokens of the added vocabulary are no longer extracted from the text (find_matches skips them,
 * added_vocabulary.rs:450-453): their characters go through the normalizer, pre-tokenizer and model like any text.  Applies to the
 * batches enqueued after the call.  Off by default. */
int tkamd_encode_special_tokens(tkamd_tokenizer* tok, int value);

/* ---- word cache --------------------------------------------------------------------------------
 * What BPE::tokenize_with_cache keeps per thread (models/bpe/model.rs:573-586, utils/cache.rs): pre-token bytes -> its tokens.
 * `enable` != 0: every workspace of the handle keeps a table (1 M entries, in HBM) of the <= 16-byte words its batches have merged
 * (results of <= 4 tokens); later ids-only batches (TKAMD_OFFSETS_NONE) look such a word up instead of merging it again.  Like the
 * reference's cache it only fills, never evicts, and never changes a result.  `clear` != 0: forget everything (every workspace
 * zeroes its table before its next batch).  Off by default -- the default is the cache's counterpart WITHIN a batch, which keeps no
 * state: the first occurrence of a word the static tables do not settle goes to the model kernel, its other occurrences in the same
 * batch share the result (in-batch claims, csrc/kernels/lookup.hip; with or without offsets).  Switching this cache on replaces them.
 * A batch that shared nothing (more than 35 % of its pre-tokens still queued) pauses the claims for the handle's next 32 batches
 * : on text that never repeats a word they only cost. */
int tkamd_word_cache(tkamd_tokenizer* tok, int enable, int clear);

/* ---- measurement hooks (bench.py roofline leg; not part of the reference surface) -----------
 * With profiling on, every kernel launch of the next device/host encode calls is bracketed by
 * HIP events on the launch stream.  tkamd_profile_read returns, per kernel, the accumulated
 * milliseconds and launch count since the last reset. */
#define TKAMD_MAX_STAGES 24
typedef struct tkamd_stage_time {
    char    name[48];
    double  ms_total;
    int64_t launches;
} tkamd_stage_time;
int tkamd_profile_enable(tkamd_tokenizer* tok, int on);
int tkamd_profile_read(tkamd_tokenizer* tok, tkamd_stage_time* stages, int max_stages, int* n_stages, int reset);

/* Work-queue sizes of the last synchronised batch: out[0] = pre-tokens sent to the 16-lane merge
 * kernel, out[1] = to the 64-lane kernel, out[2] = to the workgroup (long) kernel. */
int tkamd_profile_counters(tkamd_tokenizer* tok, uint32_t* out, int n);

/* Where the two longest kernels spend their time.  With the test hook TKAMD_PHASES (TKAMD_TEST_HOOKS=1 TKAMD_PHASES=1) the whole-word
 * lookup (which = 0) and the token compaction (which = 1) run as diagnostic instantiations that stamp the shader clock behind the
 * barriers that end their phases; out[0..7] = ticks summed over all workgroups and batches since the last reset
 * (lookup: 0 staging a tile, 1 expanding the mask bits, 2 pass 1 (LDS hot table), 3 pass 2 (perfect hash), 4 pass 3 (claims) +
 * waiting for the slowest wavefront; compaction: 0 loads + scan + publish, 1 LDS scatter, 2 look-back wait, 3 copy-out; 7 = the
 * whole kernel, both).  All zero without the variable.  A development aid: never set it in a measured run. */
int tkamd_debug_phases(tkamd_tokenizer* tok, int which, uint64_t* out, int reset);

/* Library version string, e.g. "tokenizers_amd 0.1.0 (gfx950)". */
const char* tkamd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TOKENIZERS_AMD_H */
