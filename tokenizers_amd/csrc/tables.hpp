// Table layouts and hash functions shared by the host builder (host_model.cpp) and the
// HIP kernels (kernels.hip).  Everything here is plain-old-data that is copied verbatim
// into HBM at tokenizer load (tkamd_tokenizer_from_json).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TK_HD __host__ __device__ __forceinline__
#else
#define TK_HD inline
#endif

namespace tkamd {

// ---- Unicode class flags (generated data: unicode_ranges.inc) -------------------------------
enum : uint8_t {
    UC_ONIG_L = 1,    // \p{L}  (Oniguruma)    byte_level.rs:43-46
    UC_ONIG_N = 2,    // \p{N}
    UC_ONIG_S = 4,    // \s
    UC_RX_W = 8,      // \w     (regex crate)  whitespace.rs:22
    UC_RX_S = 16,     // \s     (regex crate)
    UC_RUST_WS = 32,  // char::is_whitespace   whitespace.rs:38, bert.rs:15
    UC_BERT_P = 64,   // is_bert_punc          bert.rs:5-7
};

// 2-stage code-point table: stage1[cp >> 8] -> block index, stage2[block*256 + (cp & 255)] -> flags
constexpr int UC_STAGE1_LEN = 0x1100;

// ---- BPE merge table: (left id, right id) -> (rank, new id), models/bpe/model.rs:252-275 -----
// Static 2-choice cuckoo table: a key lives in slot h1(key) or h2(key); a lookup is exactly two
// independent 16-byte loads, hit or miss, no probe loop (built once at load, never mutated).
struct MergeSlot {
    uint32_t a, b, rank, new_id;
};
constexpr uint32_t MERGE_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t RANK_NONE = 0xFFFFFFFFu;

TK_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// 24-bit multiply (v_mul_u32_u24 / v_mad_u32_u24: full rate on CDNA, where the 32-bit v_mul_lo_u32 is quarter rate).
// Only the low 24 bits of each operand take part; token ids are far below 2^24.
TK_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)) & 0xFFFFFFFFull);
#endif
}
// Pair hashes: two multiply-adds and one xor-shift each.  The seed perturbs the MULTIPLIERS (an additive seed would
// leave colliding pairs colliding); table construction retries seeds until every key has its own slot, so the
// hash only has to spread keys, not be strong.  A merge probe is what the merge loops spend their ALU time on.
TK_HD uint32_t merge_hash1(uint32_t a, uint32_t b, uint32_t seed) {
    const uint32_t k1 = 0x9E3779u ^ (seed & 0xFFFFFEu), k2 = 0x85EBCBu ^ ((seed >> 7) & 0xFFFFFEu);
    const uint32_t x = mul24(a, k1) + mul24(b, k2);
    return x ^ (x >> 15);
}
TK_HD uint32_t merge_hash2(uint32_t a, uint32_t b, uint32_t seed) {
    const uint32_t k3 = 0xC2B2AFu ^ ((seed >> 3) & 0xFFFFFEu), k4 = 0x165667u ^ ((seed >> 11) & 0xFFFFFEu);
    const uint32_t y = mul24(a, k3) + mul24(b, k4) + 0x5BD1E995u;
    return y ^ (y >> 13);
}

// Hash-and-displace perfect hash for the merge table: bucket = hash1 & bmask selects a 16-bit displacement d,
// the key then lives in exactly ONE slot, (hash2 + d * PH_MULT) & mask.  A lookup -- hit or miss -- is one
// 16-byte load; the displacement array is small enough (<= 32 KB for 50k merges) to sit in LDS.
constexpr uint32_t PH_MULT = 0x9E3779u;
constexpr int DISP_LDS_MAX = 16384;              // merge displacement entries the merge kernels cache in LDS (32 KB); the builder stays within it if it can
TK_HD uint32_t ph_slot(uint32_t h2, uint32_t d, uint32_t mask) { return (h2 + mul24(d, PH_MULT)) & mask; }

// ---- whole-word table: raw pre-token bytes (<= 16) -> token id ------------------------------
// Serves BPE `ignore_merges` (bpe/model.rs:559-567), WordLevel (wordlevel/mod.rs:162-178) and the
// merge-stable shortcut: key = bytes zero-padded to 16 + length, 32-byte slots.
// Static TWO-CHOICE table, HOST ONLY since round 5: a key lives in slot word_slot_a or word_slot_b of its hash (cuckoo insertion at
// load, load factor <= 0.4).  It is the copy of record -- the WORD_DIRECT flags are proved on it (verify_direct_words), tkamd_probe_word
// reads it, and the table the lookup kernel PROBES, the short-word table below, is built from it.  (Round 4 probed it on the device,
// both 32-byte slots in one round trip: slower than one 16-byte slot behind an 8-bit displacement, profiles/r4i-k.)
struct WordSlot {
    uint64_t lo, hi;
    uint32_t len;    // 0 = empty slot
    uint32_t id;
    uint32_t flags;  // WORD_DIRECT: BPE merges of these bytes yield exactly [id]
    uint32_t pad;
};
constexpr uint32_t WORD_DIRECT = 1u;
constexpr int WORD_MAX_KEY = 16;

// The bucket hash of the whole-word table continues the hot-table hash of the first 12 bytes + length (hot_hash below), so the
// lookup kernel pays for the key bytes once: h1 = mix32(hot_hash ^ bytes 12..15 ^ seed); the slot hash is one more round.
TK_HD uint32_t hot_hash(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint32_t seed);
TK_HD uint32_t word_hash1_from_hot(uint32_t hot, uint32_t k3) { return mix32(hot ^ (k3 * 0x165667B1u)); }
TK_HD uint32_t word_hash1(uint64_t lo, uint64_t hi, uint32_t len, uint32_t seed);
TK_HD uint32_t word_hash2(uint32_t h1) { return (h1 * 0x9E3779B1u) ^ (h1 >> 15); }
TK_HD uint32_t word_slot_a(uint32_t h1, uint32_t mask) { return h1 & mask; }
TK_HD uint32_t word_slot_b(uint32_t h1, uint32_t mask) {       // (never slot a: two real choices for every key)
    const uint32_t b = (word_hash2(h1) >> 9) & mask;
    return b == (h1 & mask) ? (b ^ 1u) : b;
}

// ---- the short-word table: every word of the table above once more, in 16-byte slots (device only, built when the flags above are
// final) -- what pass 2 of the lookup kernel probes:  slot = {k0, k1, k2, id | len << 24 | SHORTW_DIRECT}, bytes 12..15 of the key
// in a parallel array k3[slot] that only the 3 % of pre-tokens longer than 12 bytes read (same index: no dependent load).
// Hash-and-displace with EIGHT-bit displacements over SHORTW_BUCKETS buckets: the 8 KB of displacements sit in the kernel's LDS, so
// a probe is ONE 16-byte request at ONE random line.  bucket = h1 & (SHORTW_BUCKETS - 1), slot = shortw_slot(h1, key mix, d, mask).
// How it got there, as measured in round 4 (the table was displacement-in-HBM + one 32-byte slot before: 0.233..0.243 ms for the
// kernel on C2, 0.44..0.455 on out-of-distribution text): both 32-byte slots of a two-choice table at once -- four requests, ONE
// round trip -- 0.265 ms; both 16-byte slots of a two-choice table -- two requests, one round trip -- 0.257 ms (profiles/r4i_*,
// r4j_*): fewer dependent round trips, fewer requests, and slower -- a probe costs the random LINES it touches beyond the L2 (two
// there, one here), not the length of its chain.  One 16-byte slot behind an LDS displacement: 0.235 / 0.40 ms (r4k_*).
constexpr uint32_t SHORTW_DIRECT = 0x80000000u;
constexpr uint32_t SHORTW_LEN_SHIFT = 24, SHORTW_LEN_MASK = 0x1Fu, SHORTW_ID_MASK = 0xFFFFFFu;
constexpr int SHORTW_BUCKETS = 8192;
// The slot: double hashing on the displacement -- (base + d * step) & mask, base and step from the key bytes once more.  The table
// shares the 32-byte table's seed (the kernel hashes a key once), so it cannot answer an unlucky placement with another seed: a
// displacement that moves every key of a bucket by the SAME amount (the merge table's ph_slot) can never part two keys of a bucket
// whose bases agree under the mask -- a key-dependent step does, at the next d.
TK_HD uint32_t shortw_kmix(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) { return (k0 * 0x85EBCA77u) ^ (k1 * 0xC2B2AE3Du) ^ (k2 * 0x27D4EB2Fu) ^ (k3 * 0x165667B1u); }
TK_HD uint32_t shortw_slot(uint32_t h1, uint32_t kmix, uint32_t d, uint32_t mask) {
    const uint32_t base = ((h1 * 0x9E3779B1u) ^ (h1 >> 15)) ^ kmix, step = ((kmix * 0x9E3779B1u) >> 9) | 1u;
    return (base + d * step) & mask;
}

// ---- hot-word table: the lowest-id settled words of <= 12 bytes, copied into LDS by the lookup kernel ----
// slot = {k0, k1, k2, id | len << 24} (key bytes zero padded; len 0 = empty slot).  Hash-and-displace like the tables in HBM, so the
// table holds exactly the words it is meant to hold: bucket = hot_hash & (slots / 4 - 1) selects a 16-bit displacement (the array
// follows the slots, in the same buffer), the word lives in slot (hot_hash >> 12) + d.  (Round 3's table was direct mapped: of the
// 2,048 lowest ids 1,295 kept their slot, and 46 % of C2's pre-tokens hit; placed like this the same LDS hits 61 %, and 1,024
// slots hit 50 %.)  A word that could not be placed is simply not in the table: the perfect-hash table behind it still answers.
constexpr int HOT_MAX_KEY = 12;
constexpr int HOT_SLOTS = 1024;      // (what leaves three lookup workgroups on a CU, kernels/lookup.hip)
struct HotSlot {
    uint32_t k0, k1, k2, id_len;
};
TK_HD uint32_t hot_bucket(uint32_t h, uint32_t slots) { return h & (slots / 4u - 1u); }
TK_HD uint32_t hot_slot(uint32_t h, uint32_t d, uint32_t slots) { return ((h >> 12) + d) & (slots - 1u); }
constexpr int hot_table_bytes(int slots) { return slots * 16 + slots / 4 * 2; }      // the slots, then the displacements
// The seed (the one the perfect-hash builder settled on) goes in BEFORE the multiplies: two keys that collide under one seed
// must not collide under every seed, or the builder could never separate them.
TK_HD uint32_t hot_hash(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint32_t seed) {
    uint32_t h = ((k0 + seed) * 0x9E3779B1u) ^ ((k1 ^ seed) * 0x85EBCA77u) ^ ((k2 + (seed >> 7)) * 0xC2B2AE3Du) ^ (len * 0x27D4EB2Fu);
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    return h ^ (h >> 12);
}

TK_HD uint32_t word_hash1(uint64_t lo, uint64_t hi, uint32_t len, uint32_t seed) {
    return word_hash1_from_hot(hot_hash((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, len, seed), (uint32_t)(hi >> 32));
}

// ---- BPE over characters: initial symbols (bpe/model.rs:465-550), see HostModel::char_id and kernels/bpe.hip ----
constexpr uint32_t CHAR_NONE = 0xFFFFFFFFu;
constexpr uint32_t CHAR_TABLE_WORDS = 0x110000u * 4u;
enum : uint32_t {
    CB_ON = 1,            // the model is a BPE over characters
    CB_PREFIX = 2,        // continuing_subword_prefix glued to every char but the first
    CB_SUFFIX = 4,        // end_of_word_suffix glued to the last char
    CB_UNK = 8,           // an unk_token stands for chars the vocabulary lacks ...
    CB_UNK_MISSING = 16,  // ... or was configured but is not in the vocabulary: an error the moment it is needed
    CB_FUSE = 32,         // consecutive unknown chars become ONE unk symbol
    CB_BYTES = 64,        // byte_fallback: such a char becomes the <0xXX> tokens of its bytes (all 256 exist: checked at load)
};

// ---- vocabulary entries longer than 16 bytes: open addressing over the vocabulary blob, keyed by this hash of the bytes, four at a
// time (the last word zero padded; the length goes in first, so the padding is unambiguous) ----
TK_HD uint32_t long_key_hash_step(uint32_t h, uint32_t w) {
    h = (h ^ w) * 0x01000193u;
    return h ^ (h >> 15);
}
TK_HD uint32_t long_key_hash_init(uint32_t len) { return 2166136261u ^ (len * 0x9E3779B1u); }

// ---- decode_batch entry flags (in the length word of the first-position form) ----
constexpr uint32_t DEC_SPECIAL = 0x80000000u;   // special token: dropped when skip_special_tokens
constexpr uint32_t DEC_ABSENT = 0x40000000u;    // no token has this id: always dropped (mod.rs:938-941 filter_map)
constexpr uint32_t DEC_BYTE = 0x20000000u;      // a <0xXX> token under the ByteFallback decoder: word 0 of the entry is the byte (decoders/byte_fallback.rs:31-35)
constexpr uint32_t DEC_LEN_MASK = 0x1FFFFFFFu;

// ---- tokenizer kinds -------------------------------------------------------------------------
enum ModelKind { MODEL_NONE = 0, MODEL_BPE = 1, MODEL_WORDPIECE = 2, MODEL_WORDLEVEL = 3 };
enum PretokKind {
    PT_NONE = 0,
    PT_BYTELEVEL_GPT2 = 1,   // ByteLevel(use_regex=true)                       byte_level.rs:119-148
    PT_LLAMA3 = 2,           // Sequence[Split(a pattern of the tiktoken family: SplitRule below, Isolated), ByteLevel(use_regex=false)]
    PT_WHITESPACE = 3,       // \w+|[^\w\s]+                                    whitespace.rs:20-29
    PT_WHITESPACE_SPLIT = 4, // char::is_whitespace                             whitespace.rs:35-41
    PT_BERT = 5,             // BertPreTokenizer                                bert.rs:14-17
    PT_BYTELEVEL_NOREGEX = 6 // ByteLevel(use_regex=false): whole doc is one pre-token
};
enum NormKind { NORM_NONE = 0, NORM_BERT = 1 };

// ---- the tiktoken family of Split patterns (pre_tokenizers/split.rs:76-105 with a SysRegex, tokenizer/pattern.rs:63-83) -------
// Every member is the alternation
//   [contractions |] letters | digits |  ?[^\s\p{L}\p{N}]+ tail | \s*[\r\n]+ | \s+(?!\S) | \s+
// and differs from the Llama-3 / cl100k pattern in a handful of parameters (host_model.cpp parse_split_pattern is the reader; a pattern
// that does not reduce to them is refused):
struct SplitRule {
    uint8_t contr;       // 'S|'T|'RE|'VE|'M|'LL|'D: 0 absent; 1 first alternative, case-insensitive (?i:...) (Llama-3, Qwen2); 2 first alternative,
                         // case-sensitive; 3 optional SUFFIX of the letter alternatives, case-insensitive (o200k)
    uint8_t letters;     // 0: [^\r\n\p{L}\p{N}]?\p{L}+      2: the two case-split alternatives of o200k / tekken,
                         //    [^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+ | ...[\p{Lu}..]+[\p{Ll}..]*
    uint8_t digit_max;   // \p{N}{1,k}: k = 1 (\p{N}: Qwen2), 2, 3 (Llama-3); 0: \p{N}+
    uint8_t other_tail;  // what follows  ?[^\s\p{L}\p{N}]+ : 1 [\r\n]*, 2 [\r\n/]* (o200k)
};
constexpr SplitRule SPLIT_RULE_LLAMA3 = {1, 0, 3, 1};
// the rules the bit-parallel and the tile kernel implement (kernels/pretok_llama3.hip); the others run on the sequential matcher
TK_HD bool split_rule_fast(const SplitRule& r) { return r.letters == 0 && r.other_tail == 1 && r.contr <= 2; }
// the case-split rules (o200k, tekken): the bit-parallel kernel with l3_window_starts_cs, then the sequential matcher on the sentences that
// one left a byte of undecided (no tile tier)
TK_HD bool split_rule_fast_cs(const SplitRule& r) { return r.letters == 2 && (r.contr == 0 || r.contr == 3); }
// case classes of the case-split letters (generated data: unicode_case_ranges.inc)
enum : uint8_t { UCC_UPPER = 1 /* [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}] */, UCC_LOWER = 2 /* [\p{Ll}\p{Lm}\p{Lo}\p{M}] */ };

}  // namespace tkamd
