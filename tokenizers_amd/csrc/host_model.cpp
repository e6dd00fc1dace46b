// tokenizer.json -> flat tables.  See host_model.hpp.
#include "host_model.hpp"
#include "bert_norm_core.hpp"

#include <algorithm>
#include <cstring>
#include <functional>
#include <random>
#include <unordered_map>

#include "json.hpp"

namespace tkamd {

// Two-choice (cuckoo) placement of keys with hashes h1 into a table of mask + 1 slots (word_slot_a / word_slot_b, tables.hpp): a key
// goes to its slot a, else b, else it evicts the tenant of a, who moves to ITS other slot, and so on; a walk that does not end within
// the bound means an unlucky seed (at load factor <= 0.4 that is rare).  tenant[slot] = index of the key, or 0xFFFFFFFF.
bool cuckoo_place(const std::vector<uint32_t>& h1, uint32_t mask, std::vector<uint32_t>* tenant) {
    tenant->assign((size_t)mask + 1, 0xFFFFFFFFu);
    std::vector<uint32_t>& tn = *tenant;
    for (size_t i = 0; i < h1.size(); ++i) {
        uint32_t cur = (uint32_t)i, pos = word_slot_a(h1[cur], mask);
        if (tn[pos] != 0xFFFFFFFFu && tn[word_slot_b(h1[cur], mask)] == 0xFFFFFFFFu) pos = word_slot_b(h1[cur], mask);
        for (int kicks = 0;; ++kicks) {
            if (tn[pos] == 0xFFFFFFFFu) { tn[pos] = cur; break; }
            if (kicks == 512) return false;
            std::swap(cur, tn[pos]);                              // cur takes the slot; its tenant moves to ITS other slot
            pos = pos == word_slot_a(h1[cur], mask) ? word_slot_b(h1[cur], mask) : word_slot_a(h1[cur], mask);
        }
    }
    return true;
}

// Hash-and-displace (CHD) perfect hash: keys are bucketed by h1, buckets are placed largest first, each one
// searches the smallest 16-bit displacement d that drops all its keys into free slots (h2 + d * PH_MULT) & mask.
// Returns false if some bucket cannot be placed (caller retries with another seed / a larger table).
bool chd_place(const std::vector<uint32_t>& h1, const std::vector<uint32_t>& h2, uint32_t mask, uint32_t bmask,
               std::vector<uint16_t>* disp, std::vector<uint32_t>* slot_of_key) {
    const uint32_t nb = bmask + 1, n = (uint32_t)h1.size();
    std::vector<std::vector<uint32_t>> buckets(nb);
    for (uint32_t i = 0; i < n; ++i) buckets[h1[i] & bmask].push_back(i);
    std::vector<uint32_t> order(nb);
    for (uint32_t b = 0; b < nb; ++b) order[b] = b;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return buckets[x].size() > buckets[y].size(); });
    std::vector<uint8_t> used(mask + 1, 0);
    disp->assign(nb, 0);
    slot_of_key->assign(n, 0);
    std::vector<uint32_t> slots;
    for (uint32_t b : order) {
        const auto& keys = buckets[b];
        if (keys.empty()) break;
        bool placed = false;
        for (uint32_t d = 0; d < 65536 && !placed; ++d) {
            slots.clear();
            bool clash = false;
            for (uint32_t i : keys) {
                uint32_t sl = ph_slot(h2[i], d, mask);
                if (used[sl] || std::find(slots.begin(), slots.end(), sl) != slots.end()) { clash = true; break; }
                slots.push_back(sl);
            }
            if (!clash) {
                for (size_t k = 0; k < keys.size(); ++k) { used[slots[k]] = 1; (*slot_of_key)[keys[k]] = slots[k]; }
                (*disp)[b] = (uint16_t)d;
                placed = true;
            }
        }
        if (!placed) return false;
    }
    return true;
}

namespace {

struct UcRun {
    uint32_t first, last;
    uint8_t flags;
};
const UcRun kUcRuns[] = {
#include "unicode_ranges.inc"
};
const UcRun kUcCaseRuns[] = {      // (flags: UCC_UPPER / UCC_LOWER, tables.hpp)
#include "unicode_case_ranges.inc"
};

struct BnMapRow {
    uint32_t cp, a, b, c;
};
#define BN_WANT_RUNS
const UcRun kBnRuns[] = {
#include "bert_norm_tables.inc"
};
#undef BN_WANT_RUNS
#define BN_WANT_D
const BnMapRow kBnD[] = {
#include "bert_norm_tables.inc"
};
#undef BN_WANT_D
#define BN_WANT_LC
const BnMapRow kBnLC[] = {
#include "bert_norm_tables.inc"
};
#undef BN_WANT_LC
struct BnNfdRow {
    uint32_t cp, packed;
};
#define BN_WANT_NFD
const BnNfdRow kBnNFD[] = {
#include "bert_norm_tables.inc"
};
#undef BN_WANT_NFD

// GPT-2 bytes <-> unicode map, pre_tokenizers/byte_level.rs:15-39: printable bytes map to
// themselves, the other 68 bytes to U+0100+n in byte order.
void build_bytes_char(uint32_t b2c[256]) {
    bool direct[256] = {false};
    for (int b = '!'; b <= '~'; ++b) direct[b] = true;
    for (int b = 0xA1; b <= 0xAC; ++b) direct[b] = true;
    for (int b = 0xAE; b <= 0xFF; ++b) direct[b] = true;
    uint32_t n = 0;
    for (int b = 0; b < 256; ++b) {
        if (direct[b]) b2c[b] = (uint32_t)b;
        else b2c[b] = 256 + n++;
    }
}

// decode one UTF-8 scalar; returns length or 0 if malformed
int utf8_decode(const uint8_t* s, size_t n, uint32_t* cp) {
    if (n == 0) return 0;
    uint8_t b = s[0];
    if (b < 0x80) { *cp = b; return 1; }
    if ((b & 0xE0) == 0xC0 && n >= 2) { *cp = ((b & 0x1F) << 6) | (s[1] & 0x3F); return 2; }
    if ((b & 0xF0) == 0xE0 && n >= 3) { *cp = ((b & 0x0F) << 12) | ((s[1] & 0x3F) << 6) | (s[2] & 0x3F); return 3; }
    if ((b & 0xF8) == 0xF0 && n >= 4) {
        *cp = ((b & 0x07) << 18) | ((s[1] & 0x3F) << 12) | ((s[2] & 0x3F) << 6) | (s[3] & 0x3F);
        return 4;
    }
    return 0;
}

// token string in the byte-level alphabet -> raw bytes; false if a char is outside the alphabet
bool bytelevel_to_raw(const std::string& tok, const std::unordered_map<uint32_t, uint8_t>& c2b, std::string* raw) {
    raw->clear();
    const uint8_t* s = (const uint8_t*)tok.data();
    size_t n = tok.size(), i = 0;
    while (i < n) {
        uint32_t cp;
        int l = utf8_decode(s + i, n - i, &cp);
        if (!l) return false;
        auto it = c2b.find(cp);
        if (it == c2b.end()) return false;
        raw->push_back((char)it->second);
        i += l;
    }
    return true;
}

void build_unicode(HostModel& m) {
    std::vector<uint8_t> flat(0x110000, 0);
    for (const UcRun& r : kUcRuns)
        for (uint32_t cp = r.first; cp <= r.last; ++cp) flat[cp] = r.flags;
    m.uc_stage1.assign(UC_STAGE1_LEN, 0);
    m.uc_stage2.clear();
    std::unordered_map<std::string, uint16_t> seen;
    for (uint32_t blk = 0; blk < (uint32_t)UC_STAGE1_LEN; ++blk) {
        std::string key((const char*)&flat[blk * 256], 256);
        auto it = seen.find(key);
        if (it == seen.end()) {
            uint16_t idx = (uint16_t)(m.uc_stage2.size() / 256);
            m.uc_stage2.insert(m.uc_stage2.end(), key.begin(), key.end());
            it = seen.emplace(std::move(key), idx).first;
        }
        m.uc_stage1[blk] = it->second;
    }
}

void two_stage(const std::vector<uint8_t>& flat, std::vector<uint16_t>* s1, std::vector<uint8_t>* s2) {
    s1->assign(UC_STAGE1_LEN, 0);
    s2->clear();
    std::unordered_map<std::string, uint16_t> seen;
    for (uint32_t blk = 0; blk < (uint32_t)UC_STAGE1_LEN; ++blk) {
        std::string key((const char*)&flat[blk * 256], 256);
        auto it = seen.find(key);
        if (it == seen.end()) {
            uint16_t idx = (uint16_t)(s2->size() / 256);
            s2->insert(s2->end(), key.begin(), key.end());
            it = seen.emplace(std::move(key), idx).first;
        }
        (*s1)[blk] = it->second;
    }
}

template <class Slot, class IsEmpty, class H1, class H2>
bool cuckoo_insert(std::vector<Slot>& tab, Slot item, IsEmpty is_empty, H1 h1, H2 h2, std::mt19937& rng) {
    for (int kick = 0; kick < 2000; ++kick) {
        uint32_t s1 = h1(item), s2 = h2(item);
        if (is_empty(tab[s1])) { tab[s1] = item; return true; }
        if (is_empty(tab[s2])) { tab[s2] = item; return true; }
        uint32_t victim = (rng() & 1) ? s1 : s2;
        std::swap(item, tab[victim]);
    }
    return false;
}

void build_pair_table(const std::vector<MergeSlot>& items, std::vector<MergeSlot>* out, uint32_t* out_mask, uint32_t* out_seed) {
    if (items.empty()) { out->assign(16, MergeSlot{MERGE_EMPTY, MERGE_EMPTY, RANK_NONE, 0}); *out_mask = 15; *out_seed = 0; return; }
    uint32_t cap = 16;
    while (cap < items.size() * 5 / 2) cap <<= 1;   // load factor <= 0.4
    std::mt19937 rng(12345);
    for (int attempt = 0; attempt < 64; ++attempt) {
        uint32_t seed = (uint32_t)rng();
        uint32_t mask = cap - 1;
        std::vector<MergeSlot> tab(cap, MergeSlot{MERGE_EMPTY, MERGE_EMPTY, RANK_NONE, 0});
        bool ok = true;
        for (const MergeSlot& e : items) {
            ok = cuckoo_insert(
                tab, e, [](const MergeSlot& s) { return s.a == MERGE_EMPTY; },
                [&](const MergeSlot& s) { return merge_hash1(s.a, s.b, seed) & mask; },
                [&](const MergeSlot& s) { return merge_hash2(s.a, s.b, seed) & mask; }, rng);
            if (!ok) break;
        }
        if (ok) { out->swap(tab); *out_mask = mask; *out_seed = seed; return; }
        if (attempt % 4 == 3) cap <<= 1;
    }
    throw Invalid("could not build a pair hash table");
}

void build_merge_table(HostModel& m, const std::vector<MergeSlot>& merges) {
    // new_id == rank + constant for every merge (true of every trainer-produced vocabulary: merged tokens are
    // appended in merge order): the LDS-resident merge kernel then needs no per-pair new-id array
    m.merge_newid_affine = !merges.empty();
    m.merge_newid_base = merges.empty() ? 0u : merges[0].new_id - merges[0].rank;
    for (const MergeSlot& e : merges)
        if (e.new_id - e.rank != m.merge_newid_base) { m.merge_newid_affine = false; break; }
    uint32_t cap = 16;
    while (cap < merges.size() * 5 / 2) cap <<= 1;            // load factor <= 0.4
    uint32_t nb_wide = 16;
    while (nb_wide < merges.size() / 4) nb_wide <<= 1;        // 2-4 keys per bucket (50k merges -> 16384 buckets = 32 KB)
    // The merge kernels keep up to DISP_LDS_MAX displacements in LDS; beyond that every probe of the merge chain reads its
    // displacement from global memory first -- two dependent round trips instead of one (C4's 128 k merges: 32,768 buckets,
    // the merge kernels 0.17 ms against C2's 0.10).  More keys per bucket (8 at 128 k) only cost the builder trials: at load
    // factor <= 0.4 a bucket of k keys fits a given displacement with probability >= 0.6^k, and 16 bits of displacement are 65,536 tries.
    // (TKAMD_MERGE_BUCKETS=wide: the old sizing -- the tests' way to the global-displacement branch.)
    const char* wide_env = getenv("TKAMD_MERGE_BUCKETS");
    const bool want_wide = wide_env && !strcmp(wide_env, "wide");
    std::mt19937 rng(777);
    for (int attempt = 0; attempt < 32; ++attempt) {
        const uint32_t nb = (want_wide || attempt >= 8) ? nb_wide : std::min(nb_wide, (uint32_t)DISP_LDS_MAX);
        const uint32_t seed = (uint32_t)rng();
        std::vector<uint32_t> h1(merges.size()), h2(merges.size()), where;
        for (size_t i = 0; i < merges.size(); ++i) { h1[i] = merge_hash1(merges[i].a, merges[i].b, seed); h2[i] = merge_hash2(merges[i].a, merges[i].b, seed); }
        std::vector<uint16_t> disp;
        if (chd_place(h1, h2, cap - 1, nb - 1, &disp, &where)) {
            m.merge_table.assign(cap, MergeSlot{MERGE_EMPTY, MERGE_EMPTY, RANK_NONE, 0});
            for (size_t i = 0; i < merges.size(); ++i) m.merge_table[where[i]] = merges[i];
            m.merge_disp.swap(disp);
            m.merge_mask = cap - 1; m.merge_bmask = nb - 1; m.merge_seed = seed;
            return;
        }
        if (attempt % 8 == 7) cap <<= 1;                       // a new seed almost always does it; grow (at most 8x) only as a last resort
    }
    throw Invalid("could not build the merge hash table");
}

void build_word_table(HostModel& m, const std::vector<WordSlot>& words) {
    uint32_t cap = 16;
    while (cap < words.size() * 5 / 2) cap <<= 1;
    std::mt19937 rng(54321);
    for (int attempt = 0; attempt < 32; ++attempt) {
        const uint32_t seed = (uint32_t)rng();
        std::vector<uint32_t> h1(words.size()), tenant;
        for (size_t i = 0; i < words.size(); ++i) h1[i] = word_hash1(words[i].lo, words[i].hi, words[i].len, seed);
        if (cuckoo_place(h1, cap - 1, &tenant)) {
            m.word_table.assign(cap, WordSlot{0, 0, 0, 0, 0, 0});
            for (uint32_t sidx = 0; sidx < cap; ++sidx)
                if (tenant[sidx] != 0xFFFFFFFFu) m.word_table[sidx] = words[tenant[sidx]];
            m.word_mask = cap - 1; m.word_seed = seed;
            return;
        }
        if (attempt % 8 == 7) cap <<= 1;                          // a new seed almost always does it; grow only as a last resort
    }
    throw Invalid("could not build the whole-word hash table");
}

void build_long_table(HostModel& m) {
    size_t n = m.long_id.size();
    uint32_t cap = 16;
    while (cap < n * 2 + 1) cap <<= 1;
    m.long_mask = cap - 1;
    m.long_table.assign(cap, 0);
    for (size_t e = 0; e < n; ++e) {
        uint32_t h = fnv1a(&m.long_blob[m.long_off[e]], m.long_off[e + 1] - m.long_off[e]) & m.long_mask;
        while (m.long_table[h]) h = (h + 1) & m.long_mask;
        m.long_table[h] = (uint32_t)e + 1;
    }
}

// Build the two-root byte trie for WordPiece and flatten it into a cuckoo table.
void build_trie(HostModel& m, const std::vector<std::pair<std::string, uint32_t>>& initial,
                const std::vector<std::pair<std::string, uint32_t>>& cont) {
    struct Node {
        std::unordered_map<uint8_t, uint32_t> kids;
        uint32_t id = 0xFFFFFFFFu;
    };
    std::vector<Node> nodes(2);
    auto insert = [&](uint32_t root, const std::string& key, uint32_t id) {
        uint32_t cur = root;
        for (unsigned char c : key) {
            auto it = nodes[cur].kids.find(c);
            uint32_t nxt;
            if (it == nodes[cur].kids.end()) {
                nxt = (uint32_t)nodes.size();
                nodes.emplace_back();
                nodes[cur].kids.emplace(c, nxt);
            } else nxt = it->second;
            cur = nxt;
        }
        nodes[cur].id = id;
    };
    for (auto& kv : initial) insert(0, kv.first, kv.second);
    for (auto& kv : cont) insert(1, kv.first, kv.second);
    std::vector<MergeSlot> items;
    for (uint32_t n = 0; n < nodes.size(); ++n)
        for (auto& e : nodes[n].kids) items.push_back(MergeSlot{n, (uint32_t)e.first, e.second, nodes[e.second].id});
    if (nodes.size() >= ((size_t)1 << 24)) throw Unsupported("WordPiece trie beyond 2^24 nodes");
    m.trie.n_nodes = (uint32_t)nodes.size();
    build_pair_table(items, &m.trie.table, &m.trie.mask, &m.trie.seed);
}

void build_pair_table(const std::vector<MergeSlot>& items, std::vector<MergeSlot>* out, uint32_t* out_mask, uint32_t* out_seed);

// BertNormalizer data: flags as a 2-stage table, the NFD+strip (kind 0) and lowercase (kind 1) maps as one
// cuckoo table keyed (cp, kind) with the up-to-3 output code points packed 21 bits each
void build_bert_norm(HostModel& m) {
    std::vector<uint8_t> flat(0x110000, 0);
    for (const UcRun& r : kBnRuns)
        for (uint32_t cp = r.first; cp <= r.last; ++cp) flat[cp] = r.flags;
    two_stage(flat, &m.bn_stage1, &m.bn_stage2);
    std::vector<MergeSlot> items;
    auto add = [&](const BnMapRow& r, uint32_t kind) {
        uint64_t v = (uint64_t)r.a | ((uint64_t)r.b << 21) | ((uint64_t)r.c << 42);
        items.push_back(MergeSlot{r.cp, kind, (uint32_t)v, (uint32_t)(v >> 32)});
    };
    for (const BnMapRow& r : kBnD) add(r, 0);
    for (const BnMapRow& r : kBnLC) add(r, 1);
    for (const BnNfdRow& r : kBnNFD) items.push_back(MergeSlot{r.cp, 2u, r.packed, 0u});      // kind 2: classes of the NFD pieces (bert_norm_core.hpp)
    build_pair_table(items, &m.bn_map, &m.bn_mask, &m.bn_seed);
}

// The top-level alternatives of a regex source: cut at every '|' outside (...) and [...]; a backslash escapes the next character.
std::vector<std::string> regex_alternatives(const std::string& rx) {
    std::vector<std::string> out(1);
    int depth = 0;
    bool cls = false;
    for (size_t i = 0; i < rx.size(); ++i) {
        const char c = rx[i];
        if (c == '\\' && i + 1 < rx.size()) { out.back() += c; out.back() += rx[++i]; continue; }
        if (cls) { if (c == ']') cls = false; }
        else if (c == '[') cls = true;
        else if (c == '(') ++depth;
        else if (c == ')') --depth;
        else if (c == '|' && depth == 0) { out.emplace_back(); continue; }
        out.back() += c;
    }
    return out;
}

// Split(Regex(pattern), Isolated) of the tiktoken family -> the parameters of tables.hpp SplitRule (pre_tokenizers/split.rs:76-105 hands
// the pattern to Oniguruma, tokenizer/pattern.rs:63-83; here the pattern is READ, alternative by alternative, and everything that is not
// one of the family's spellings is refused).  *gpt2: the pattern is the GPT-2 regex itself (byte_level.rs:43-46), served by that kernel.
bool parse_split_pattern(const std::string& rx, SplitRule* rule, bool* gpt2, std::string* why) {
    *gpt2 = rx == "'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";
    if (*gpt2) return true;
    const std::vector<std::string> alt = regex_alternatives(rx);
    size_t k = 0;
    SplitRule r{0, 0, 3, 1};
    static const char* const lits[7] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
    const std::string ci = "(?i:'s|'t|'re|'ve|'m|'ll|'d)";
    auto at = [&](size_t i) -> const std::string& { static const std::string none; return i < alt.size() ? alt[i] : none; };
    if (at(k) == ci || at(k) == "'(?i:[sdmt]|ll|ve|re)") { r.contr = 1; ++k; }
    else {
        bool all = alt.size() >= 7;
        for (size_t q = 0; q < 7 && all; ++q) all = alt[q] == lits[q];
        if (all) { r.contr = 2; k = 7; }
    }
    const std::string pre = "[^\\r\\n\\p{L}\\p{N}]?", up = "[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]", lo = "[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]";
    if (at(k) == pre + "\\p{L}+") { r.letters = 0; ++k; }
    else if ((at(k) == pre + up + "*" + lo + "+" && at(k + 1) == pre + up + "+" + lo + "*") ||
             (at(k) == pre + up + "*" + lo + "+" + ci + "?" && at(k + 1) == pre + up + "+" + lo + "*" + ci + "?")) {
        if (at(k).size() > (pre + up + "*" + lo + "+").size()) {
            if (r.contr) { *why = "contractions both as an alternative and as a suffix of the letter alternatives"; return false; }
            r.contr = 3;
        }
        r.letters = 2;
        k += 2;
    } else { *why = "letter alternative '" + at(k) + "'"; return false; }
    if (at(k) == "\\p{N}{1,3}") r.digit_max = 3;
    else if (at(k) == "\\p{N}{1,2}") r.digit_max = 2;
    else if (at(k) == "\\p{N}" || at(k) == "\\p{N}{1}" || at(k) == "\\p{N}{1,1}") r.digit_max = 1;
    else if (at(k) == "\\p{N}+") r.digit_max = 0;
    else { *why = "digit alternative '" + at(k) + "'"; return false; }
    ++k;
    if (at(k) == " ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*") r.other_tail = 1;
    else if (at(k) == " ?[^\\s\\p{L}\\p{N}]+[\\r\\n/]*") r.other_tail = 2;
    else { *why = "alternative '" + at(k) + "'"; return false; }
    ++k;
    if (at(k) != "\\s*[\\r\\n]+" || at(k + 1) != "\\s+(?!\\S)" || at(k + 2) != "\\s+" || alt.size() != k + 3) {
        *why = "the whitespace alternatives are not \\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
        return false;
    }
    *rule = r;
    return true;
}

PretokKind parse_pretok(const JsonValue* pt, HostModel& m) {
    if (!pt || pt->is_null()) throw Unsupported("pre_tokenizer: null is outside the hot path");
    std::string type = pt->get_str("type");
    if (type == "ByteLevel") {
        m.byte_level = true;
        m.add_prefix_space = pt->get_bool("add_prefix_space", true);
        bool use_regex = pt->get_bool("use_regex", true);
        return use_regex ? PT_BYTELEVEL_GPT2 : PT_BYTELEVEL_NOREGEX;
    }
    if (type == "Whitespace") return PT_WHITESPACE;
    if (type == "WhitespaceSplit") return PT_WHITESPACE_SPLIT;
    if (type == "BertPreTokenizer") return PT_BERT;
    if (type == "Sequence") {
        const JsonValue* seq = pt->get("pretokenizers");
        if (seq && seq->is_array() && seq->arr.size() == 2) {
            const JsonValue* a = seq->arr[0].get();
            const JsonValue* b = seq->arr[1].get();
            if (a->get_str("type") == "Split" && b->get_str("type") == "ByteLevel") {
                const JsonValue* pat = a->get("pattern");
                std::string rx = pat ? pat->get_str("Regex") : "";
                bool invert = a->get_bool("invert", false);
                std::string beh = a->get_str("behavior");
                if (rx.empty()) throw Unsupported("pre_tokenizer: Split with a String pattern (only Regex patterns of the tiktoken family are on the path)");
                if (invert || beh != "Isolated") throw Unsupported("pre_tokenizer: Split with behavior '" + beh + "'" + (invert ? " inverted" : "") + " (only Isolated is on the path)");
                if (b->get_bool("use_regex", true)) throw Unsupported("pre_tokenizer: Sequence[Split, ByteLevel(use_regex=true)] applies two regexes");
                // ByteLevel::pre_tokenize puts its prefix space in front of every SPLIT it is handed (byte_level.rs:122-125) -- behind a Split
                // that is every pre-token, not every document: no tokenizer in use is configured that way, and the path does not build it
                if (b->get_bool("add_prefix_space", true))
                    throw Unsupported("pre_tokenizer: Sequence[Split, ByteLevel(add_prefix_space=true)] (a prefix space in front of every pre-token)");
                SplitRule rule{};
                bool gpt2 = false;
                std::string why;
                if (!parse_split_pattern(rx, &rule, &gpt2, &why))
                    throw Unsupported("pre_tokenizer: Split pattern outside the tiktoken family (" + why + ")");
                m.byte_level = true;
                m.add_prefix_space = false;
                if (gpt2) return PT_BYTELEVEL_GPT2;
                m.split_rule = rule;
                return PT_LLAMA3;
            }
        }
        throw Unsupported("pre_tokenizer: this Sequence is outside the hot path");
    }
    throw Unsupported("pre_tokenizer: type '" + type + "' is outside the hot path");
}

}  // namespace

// hash of a long vocabulary key (tables.hpp: long_key_hash_*); the name is historical
uint32_t fnv1a(const uint8_t* p, size_t n) {
    uint32_t h = long_key_hash_init((uint32_t)n);
    for (size_t i = 0; i < n; i += 4) {
        uint32_t w = 0;
        for (size_t q = 0; q < 4 && i + q < n; ++q) w |= (uint32_t)p[i + q] << (8 * q);
        h = long_key_hash_step(h, w);
    }
    return h;
}

// WordPiece decoder cleanup (decoders/wordpiece.rs:31-44): eleven literal replacements applied in order
static std::string wp_cleanup(std::string t) {
    static const char* const rules[][2] = {{" .", "."}, {" ?", "?"}, {" !", "!"}, {" ,", ","}, {" ' ", "'"}, {" n't", "n't"},
                                           {" 'm", "'m"}, {" do not", " don't"}, {" 's", "'s"}, {" 've", "'ve"}, {" 're", "'re"}};
    for (auto& r : rules) {
        const std::string from = r[0], to = r[1];
        std::string out;
        size_t pos = 0;
        for (;;) {
            size_t hit = t.find(from, pos);
            if (hit == std::string::npos) { out.append(t, pos, std::string::npos); break; }
            out.append(t, pos, hit - pos);
            out += to;
            pos = hit + from.size();
        }
        t.swap(out);
    }
    return t;
}

// decode_batch tables.  Tokenizer::decode (tokenizer/mod.rs:935-953) maps every id to a token string (added
// vocabulary first, then the model; unknown ids vanish), drops specials on request and hands the strings to the
// decoder, whose result is joined.  For the decoders on this path the contribution of a token depends only on its id
// and on whether it is the first kept token of its sequence:
//   ByteLevel (pre_tokenizers/byte_level.rs:155-171)  bytes of the token through the inverse byte alphabet, or the
//                                                     token's own UTF-8 if a char is outside the alphabet
//   WordPiece (decoders/wordpiece.rs:46-64)           first: token; later: token minus the prefix, or " " + token;
//                                                     then cleanup()
//   none      (mod.rs:950-952)                        tokens.join(" ")
// so decoding is a gather of precomputed byte strings.
static void build_decode_tables(HostModel& m, const JsonValue* root, const std::unordered_map<std::string, uint32_t>& vocab,
                                const std::unordered_map<uint32_t, uint8_t>& c2b) {
    const JsonValue* dec = root->get("decoder");
    std::string prefix = "##", suffix = "</w>", ctc_pad, ctc_delim, post_strip;
    bool cleanup = true, chain_bytes = false;
    struct DecEl { int kind; std::string a, b; int64_t start, stop; };      // 0: Replace a -> b; 1: Strip a (one char), start / stop
    std::vector<DecEl> chain;
    auto replace_lit = [](const std::string& t, const std::string& from, const std::string& to) {
        std::string out;
        size_t pos = 0;
        for (;;) {
            const size_t hit = t.find(from, pos);
            if (hit == std::string::npos) { out.append(t, pos, std::string::npos); break; }
            out.append(t, pos, hit - pos);
            out += to;
            pos = hit + from.size();
        }
        return out;
    };
    // chars of a UTF-8 string as byte ranges (Strip counts chars)
    auto char_starts = [](const std::string& t) {
        std::vector<size_t> st;
        for (size_t i = 0; i < t.size(); ++i) if (((uint8_t)t[i] & 0xC0u) != 0x80u) st.push_back(i);
        st.push_back(t.size());
        return st;
    };
    auto strip = [&](const std::string& t, const std::string& c, int64_t start, int64_t stop) {
        const std::vector<size_t> cs = char_starts(t);
        const size_t n = cs.size() - 1;
        auto is_c = [&](size_t k) { return t.compare(cs[k], cs[k + 1] - cs[k], c) == 0; };
        size_t a = 0, b = n;
        for (size_t k = 0; k < n && (int64_t)k < start; ++k) { if (!is_c(k)) break; a = k + 1; }
        for (int64_t k = 0; k < stop && (size_t)k < n; ++k) { const size_t j = n - (size_t)k - 1; if (!is_c(j)) break; b = j; }
        return b >= a ? t.substr(cs[a], cs[b] - cs[a]) : std::string();
    };
    if (!dec || dec->is_null()) m.decoder = DEC_JOIN_SPACE;
    else {
        const std::string t = dec->get_str("type");
        if (t == "ByteLevel") m.decoder = DEC_BYTELEVEL;
        else if (t == "WordPiece") {
            m.decoder = DEC_WORDPIECE;
            prefix = dec->get_str("prefix", "##");
            cleanup = dec->get_bool("cleanup", true);
        } else if (t == "BPEDecoder") {
            m.decoder = DEC_BPE;
            suffix = dec->get_str("suffix", "</w>");
            if (suffix.empty()) {        // (str::replace("", " ") puts a space between all chars: not a shape worth tables)
                m.decoder = DEC_UNSUPPORTED;
                m.dec_unsupported = "BPEDecoder with an empty suffix is outside the decode path";
                return;
            }
        } else if (t == "ByteFallback") m.decoder = DEC_BYTE_FALLBACK;
        else if (t == "Fuse") m.decoder = DEC_FUSE;
        else if (t == "CTC") {
            // decoders/ctc.rs:45-63: consecutive equal tokens collapse (on the device: an id equal to the kept id in front of it), then per
            // token: pad_token -> "", cleanup() and word_delimiter_token -> " "; a token that ends up empty contributes nothing
            m.decoder = DEC_CTC;
            m.dec_dedup = true;
            ctc_pad = dec->get_str("pad_token", "<pad>");
            ctc_delim = dec->get_str("word_delimiter_token", "|");
            cleanup = dec->get_bool("cleanup", true);
        } else if (t == "Sequence" || t == "Replace" || t == "Strip") {
            // A chain (decoders/sequence.rs:26-33: every member's decode_chain in turn; Decoder::decode joins the result with "",
            // tokenizer/mod.rs:184-187) of the per-token members -- Replace with a literal pattern (normalizers/replace.rs:88-106), Strip
            // (decoders/strip.rs:27-60) --, then ByteFallback, then Fuse, then Strip { start <= 1, stop 0 }: what the SentencePiece-style
            // tokenizers carry ([Replace("\u2581", " "), ByteFallback, Fuse, Strip(" ", 1, 0)]).  Behind Fuse there is ONE string, so that Strip
            // only ever looks at the first kept token -- its first-position form.  Anything else is refused.
            std::vector<const JsonValue*> members;
            if (t == "Sequence") {
                const JsonValue* ds = dec->get("decoders");
                if (ds && ds->is_array()) for (const auto& d : ds->arr) members.push_back(d.get());
            } else members.push_back(dec);
            int stage = 0;                      // 0: per-token members, 1: behind ByteFallback, 2: behind Fuse, 3: behind the leading Strip
            auto refuse = [&](const std::string& why) { m.decoder = DEC_UNSUPPORTED; m.dec_unsupported = why; };
            m.decoder = DEC_CHAIN;
            for (const JsonValue* d : members) {
                const std::string k = d->get_str("type");
                if (k == "Replace" && stage == 0) {
                    const JsonValue* pat = d->get("pattern");
                    const std::string lit = pat ? pat->get_str("String") : "";
                    if (lit.empty()) { refuse("a Replace decoder with a Regex (or empty) pattern is outside the decode path"); return; }
                    chain.push_back(DecEl{0, lit, d->get_str("content"), 0, 0});
                } else if (k == "Strip" && (stage == 0 || stage == 2)) {
                    const std::string c = d->get_str("content");
                    const int64_t a = (int64_t)d->get_num("start", 0), b = (int64_t)d->get_num("stop", 0);
                    if (c.empty() || a < 0 || b < 0) { refuse("a Strip decoder without a content char"); return; }
                    if (stage == 2) {
                        if (a > 1 || b != 0) { refuse("a Strip decoder behind Fuse with start > 1 or stop > 0 is outside the decode path"); return; }
                        if (a == 1) { post_strip = c; stage = 3; }
                    } else chain.push_back(DecEl{1, c, "", a, b});
                } else if (k == "ByteFallback" && stage == 0) { chain_bytes = true; stage = 1; }
                else if (k == "Fuse" && stage <= 1) stage = 2;
                else { refuse("decoder '" + k + "' at this place of a decoder Sequence is outside the decode path"); return; }
            }
            if (chain_bytes && !post_strip.empty() && (uint8_t)post_strip[0] >= 0x80u) { refuse("a non-ASCII Strip behind ByteFallback is outside the decode path"); return; }
        } else {
            m.decoder = DEC_UNSUPPORTED;
            m.dec_unsupported = "decoder type '" + t + "' is outside the decode path";
            return;
        }
    }
    // id -> token string: model vocabulary, overridden by the added vocabulary (added_vocabulary.rs:239-246)
    uint32_t n_ids = 0;
    for (auto& kv : vocab) n_ids = std::max(n_ids, kv.second + 1);
    for (const AddedToken& a : m.added_tokens) n_ids = std::max(n_ids, a.id + 1);
    if (n_ids > (1u << 26)) { m.decoder = DEC_UNSUPPORTED; m.dec_unsupported = "token ids beyond 2^26"; return; }
    std::vector<const std::string*> tok(n_ids, nullptr);
    std::vector<uint8_t> special(n_ids, 0);
    for (auto& kv : vocab) tok[kv.second] = &kv.first;
    std::unordered_map<std::string, bool> special_set;
    for (const AddedToken& a : m.added_tokens) {
        if (a.normalized && m.norm != NORM_NONE) {
            m.decoder = DEC_UNSUPPORTED;
            m.dec_unsupported = "added token '" + a.content + "' is normalized behind a normalizer (its decoded form is version dependent)";
            return;
        }
        tok[a.id] = &a.content;
        if (a.special) special_set[a.content] = true;
    }
    for (uint32_t id = 0; id < n_ids; ++id)
        if (tok[id] && special_set.count(*tok[id])) special[id] = 1;     // is_special_token tests the STRING (added_vocabulary.rs:258-260)
    m.dec_entry.assign((size_t)n_ids * 4, 0);
    m.dec_position_dependent = false;
    auto put = [&](const std::string& bytes) -> std::pair<uint32_t, uint32_t> {
        uint32_t off = (uint32_t)m.dec_blob.size();
        m.dec_blob.insert(m.dec_blob.end(), bytes.begin(), bytes.end());
        return {off, (uint32_t)bytes.size()};
    };
    for (uint32_t id = 0; id < n_ids; ++id) {
        uint32_t* e = &m.dec_entry[(size_t)id * 4];
        if (!tok[id]) { e[1] = DEC_ABSENT; continue; }
        const std::string& t = *tok[id];
        std::string first, rest;
        if (m.decoder == DEC_BYTELEVEL) {
            if (!bytelevel_to_raw(t, c2b, &first)) first = t;
            rest = first;
        } else if (m.decoder == DEC_WORDPIECE) {
            first = t;
            if (!prefix.empty() && t.compare(0, prefix.size(), prefix) == 0) rest = t.substr(prefix.size());
            else if (prefix.empty()) rest = t;                 // strip_prefix("") always succeeds
            else rest = " " + t;
            if (cleanup) { first = wp_cleanup(first); rest = wp_cleanup(rest); }
        } else if (m.decoder == DEC_BPE) {
            // token.replace(suffix, " "), and token.replace(suffix, "") on the LAST token (decoders/bpe.rs:30-37): `first` is the form of
            // the position that has one of its own -- here the last kept token (dec_special_is_last)
            auto replace_all = [&](const std::string& with) {
                std::string out;
                size_t pos = 0;
                for (;;) {
                    const size_t hit = t.find(suffix, pos);
                    if (hit == std::string::npos) { out.append(t, pos, std::string::npos); break; }
                    out.append(t, pos, hit - pos);
                    out += with;
                    pos = hit + suffix.size();
                }
                return out;
            };
            first = replace_all("");
            rest = replace_all(" ");
        } else if (m.decoder == DEC_CTC) {
            first = replace_lit(t, ctc_pad, "");
            if (ctc_pad.empty()) first = t;
            if (cleanup) { first = wp_cleanup(first); if (!ctc_delim.empty()) first = replace_lit(first, ctc_delim, " "); }
            rest = first;
        } else if (m.decoder == DEC_FUSE || m.decoder == DEC_BYTE_FALLBACK || m.decoder == DEC_CHAIN) {
            std::string u = t;
            for (const DecEl& el : chain) u = el.kind == 0 ? replace_lit(u, el.a, el.b) : strip(u, el.a, el.start, el.stop);
            const std::string& t = u;                       // (what ByteFallback / Fuse see)
            first = rest = t;
            if (!post_strip.empty()) {
                // Strip(c, 1, 0) behind Fuse: one leading c off the FUSED string -- off the first kept token.  (An empty token would hand the
                // strip on to its successor: no table says that.)
                if (t.empty() && !special[id]) { m.decoder = DEC_UNSUPPORTED; m.dec_unsupported = "a token that decodes to nothing in front of a Strip behind Fuse"; return; }
                first = strip(t, post_strip, 1, 0);
            }
            const bool bytes_on = m.decoder == DEC_BYTE_FALLBACK || chain_bytes;
            // <0xXX>: six bytes, "<0x", two hex digits as u8::from_str_radix reads them (a leading '+' counts as a digit's place), ">"
            if (bytes_on && t.size() == 6 && t.compare(0, 3, "<0x") == 0 && t[5] == '>') {
                auto hex = [](char c) -> int { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
                int v = -1;
                if (t[3] == '+') { const int lo = hex(t[4]); if (lo >= 0) v = lo; }
                else { const int hi = hex(t[3]), lo = hex(t[4]); if (hi >= 0 && lo >= 0) v = hi * 16 + lo; }
                if (v >= 0) {
                    // (as the first kept token under a leading Strip of this very byte: nothing -- unless its run is not UTF-8, which the
                    // device decides: the length of the first-position form is 0 then)
                    const bool gone_first = !post_strip.empty() && post_strip.size() == 1 && (uint8_t)post_strip[0] == (uint8_t)v;
                    if (gone_first) m.dec_position_dependent = true;
                    e[0] = (uint32_t)v;
                    e[1] = (gone_first ? 0u : 1u) | DEC_BYTE | (special[id] ? DEC_SPECIAL : 0u);
                    e[2] = (uint32_t)v;
                    e[3] = 1u;
                    m.dec_has_bytes = true;
                    continue;
                }
            }
        } else {
            first = t;
            rest = " " + t;
        }
        if (first.size() > DEC_LEN_MASK || rest.size() > DEC_LEN_MASK) { m.decoder = DEC_UNSUPPORTED; m.dec_unsupported = "token too long"; return; }
        auto a = put(first);
        e[0] = a.first;
        e[1] = a.second | (special[id] ? DEC_SPECIAL : 0u);
        if (rest == first) { e[2] = a.first; e[3] = a.second; }
        else { auto b = put(rest); e[2] = b.first; e[3] = b.second; m.dec_position_dependent = true; }
    }
    m.dec_special_is_last = m.decoder == DEC_BPE;
    m.dec_blob.resize(m.dec_blob.size() + 16, 0);                        // readable slack for vector loads
}

HostModel HostModel::from_json(const char* json, size_t len) {
    JsonPtr root;
    try {
        root = json_parse(json, len);
    } catch (const std::exception& e) {
        throw Invalid(e.what());
    }
    if (!root->is_object()) throw Invalid("tokenizer.json: top level is not an object");
    HostModel m;
    std::fill(m.byte_id, m.byte_id + 256, 0xFFFFFFFFu);

    // ---- truncation / padding: an epilogue over the finished token CSR (tokenizer/mod.rs:1265-1317) ----
    const JsonValue* trunc = root->get("truncation");
    if (trunc && !trunc->is_null()) {
        m.trunc_on = true;
        m.trunc_max_length = (uint32_t)trunc->get_num("max_length", 512);
        m.trunc_stride = (uint32_t)trunc->get_num("stride", 0);
        m.trunc_left = trunc->get_str("direction", "Right") == "Left";
        const std::string st = trunc->get_str("strategy", "LongestFirst");
        if (st == "LongestFirst") m.trunc_strategy = 0;
        else if (st == "OnlyFirst") m.trunc_strategy = 1;
        else if (st == "OnlySecond") m.trunc_strategy = 2;
        else throw Invalid("tokenizer.json: unknown truncation strategy '" + st + "'");
    }
    const JsonValue* pad = root->get("padding");
    if (pad && !pad->is_null()) {
        m.pad_on = true;
        const JsonValue* ps = pad->get("strategy");
        if (ps && ps->is_object() && ps->get("Fixed")) { m.pad_fixed = true; m.pad_length = (uint32_t)ps->get_num("Fixed", 0); }
        else if (ps && ps->is_string() && ps->str == "BatchLongest") m.pad_fixed = false;
        else throw Invalid("tokenizer.json: bad padding strategy");
        m.pad_left = pad->get_str("direction", "Right") == "Left";
        const JsonValue* mult = pad->get("pad_to_multiple_of");
        m.pad_multiple = (mult && mult->is_number()) ? (uint32_t)mult->num : 0u;
        m.pad_id = (uint32_t)pad->get_num("pad_id", 0);
        m.pad_type_id = (uint32_t)pad->get_num("pad_type_id", 0);
        m.pad_token = pad->get_str("pad_token", "[PAD]");
        if (m.pad_id >= (1u << 24)) throw Unsupported("padding id beyond 2^24");
    }

    // ---- normalizer ----
    const JsonValue* norm = root->get("normalizer");
    if (norm && !norm->is_null()) {
        std::string t = norm->get_str("type");
        if (t == "BertNormalizer") {
            m.norm = NORM_BERT;
            m.bn_clean_text = norm->get_bool("clean_text", true);
            m.bn_handle_chinese = norm->get_bool("handle_chinese_chars", true);
            m.bn_lowercase = norm->get_bool("lowercase", true);
            const JsonValue* sa = norm->get("strip_accents");
            m.bn_strip_accents = (sa && sa->is_bool()) ? sa->b : m.bn_lowercase;  // normalizers/bert.rs:124
        } else {
            throw Unsupported("normalizer: type '" + t + "' is outside the hot path");
        }
    }

    // ---- pre-tokenizer ----
    m.pretok = parse_pretok(root->get("pre_tokenizer"), m);
    if (m.pretok == PT_LLAMA3 && m.split_rule.letters == 2) {         // the case classes of the case-split letter alternatives
        std::vector<uint8_t> flat(0x110000, 0);
        for (const UcRun& r : kUcCaseRuns)
            for (uint32_t cp = r.first; cp <= r.last; ++cp) flat[cp] = r.flags;
        two_stage(flat, &m.ucc_stage1, &m.ucc_stage2);
    }

    // ---- post-processor: offset trimming + the special tokens it puts around a single sequence ----
    {
        std::function<void(const JsonValue*)> apply = [&](const JsonValue* pp) {
            if (!pp || pp->is_null()) return;
            std::string t = pp->get_str("type");
            if (t == "BertProcessing" || t == "RobertaProcessing") {
                // PostProcessorWrapper is an untagged enum that tries Roberta before Bert and "serde does not validate tags"
                // (processors/mod.rs:19-23): with both of Roberta's flags present it is a RobertaProcessing whatever `type` says, else Bert
                const JsonValue* a = pp->get("trim_offsets");
                const JsonValue* b = pp->get("add_prefix_space");
                t = (a && a->is_bool() && b && b->is_bool()) ? "RobertaProcessing" : "BertProcessing";
            }
            if (t == "ByteLevel" || t == "RobertaProcessing") {
                m.trim_offsets = pp->get_bool("trim_offsets", true);
                m.pp_add_prefix_space = pp->get_bool("add_prefix_space", true);
            }
            if (t == "ByteLevel") return;
            if (t == "BertProcessing" || t == "RobertaProcessing") {      // [cls] A [sep]   (processors/bert.rs:51-120, roberta.rs)
                const JsonValue* cls = pp->get("cls");
                const JsonValue* sep = pp->get("sep");
                if (!cls || !sep || !cls->is_array() || !sep->is_array() || cls->arr.size() != 2 || sep->arr.size() != 2)
                    throw Invalid("tokenizer.json: bad cls/sep in post_processor");
                m.pp_prefix.insert(m.pp_prefix.begin(), (uint32_t)cls->arr[1]->num);
                m.pp_suffix.push_back((uint32_t)sep->arr[1]->num);
                m.pp_prefix_ty.insert(m.pp_prefix_ty.begin(), (uint8_t)0);
                m.pp_suffix_ty.push_back((uint8_t)0);
                if (m.pp_single_typed) m.pp_unsupported = "a TemplateProcessing with type ids combined with another post-processor";
                if (!m.pp_pair.empty()) m.pp_pair_unsupported = "two post-processors that add special tokens";
                const uint32_t c = (uint32_t)cls->arr[1]->num, e = (uint32_t)sep->arr[1]->num;
                if (t == "BertProcessing")         // [CLS] A [SEP] : 0   B [SEP] : 1          (processors/bert.rs:121-150)
                    m.pp_pair = {{2, c, 0}, {0, 0, 0}, {2, e, 0}, {1, 0, 1}, {2, e, 1}};
                else                               // <s> A </s> </s> B </s>, all type 0        (processors/roberta.rs)
                    m.pp_roberta = true, m.pp_pair_plain = {{0, 0, 0}, {1, 0, 0}}, m.pp_pair = {{2, c, 0}, {0, 0, 0}, {2, e, 0}, {2, e, 0}, {1, 0, 0}, {2, e, 0}};
                return;
            }
            if (t == "TemplateProcessing") {                               // processors/template.rs:544-590, `single` template
                const JsonValue* single = pp->get("single");
                const JsonValue* sp = pp->get("special_tokens");
                if (!single || !single->is_array()) throw Invalid("tokenizer.json: TemplateProcessing without `single`");
                std::vector<uint32_t> pre, post;
                std::vector<uint8_t> pre_ty, post_ty;
                double seq_ty = 0, max_ty = 0;
                int n_seq = 0;
                // (a single template that is not "ids around sequence A" is refused when a single sequence is encoded with special tokens --
                // the pair template below is parsed regardless)
                for (auto& piece : single->arr) {
                    if (const JsonValue* sq = piece->get("Sequence")) {
                        if (sq->get_str("id") != "A") m.pp_unsupported = "TemplateProcessing single template refers to sequence B";
                        seq_ty = sq->get_num("type_id", 0);
                        max_ty = std::max(max_ty, seq_ty);
                        ++n_seq;
                    } else if (const JsonValue* st = piece->get("SpecialToken")) {
                        std::string name = st->get_str("id");
                        const JsonValue* def = sp ? sp->get(name.c_str()) : nullptr;
                        const JsonValue* ids = def ? def->get("ids") : nullptr;
                        if (!ids || !ids->is_array()) throw Invalid("tokenizer.json: TemplateProcessing special token '" + name + "' is not defined");
                        const double ty = st->get_num("type_id", 0);
                        max_ty = std::max(max_ty, ty);
                        for (auto& x : ids->arr) { (n_seq ? post : pre).push_back((uint32_t)x->num); (n_seq ? post_ty : pre_ty).push_back((uint8_t)ty); }
                    } else throw Invalid("tokenizer.json: bad TemplateProcessing piece");
                }
                if (n_seq != 1 && m.pp_unsupported.empty()) m.pp_unsupported = "TemplateProcessing single template must contain sequence A exactly once";
                if (max_ty > 255 && m.pp_unsupported.empty()) m.pp_unsupported = "TemplateProcessing type id above 255";
                if (max_ty > 0 && m.pp_unsupported.empty()) {
                    // (a second post-processor would see -- and a second template overwrite -- these type ids)
                    if (!m.pp_prefix.empty() || !m.pp_suffix.empty() || m.pp_single_typed) m.pp_unsupported = "a TemplateProcessing with type ids combined with another post-processor";
                    else m.pp_single_typed = true, m.pp_seq_ty = (uint32_t)seq_ty;
                } else if (m.pp_single_typed && m.pp_unsupported.empty()) {
                    m.pp_unsupported = "a TemplateProcessing with type ids combined with another post-processor";
                }
                if (m.pp_unsupported.empty()) {
                    m.pp_prefix.insert(m.pp_prefix.begin(), pre.begin(), pre.end());
                    m.pp_suffix.insert(m.pp_suffix.end(), post.begin(), post.end());
                    m.pp_prefix_ty.insert(m.pp_prefix_ty.begin(), pre_ty.begin(), pre_ty.end());
                    m.pp_suffix_ty.insert(m.pp_suffix_ty.end(), post_ty.begin(), post_ty.end());
                } else {
                    m.pp_single_typed = false;
                    m.pp_single_refused = true;           // (sequence A twice, or typed, comes out that way without special tokens too)
                }
                // the `pair` template (processors/template.rs:544-590): any order of A, B and special tokens, each with its type id
                if (!m.pp_pair.empty()) m.pp_pair_unsupported = "two post-processors that add special tokens";
                const JsonValue* pair = pp->get("pair");
                if (!pair || !pair->is_array()) { m.pp_pair_unsupported = "TemplateProcessing without a `pair` template"; return; }
                int na = 0, nb = 0;
                for (auto& piece : pair->arr) {
                    if (const JsonValue* sq = piece->get("Sequence")) {
                        const bool is_a = sq->get_str("id") == "A";
                        (is_a ? na : nb)++;
                        m.pp_pair.push_back({is_a ? 0u : 1u, 0u, (uint32_t)sq->get_num("type_id", 0)});
                    } else if (const JsonValue* st = piece->get("SpecialToken")) {
                        std::string name = st->get_str("id");
                        const JsonValue* def = sp ? sp->get(name.c_str()) : nullptr;
                        const JsonValue* ids = def ? def->get("ids") : nullptr;
                        if (!ids || !ids->is_array()) throw Invalid("tokenizer.json: TemplateProcessing special token '" + name + "' is not defined");
                        for (auto& x : ids->arr) m.pp_pair.push_back({2u, (uint32_t)x->num, (uint32_t)st->get_num("type_id", 0)});
                    } else throw Invalid("tokenizer.json: bad TemplateProcessing piece");
                }
                if (na != 1 || nb != 1) m.pp_pair_unsupported = "TemplateProcessing pair template must contain A and B exactly once each";
                m.pp_pair_plain.clear();
                for (const HostModel::TplPiece& q : m.pp_pair)
                    if (q.kind != 2u) m.pp_pair_plain.push_back(q);
                if (m.pp_pair.size() > 64) m.pp_pair_unsupported = "TemplateProcessing pair template with more than 64 pieces";
                return;
            }
            if (t == "Sequence") {                                         // processors/sequence.rs: applied in order
                const JsonValue* ps = pp->get("processors");
                if (ps && ps->is_array())
                    for (auto& q : ps->arr) apply(q.get());
                return;
            }
            m.pp_unsupported = "post_processor type '" + t + "' is outside the hot path";
        };
        apply(root->get("post_processor"));
        for (size_t k = 0; k < m.pp_prefix.size(); ++k) m.pp_single.push_back({2u, m.pp_prefix[k], k < m.pp_prefix_ty.size() ? (uint32_t)m.pp_prefix_ty[k] : 0u});
        m.pp_single.push_back({0u, 0u, m.pp_single_typed ? m.pp_seq_ty : 0u});
        for (size_t k = 0; k < m.pp_suffix.size(); ++k) m.pp_single.push_back({2u, m.pp_suffix[k], k < m.pp_suffix_ty.size() ? (uint32_t)m.pp_suffix_ty[k] : 0u});
        m.pp_single_plain = {{0u, 0u, m.pp_single_typed ? m.pp_seq_ty : 0u}};
    }

    // ---- added tokens ----
    const JsonValue* at = root->get("added_tokens");
    if (at && at->is_array()) {
        for (auto& e : at->arr) {
            AddedToken a;
            a.content = e->get_str("content");
            a.id = (uint32_t)e->get_num("id", 0);
            if (e->get_num("id", 0) < 0 || a.id >= (1u << 24)) throw Unsupported("added token id beyond 2^24");
            a.special = e->get_bool("special", false);
            a.single_word = e->get_bool("single_word", false);
            a.lstrip = e->get_bool("lstrip", false);
            a.rstrip = e->get_bool("rstrip", false);
            a.normalized = e->get_bool("normalized", false);
            m.added_tokens.push_back(std::move(a));
        }
    }

    // ---- model ----
    const JsonValue* model = root->get("model");
    if (!model || !model->is_object()) throw Invalid("tokenizer.json: missing model");
    std::string mtype = model->get_str("type");
    const JsonValue* vocab = model->get("vocab");
    if (mtype.empty()) {
        // legacy untagged models (models/mod.rs:71-136): BPE has merges, WordPiece has a prefix
        if (model->get("merges")) mtype = "BPE";
        else if (model->get("continuing_subword_prefix")) mtype = "WordPiece";
        else mtype = "WordLevel";
    }
    if (!vocab || !vocab->is_object()) throw Invalid("tokenizer.json: model.vocab missing");

    std::unordered_map<std::string, uint32_t> v;
    v.reserve(vocab->obj.size() * 2);
    for (auto& kv : vocab->obj) {
        if (!kv.second->is_number()) throw Invalid("tokenizer.json: vocab id is not a number");
        v[kv.first] = (uint32_t)kv.second->num;   // duplicate keys: last wins, like serde's map
        // the pair hashes (tables.hpp) multiply 24-bit operands and the result rows keep ids in 24 bits: larger ids would
        // collide for every seed, so they are refused here with the reason instead of failing table construction
        if (kv.second->num < 0 || kv.second->num >= (double)(1u << 24)) throw Unsupported("token id beyond 2^24 (vocab entry '" + kv.first + "')");
    }
    m.vocab_size = (uint32_t)v.size();
    if (m.trim_offsets && !m.byte_level) {
        // a trimming post-processor (ByteLevel / RobertaProcessing) on a model that is not byte-level: process_offsets moves the offsets
        // of every token whose STRING starts or ends with whitespace or 'Ġ' (byte_level.rs:202-234).  The added tokens' raw slices
        // can (k_token_meta trims those); a vocabulary entry that could is outside the path
        if (m.uc_stage1.empty()) build_unicode(m);
        for (auto& kv : v) {
            const uint8_t* b = (const uint8_t*)kv.first.data();
            const size_t n = kv.first.size();
            if (!n) continue;
            uint32_t c0 = 0, c1 = 0;
            utf8_decode(b, n, &c0);
            size_t q = n - 1;
            while (q > 0 && (b[q] & 0xC0u) == 0x80u) --q;
            utf8_decode(b + q, n - q, &c1);
            auto trimmed = [&](uint32_t c) { return c == 0x120u || (c < 0x110000u && (m.uc_stage2[((size_t)m.uc_stage1[c >> 8] << 8) | (c & 255u)] & UC_RUST_WS)); };
            if (trimmed(c0) || trimmed(c1)) throw Unsupported("post_processor with trim_offsets on a vocabulary that is not byte-level and holds an entry with leading / trailing whitespace or 'Ġ' ('" + kv.first + "')");
        }
    }

    // The ids the reference gives the added tokens are not the ones in the file: deserialisation hands them, in file order, to
    // AddedVocabulary::add_tokens (serialization.rs:153-167 -- it only WARNS when the outcome differs from the file), which gives a
    // token whose content the model knows the model's id and every other one the next free id from the model's vocabulary size on; a
    // content seen twice keeps its first id and its last properties (added_vocabulary.rs:272-360).  Files the library wrote agree with
    // that already.
    {
        std::unordered_map<std::string, size_t> seen;
        std::vector<AddedToken> kept;
        uint32_t next_id = m.vocab_size;
        for (AddedToken& a : m.added_tokens) {
            if (a.content.empty()) continue;
            auto it = seen.find(a.content);
            if (it != seen.end()) {
                AddedToken& old = kept[it->second];
                a.id = old.id;
                old = a;
                continue;
            }
            auto vit = v.find(a.content);
            a.id = vit != v.end() ? vit->second : next_id++;
            if (a.id >= (1u << 24)) throw Unsupported("added token id beyond 2^24");
            seen[a.content] = kept.size();
            kept.push_back(a);
        }
        m.added_tokens = std::move(kept);
    }

    uint32_t b2c[256];
    build_bytes_char(b2c);
    std::unordered_map<uint32_t, uint8_t> c2b;
    for (int b = 0; b < 256; ++b) c2b[b2c[b]] = (uint8_t)b;

    // raw-byte view of the vocab
    m.raw_tokens.reserve(v.size());
    m.raw_ids.reserve(v.size());
    for (auto& kv : v) {
        if (m.byte_level) {
            std::string raw;
            if (!bytelevel_to_raw(kv.first, c2b, &raw)) continue;   // not producible from bytes
            m.raw_tokens.push_back(std::move(raw));
        } else {
            m.raw_tokens.push_back(kv.first);
        }
        m.raw_ids.push_back(kv.second);
    }
    const size_t n_raw_plain = m.raw_tokens.size();

    build_decode_tables(m, root.get(), v, c2b);

    auto opt_str = [&](const char* key, std::string* out) -> bool {
        const JsonValue* x = model->get(key);
        if (x && x->is_string()) { *out = x->str; return true; }
        return false;
    };

    if (mtype == "BPE") {
        m.model = MODEL_BPE;
        const JsonValue* dr = model->get("dropout");
        if (dr && dr->is_number() && dr->num != 0.0) throw Unsupported("BPE dropout needs the reference's RNG (bpe/word.rs:181)");
        opt_str("continuing_subword_prefix", &m.bpe_prefix);
        opt_str("end_of_word_suffix", &m.bpe_suffix);
        m.fuse_unk = model->get_bool("fuse_unk", false);
        m.byte_fallback = model->get_bool("byte_fallback", false);
        m.ignore_merges = model->get_bool("ignore_merges", false);
        if (opt_str("unk_token", &m.unk_token)) {
            m.unk_configured = true;
            auto it = v.find(m.unk_token);
            if (it != v.end()) { m.has_unk = true; m.unk_id = it->second; }
        }
        m.char_bpe = !m.byte_level;
        if (m.byte_level && (!m.bpe_prefix.empty() || !m.bpe_suffix.empty())) throw Unsupported("byte-level BPE with continuing_subword_prefix / end_of_word_suffix");
        if (m.byte_level && m.byte_fallback) throw Unsupported("byte-level BPE with byte_fallback");
        if (m.char_bpe) {
            // BPE over characters (BPE::merge_word, bpe/model.rs:465-550): a char's initial symbol is the vocabulary entry of the char with
            // the affixes its place in the word glues on.  Every (char, affix combination) the vocabulary knows goes into one direct
            // indexed table
            if (m.bpe_prefix.size() > 31 || m.bpe_suffix.size() > 31) throw Unsupported("BPE continuing_subword_prefix / end_of_word_suffix longer than 31 bytes");
            m.char_id.assign(CHAR_TABLE_WORDS, CHAR_NONE);
            for (auto& kv : v) {
                const std::string& e = kv.first;
                for (uint32_t var = 0; var < 4; ++var) {
                    if ((var & 1u) && m.bpe_prefix.empty()) continue;
                    if ((var & 2u) && m.bpe_suffix.empty()) continue;
                    size_t a = 0, b = e.size();
                    if (var & 1u) { if (e.compare(0, m.bpe_prefix.size(), m.bpe_prefix) != 0 || e.size() < m.bpe_prefix.size()) continue; a = m.bpe_prefix.size(); }
                    if (var & 2u) { if (b - a < m.bpe_suffix.size() || e.compare(b - m.bpe_suffix.size(), m.bpe_suffix.size(), m.bpe_suffix) != 0) continue; b -= m.bpe_suffix.size(); }
                    if (b <= a || b - a > 4) continue;
                    uint32_t cp = 0;
                    const size_t l = utf8_decode((const uint8_t*)e.data() + a, b - a, &cp);
                    if (l != b - a || cp >= 0x110000u) continue;                  // exactly one char between the affixes
                    m.char_id[((size_t)cp << 2) | var] = kv.second;
                }
            }
            for (int b = 0; b < 256; ++b) m.byte_id[b] = CHAR_NONE;
            if (m.byte_fallback) {
                // (merge_word glues the affixes on BEFORE it falls back to bytes, so they would be spelt out as <0xXX> tokens too --
                // symbols the word has no bytes for; and a byte without its token sends the char on to the unk path behind symbols
                // already added: both corners are refused)
                if (!m.bpe_prefix.empty() || !m.bpe_suffix.empty()) throw Unsupported("BPE byte_fallback together with continuing_subword_prefix / end_of_word_suffix");
                for (int b = 0; b < 256; ++b) {
                    char code[8];
                    snprintf(code, sizeof(code), "<0x%02X>", b);
                    auto it = v.find(code);
                    if (it == v.end()) throw Unsupported(std::string("BPE byte_fallback without the byte token ") + code);
                    m.byte_id[b] = it->second;
                }
            }
        }
        // byte -> initial symbol id (bpe/model.rs:494-499 with the byte-level alphabet)
        for (int b = 0; b < 256 && m.byte_level; ++b) {
            std::string ch;
            uint32_t cp = b2c[b];
            if (cp < 0x80) ch.push_back((char)cp);
            else { ch.push_back((char)(0xC0 | (cp >> 6))); ch.push_back((char)(0x80 | (cp & 0x3F))); }
            auto it = v.find(ch);
            if (it == v.end())
                throw Unsupported("byte-level BPE vocab lacks a byte symbol (unk / byte_fallback / dropped-char paths, bpe/model.rs:501-541)");
            m.byte_id[b] = it->second;
        }
        // merges (bpe/model.rs:252-275): rank = position, new_id = vocab[a+b]; duplicate pairs: last wins
        const JsonValue* mg = model->get("merges");
        if (!mg || !mg->is_array()) throw Invalid("tokenizer.json: BPE merges missing");
        std::unordered_map<uint64_t, std::pair<uint32_t, uint32_t>> mm;
        mm.reserve(mg->arr.size() * 2);
        uint32_t rank = 0;
        for (auto& e : mg->arr) {
            std::string a, b;
            if (e->is_array() && e->arr.size() == 2 && e->arr[0]->is_string() && e->arr[1]->is_string()) {
                a = e->arr[0]->str; b = e->arr[1]->str;
            } else if (e->is_string()) {     // legacy "a b" (bpe/serialization.rs:141-149)
                size_t sp = e->str.find(' ');
                if (sp == std::string::npos || e->str.find(' ', sp + 1) != std::string::npos)
                    throw Invalid("tokenizer.json: bad legacy merge entry");
                a = e->str.substr(0, sp); b = e->str.substr(sp + 1);
            } else throw Invalid("tokenizer.json: bad merge entry");
            // (new token = a + b minus the continuing_subword_prefix's LENGTH in bytes, whatever b starts with: BpeBuilder::build :246-271)
            if (b.size() < m.bpe_prefix.size()) throw Invalid("tokenizer.json: a merge's right-hand token is shorter than continuing_subword_prefix");
            auto ia = v.find(a), ib = v.find(b), in = v.find(a + b.substr(m.bpe_prefix.size()));
            if (ia == v.end() || ib == v.end() || in == v.end())
                throw Invalid("tokenizer.json: merge token out of vocabulary (MergeTokenOutOfVocabulary)");
            mm[((uint64_t)ia->second << 32) | ib->second] = {rank, in->second};
            ++rank;
        }
        if (rank >= (1u << 20)) throw Unsupported("more than 2^20 merges");
        m.n_merges = rank;
        std::vector<MergeSlot> merges;
        merges.reserve(mm.size());
        for (auto& kv : mm) merges.push_back(MergeSlot{(uint32_t)(kv.first >> 32), (uint32_t)kv.first, kv.second.first, kv.second.second});
        std::sort(merges.begin(), merges.end(), [](const MergeSlot& x, const MergeSlot& y) { return x.rank < y.rank; });
        build_merge_table(m, merges);
    } else if (mtype == "WordPiece") {
        m.model = MODEL_WORDPIECE;
        if (m.byte_level) throw Unsupported("WordPiece behind ByteLevel");
        m.unk_token = model->get_str("unk_token", "[UNK]");
        m.cont_prefix = model->get_str("continuing_subword_prefix", "##");
        m.max_input_chars = (uint32_t)model->get_num("max_input_chars_per_word", 100);
        auto it = v.find(m.unk_token);
        if (it != v.end()) { m.has_unk = true; m.unk_id = it->second; }
    } else if (mtype == "WordLevel") {
        m.model = MODEL_WORDLEVEL;
        if (m.byte_level) throw Unsupported("WordLevel behind ByteLevel");
        m.unk_token = model->get_str("unk_token", "<unk>");
        auto it = v.find(m.unk_token);
        if (it != v.end()) { m.has_unk = true; m.unk_id = it->second; }
    } else {
        throw Unsupported("model: type '" + mtype + "' is outside the hot path");
    }

    // BPE over characters with an end_of_word_suffix: the vocabulary entry of a WHOLE word carries the suffix its text does not.  The
    // whole-word table (below) is keyed by text: the merge-stable shortcut's candidates are the entries minus their suffix (an entry
    // without it cannot be a word's last symbol, let alone the whole word) -- each is then run through the device's merge kernel and
    // only kept if the result is exactly its own id (capi.cpp verify_direct_words).  ignore_merges looks the TEXT up as it stands
    // (vocab.get(sequence), bpe/model.rs:559-567): no stripping there.
    if (m.char_bpe && !m.ignore_merges && !m.bpe_suffix.empty()) {
        (void)n_raw_plain;
        for (std::string& r : m.raw_tokens) {
            if (r.size() > m.bpe_suffix.size() && r.compare(r.size() - m.bpe_suffix.size(), m.bpe_suffix.size(), m.bpe_suffix) == 0) r.resize(r.size() - m.bpe_suffix.size());
            else r.clear();
        }
        // (two entries may now share a text -- "ab</w>" and a stray "ab</w></w>" -- : the table wants distinct keys, the first one stays)
        std::unordered_map<std::string, size_t> seen;
        for (size_t i = 0; i < m.raw_tokens.size(); ++i)
            if (!m.raw_tokens[i].empty() && !seen.emplace(m.raw_tokens[i], i).second) m.raw_tokens[i].clear();
    }

    // ---- whole-word tables (BPE: ignore_merges + merge-stable shortcut; WordLevel: the model) ----
    if (m.model == MODEL_BPE || m.model == MODEL_WORDLEVEL || m.model == MODEL_WORDPIECE) {
        // (WordPiece: the longest candidate piece is the whole word, so a whole-word hit ends the match at once)
        std::vector<WordSlot> words;
        m.long_off.push_back(0);
        for (size_t i = 0; i < m.raw_tokens.size(); ++i) {
            const std::string& r = m.raw_tokens[i];
            if (r.empty()) continue;
            if (r.size() <= (size_t)WORD_MAX_KEY) {
                uint8_t buf[16] = {0};
                memcpy(buf, r.data(), r.size());
                WordSlot s{};
                memcpy(&s.lo, buf, 8);
                memcpy(&s.hi, buf + 8, 8);
                s.len = (uint32_t)r.size();
                s.id = m.raw_ids[i];
                s.flags = 0;
                words.push_back(s);
            } else {
                m.long_blob.insert(m.long_blob.end(), r.begin(), r.end());
                m.long_off.push_back((uint32_t)m.long_blob.size());
                m.long_id.push_back(m.raw_ids[i]);
            }
        }
        m.n_words = (uint32_t)words.size();
        build_word_table(m, words);
        build_long_table(m);
    }
    if (m.model == MODEL_WORDPIECE) {
        std::vector<std::pair<std::string, uint32_t>> ini, cont;
        for (size_t i = 0; i < m.raw_tokens.size(); ++i) {
            const std::string& r = m.raw_tokens[i];
            if (r.empty()) continue;
            ini.emplace_back(r, m.raw_ids[i]);
            if (!m.cont_prefix.empty() && r.size() > m.cont_prefix.size() && r.compare(0, m.cont_prefix.size(), m.cont_prefix) == 0)
                cont.emplace_back(r.substr(m.cont_prefix.size()), m.raw_ids[i]);
            else if (m.cont_prefix.empty())
                cont.emplace_back(r, m.raw_ids[i]);
        }
        build_trie(m, ini, cont);
    }

    build_unicode(m);
    if (m.norm == NORM_BERT) build_bert_norm(m);

    // ---- AddedVocabulary pattern sets (needs the normalizer tables: normalized tokens match by their normalized form) ----
    {
        struct Pat { std::string s; uint32_t id, flags; };
        std::vector<Pat> pats[2];
        // the automaton is built over the special tokens first, then the others, each in the order they were added
        // (refresh_added_tokens, added_vocabulary.rs:379-399): of two tokens with one pattern -- "Ab" and "AB" behind a lowercasing
        // normalizer -- the first in THAT order is the one a match reports
        std::vector<const AddedToken*> order;
        for (int pass = 0; pass < 2; ++pass)
            for (const AddedToken& a : m.added_tokens)
                if ((pass == 0) == a.special) order.push_back(&a);
        for (const AddedToken* ap : order) {
            const AddedToken& a = *ap;
            if (a.content.empty()) continue;                       // add_tokens ignores empty contents (added_vocabulary.rs:288-291)
            std::string pat = a.content;
            if (a.normalized && m.norm == NORM_BERT) {
                bool refused = false;
                pat = m.bert_normalize(a.content, &refused);
                if (refused) throw Unsupported("added token '" + a.content + "' holds a character whose NFD reordering is context dependent");
                if (pat.empty()) continue;                         // normalizes to nothing: the automaton has nothing to match
            }
            pats[a.normalized ? 1 : 0].push_back(Pat{pat, a.id, (a.single_word ? 1u : 0u) | (a.lstrip ? 2u : 0u) | (a.rstrip ? 4u : 0u) | (a.special ? 8u : 0u)});
        }
        for (int c = 0; c < 2; ++c) {
            std::vector<Pat>& ps = pats[c];
            // equal patterns: the first one in the automaton's order is reported
            std::stable_sort(ps.begin(), ps.end(), [](const Pat& x, const Pat& y) { return x.s < y.s; });
            std::vector<Pat> uniq;
            for (const Pat& p : ps) { if (uniq.empty() || uniq.back().s != p.s) uniq.push_back(p); }
            PatternSet& S = m.at[c];
            S.first.assign(257, 0);
            S.off.push_back(0);
            for (const Pat& p : uniq) {
                S.first[(uint8_t)p.s[0] + 1]++;
                S.blob.insert(S.blob.end(), p.s.begin(), p.s.end());
                S.off.push_back((uint32_t)S.blob.size());
                S.id.push_back(p.id);
                S.flags.push_back(p.flags);
            }
            for (int b = 0; b < 256; ++b) S.first[b + 1] += S.first[b];
        }
    }
    return m;
}

int HostModel::bn_expand_cp(uint32_t cp, uint32_t* out, int* refused) const {
    constexpr uint32_t DROP = 1, WS = 2, CJK = 4, REORDER = 8, D = 16, LC = 32;     // bert_norm_tables.inc flag bits
    auto flags = [&](uint32_t c) -> uint32_t { return c >= 0x110000u ? 0u : bn_stage2[((uint32_t)bn_stage1[c >> 8] << 8) | (c & 255u)]; };
    auto lookup = [&](uint32_t c, uint32_t kind, uint32_t* o) -> int {
        const MergeSlot& x = bn_map[merge_hash1(c, kind, bn_seed) & bn_mask];
        const MergeSlot& y = bn_map[merge_hash2(c, kind, bn_seed) & bn_mask];
        const MergeSlot* hit = (x.a == c && x.b == kind) ? &x : (y.a == c && y.b == kind) ? &y : nullptr;
        if (!hit) { o[0] = c; return 1; }
        const unsigned long long v = ((unsigned long long)hit->new_id << 32) | hit->rank;
        int k = 0;
        const uint32_t a = (uint32_t)(v & 0x1FFFFFu), c1 = (uint32_t)((v >> 21) & 0x1FFFFFu), c2 = (uint32_t)((v >> 42) & 0x1FFFFFu);
        if (a != 0x1FFFFFu) o[k++] = a;
        if (c1 != 0x1FFFFFu) o[k++] = c1;
        if (c2 != 0x1FFFFFu) o[k++] = c2;
        return k;
    };
    *refused = 0;
    uint32_t f = flags(cp);
    if (bn_clean_text) {
        if (f & DROP) return 0;
        if (f & WS) { cp = ' '; f = 0; }
    }
    int k = 0;
    const bool cjk = bn_handle_chinese && (f & CJK);
    if (cjk) out[k++] = ' ';
    uint32_t seq[3] = {cp, 0, 0};
    int n1 = 1;
    if (bn_strip_accents) {
        if (f & REORDER) *refused = 1;
        if (f & D) n1 = lookup(cp, 0, seq);
    }
    for (int q = 0; q < n1; ++q) {
        const uint32_t y = seq[q];
        if (bn_lowercase && (flags(y) & LC)) k += lookup(y, 1, out + k);
        else out[k++] = y;
    }
    if (cjk) out[k++] = ' ';
    return k;
}

std::string HostModel::bert_normalize(const std::string& s, bool* refused) const {
    std::string out;
    size_t i = 0;
    while (i < s.size()) {
        const uint8_t b0 = (uint8_t)s[i];
        uint32_t cp, len;
        auto cb = [&](size_t k) -> uint32_t { return i + k < s.size() ? ((uint8_t)s[i + k] & 0x3Fu) : 0u; };
        if (b0 < 0x80u) { cp = b0; len = 1; }
        else if (b0 < 0xE0u) { cp = ((b0 & 0x1Fu) << 6) | cb(1); len = 2; }
        else if (b0 < 0xF0u) { cp = ((b0 & 0x0Fu) << 12) | (cb(1) << 6) | cb(2); len = 3; }
        else { cp = ((b0 & 0x07u) << 18) | (cb(1) << 12) | (cb(2) << 6) | cb(3); len = 4; }
        const size_t at = i;
        i += len;
        uint32_t o[12];
        int r = 0;
        const int n = bn_expand_cp(cp, o, &r);
        // (a character NFD's canonical ordering could move: fine as long as it is alone in its run, bert_norm_core.hpp)
        if (r && refused &&
            !bn_alone_in_run(bn_stage1.data(), bn_stage2.data(), bn_clean_text, (const uint8_t*)s.data(), 0, (int64_t)s.size(), (int64_t)at, len,
                             bn_core_flags(bn_stage1.data(), bn_stage2.data(), cp), nullptr))
            *refused = true;
        for (int q = 0; q < n; ++q) {
            const uint32_t c = o[q];
            if (c < 0x80u) out.push_back((char)c);
            else if (c < 0x800u) { out.push_back((char)(0xC0u | (c >> 6))); out.push_back((char)(0x80u | (c & 0x3Fu))); }
            else if (c < 0x10000u) { out.push_back((char)(0xE0u | (c >> 12))); out.push_back((char)(0x80u | ((c >> 6) & 0x3Fu))); out.push_back((char)(0x80u | (c & 0x3Fu))); }
            else { out.push_back((char)(0xF0u | (c >> 18))); out.push_back((char)(0x80u | ((c >> 12) & 0x3Fu))); out.push_back((char)(0x80u | ((c >> 6) & 0x3Fu))); out.push_back((char)(0x80u | (c & 0x3Fu))); }
        }
    }
    return out;
}

}  // namespace tkamd
