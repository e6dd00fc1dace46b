/*
 * oracle.c -- CPU restatement of the reference's encode path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under tokenizers_amd/ links, imports or calls this file; it exists so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the HIP path against an
 * independent, sequential, per-document implementation that follows the reference line by line:
 *
 *   pre-tokenizers : byte_level.rs:43-46,119-148  (GPT-2 regex, matched alternative by alternative,
 *                    leftmost-first, exactly as a backtracking engine would -- NOT the window
 *                    predicate the GPU uses), split.rs:96-104 with the Llama-3 pattern,
 *                    whitespace.rs:20-41, bert.rs:5-17
 *   normalizer     : normalizers/bert.rs:92-138 (ASCII documents only; returns an error otherwise)
 *   models         : bpe/model.rs:465-612 + bpe/word.rs:162-250 (binary heap ordered by (rank,pos),
 *                    lazy invalidation by new_id, two re-pushes), wordpiece/mod.rs:224-283,
 *                    wordlevel/mod.rs:162-178
 *   encoding       : pre_tokenizer.rs:198-263 (ids, byte offsets into the original, word = split index)
 *
 * Parity pinning: tests/test_oracle.py replays the reference's own inline known-answer tests
 *  and the committed golden vectors produced by the reference wheel.
 * Unicode classes come from the generated table (oracle/gen_unicode_tables.py probes the wheel).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* unicode flags                                                                               */
/* ------------------------------------------------------------------------------------------- */
#define UC_ONIG_L 1
#define UC_ONIG_N 2
#define UC_ONIG_S 4
#define UC_RX_W 8
#define UC_RX_S 16
#define UC_RUST_WS 32
#define UC_BERT_P 64

typedef struct { uint32_t first, last; uint8_t flags; } uc_run;
static const uc_run UC_RUNS[] = {
#include "../tokenizers_amd/csrc/unicode_ranges.inc"
};
static uint8_t* UC_FLAT = NULL;

static void uc_init(void) {
    if (UC_FLAT) return;
    UC_FLAT = (uint8_t*)calloc(0x110000, 1);
    for (size_t i = 0; i < sizeof(UC_RUNS) / sizeof(UC_RUNS[0]); ++i)
        for (uint32_t cp = UC_RUNS[i].first; cp <= UC_RUNS[i].last; ++cp) UC_FLAT[cp] = UC_RUNS[i].flags;
}
static inline uint8_t uc(uint32_t cp) { return cp < 0x110000 ? UC_FLAT[cp] : 0; }

/* decode one scalar at s[i] (valid UTF-8 assumed, as Rust &str guarantees) */
static inline uint32_t u8dec(const uint8_t* s, int64_t i, int64_t n, int* len) {
    uint8_t b = s[i];
    if (b < 0x80) { *len = 1; return b; }
    if (b < 0xE0 && i + 1 < n) { *len = 2; return ((b & 0x1Fu) << 6) | (s[i + 1] & 0x3Fu); }
    if (b < 0xF0 && i + 2 < n) { *len = 3; return ((b & 0x0Fu) << 12) | ((s[i + 1] & 0x3Fu) << 6) | (s[i + 2] & 0x3Fu); }
    if (i + 3 < n) { *len = 4; return ((b & 0x07u) << 18) | ((s[i + 1] & 0x3Fu) << 12) | ((s[i + 2] & 0x3Fu) << 6) | (s[i + 3] & 0x3Fu); }
    *len = 1;
    return 0xFFFD;
}
static inline int u8len(uint8_t b) { return b < 0x80 ? 1 : b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4; }

/* ------------------------------------------------------------------------------------------- */
/* string -> u32 hash map (vocab: AHashMap<String,u32>)                                         */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t* blob; int64_t blob_len, blob_cap;
    int64_t* off; uint32_t* len; uint32_t* val; int64_t n, cap;
    int64_t* tab; int64_t tab_mask;
} strmap;

static uint64_t fnv64(const uint8_t* p, int64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
static void strmap_init(strmap* m) { memset(m, 0, sizeof(*m)); }
static void strmap_free(strmap* m) { free(m->blob); free(m->off); free(m->len); free(m->val); free(m->tab); memset(m, 0, sizeof(*m)); }
static void strmap_rehash(strmap* m) {
    int64_t cap = 64;
    while (cap < (m->n + 1) * 2) cap <<= 1;
    free(m->tab);
    m->tab = (int64_t*)malloc(sizeof(int64_t) * cap);
    for (int64_t i = 0; i < cap; ++i) m->tab[i] = -1;
    m->tab_mask = cap - 1;
    for (int64_t e = 0; e < m->n; ++e) {
        uint64_t h = fnv64(m->blob + m->off[e], m->len[e]) & (uint64_t)m->tab_mask;
        while (m->tab[h] >= 0) h = (h + 1) & (uint64_t)m->tab_mask;
        m->tab[h] = e;
    }
}
static int64_t strmap_find(const strmap* m, const uint8_t* k, int64_t klen) {
    if (!m->tab) return -1;
    uint64_t h = fnv64(k, klen) & (uint64_t)m->tab_mask;
    while (m->tab[h] >= 0) {
        int64_t e = m->tab[h];
        if ((int64_t)m->len[e] == klen && !memcmp(m->blob + m->off[e], k, (size_t)klen)) return e;
        h = (h + 1) & (uint64_t)m->tab_mask;
    }
    return -1;
}
static void strmap_put(strmap* m, const uint8_t* k, int64_t klen, uint32_t v) {
    int64_t e = strmap_find(m, k, klen);
    if (e >= 0) { m->val[e] = v; return; }     /* duplicate key: last wins */
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 1024;
        m->off = (int64_t*)realloc(m->off, sizeof(int64_t) * m->cap);
        m->len = (uint32_t*)realloc(m->len, sizeof(uint32_t) * m->cap);
        m->val = (uint32_t*)realloc(m->val, sizeof(uint32_t) * m->cap);
    }
    if (m->blob_len + klen > m->blob_cap) {
        m->blob_cap = (m->blob_cap + klen) * 2 + 64;
        m->blob = (uint8_t*)realloc(m->blob, (size_t)m->blob_cap);
    }
    memcpy(m->blob + m->blob_len, k, (size_t)klen);
    m->off[m->n] = m->blob_len;
    m->len[m->n] = (uint32_t)klen;
    m->val[m->n] = v;
    m->blob_len += klen;
    m->n++;
    if (!m->tab || m->n * 2 > m->tab_mask) strmap_rehash(m);
    else {
        uint64_t h = fnv64(k, klen) & (uint64_t)m->tab_mask;
        while (m->tab[h] >= 0) h = (h + 1) & (uint64_t)m->tab_mask;
        m->tab[h] = m->n - 1;
    }
}
static int strmap_get(const strmap* m, const uint8_t* k, int64_t klen, uint32_t* v) {
    int64_t e = strmap_find(m, k, klen);
    if (e < 0) return 0;
    *v = m->val[e];
    return 1;
}

/* pair -> (rank,new_id) map (MergeMap = AHashMap<Pair,(u32,u32)>) */
typedef struct { uint64_t* key; uint32_t* rank; uint32_t* nid; int64_t mask; } pairmap;
static void pairmap_init(pairmap* m, int64_t n) {
    int64_t cap = 64;
    while (cap < n * 2 + 2) cap <<= 1;
    m->key = (uint64_t*)malloc(sizeof(uint64_t) * cap);
    m->rank = (uint32_t*)malloc(sizeof(uint32_t) * cap);
    m->nid = (uint32_t*)malloc(sizeof(uint32_t) * cap);
    for (int64_t i = 0; i < cap; ++i) m->key[i] = ~0ull;
    m->mask = cap - 1;
}
static void pairmap_free(pairmap* m) { free(m->key); free(m->rank); free(m->nid); memset(m, 0, sizeof(*m)); }
static inline uint64_t pair_hash(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }
static void pairmap_put(pairmap* m, uint32_t a, uint32_t b, uint32_t rank, uint32_t nid) {
    uint64_t k = ((uint64_t)a << 32) | b, h = pair_hash(k) & (uint64_t)m->mask;
    while (m->key[h] != ~0ull && m->key[h] != k) h = (h + 1) & (uint64_t)m->mask;
    m->key[h] = k; m->rank[h] = rank; m->nid[h] = nid;
}
static int pairmap_get(const pairmap* m, uint32_t a, uint32_t b, uint32_t* rank, uint32_t* nid) {
    if (!m->key) return 0;
    uint64_t k = ((uint64_t)a << 32) | b, h = pair_hash(k) & (uint64_t)m->mask;
    while (m->key[h] != ~0ull) {
        if (m->key[h] == k) { *rank = m->rank[h]; *nid = m->nid[h]; return 1; }
        h = (h + 1) & (uint64_t)m->mask;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* tokenizer object                                                                            */
/* ------------------------------------------------------------------------------------------- */
enum { M_BPE = 1, M_WORDPIECE = 2, M_WORDLEVEL = 3 };
enum { PT_GPT2 = 1, PT_LLAMA3 = 2, PT_WS = 3, PT_WSSPLIT = 4, PT_BERT = 5, PT_BL_NOREGEX = 6 };
enum { N_NONE = 0, N_BERT = 1 };

typedef struct oracle_tok {
    int model, pretok, norm;
    int byte_level, add_prefix_space, ignore_merges, trim_offsets, pp_add_prefix_space;
    int char_offsets;               /* OffsetType::Char (BytesToCharOffsetConverter, pre_tokenizer.rs:329-364) */
    int has_unk; uint32_t unk_id;
    uint8_t cont_prefix[16]; int cont_prefix_len;
    int max_input_chars;
    strmap vocab;
    pairmap merges; int have_merges; int64_t n_merges;
    uint32_t b2c[256];              /* GPT-2 byte -> code point (byte_level.rs:15-39) */
    char err[256];
} oracle_tok;

#define ORACLE_OK 0
#define ORACLE_ERR_UNK -4           /* MissingUnkToken */
#define ORACLE_ERR_UNSUPPORTED -2

oracle_tok* oracle_new(int model, int pretok, int norm, int add_prefix_space, int ignore_merges, int trim_offsets) {
    uc_init();
    oracle_tok* t = (oracle_tok*)calloc(1, sizeof(oracle_tok));
    t->model = model; t->pretok = pretok; t->norm = norm;
    t->add_prefix_space = add_prefix_space; t->ignore_merges = ignore_merges; t->trim_offsets = trim_offsets;
    t->byte_level = (pretok == PT_GPT2 || pretok == PT_LLAMA3 || pretok == PT_BL_NOREGEX);
    t->max_input_chars = 100;
    memcpy(t->cont_prefix, "##", 2); t->cont_prefix_len = 2;
    strmap_init(&t->vocab);
    /* bytes_char(), byte_level.rs:15-39 */
    int direct[256] = {0}; uint32_t n = 0;
    for (int b = '!'; b <= '~'; ++b) direct[b] = 1;
    for (int b = 0xA1; b <= 0xAC; ++b) direct[b] = 1;
    for (int b = 0xAE; b <= 0xFF; ++b) direct[b] = 1;
    for (int b = 0; b < 256; ++b) t->b2c[b] = direct[b] ? (uint32_t)b : 256 + n++;
    return t;
}
void oracle_set_char_offsets(oracle_tok* t, int on) { t->char_offsets = on; }
void oracle_set_trim(oracle_tok* t, int trim, int pp_add_prefix_space) { t->trim_offsets = trim; t->pp_add_prefix_space = pp_add_prefix_space; }
void oracle_free(oracle_tok* t) {
    if (!t) return;
    strmap_free(&t->vocab);
    if (t->have_merges) pairmap_free(&t->merges);
    free(t);
}
const char* oracle_error(const oracle_tok* t) { return t->err; }

void oracle_set_vocab(oracle_tok* t, const uint8_t* blob, const int64_t* off, const uint32_t* ids, int64_t n) {
    for (int64_t i = 0; i < n; ++i) strmap_put(&t->vocab, blob + off[i], off[i + 1] - off[i], ids[i]);
}
void oracle_set_unk(oracle_tok* t, const uint8_t* s, int64_t n) {
    uint32_t id = 0;
    t->has_unk = strmap_get(&t->vocab, s, n, &id);
    t->unk_id = id;
}
void oracle_set_wordpiece(oracle_tok* t, const uint8_t* prefix, int64_t plen, int max_chars) {
    if (plen > 15) plen = 15;
    memcpy(t->cont_prefix, prefix, (size_t)plen); t->cont_prefix_len = (int)plen;
    t->max_input_chars = max_chars;
}
/* BpeBuilder::build, bpe/model.rs:252-275: (a_id,b_id) -> (rank=i, new_id=vocab[a+b]) */
int oracle_set_merges(oracle_tok* t, const uint8_t* blob_a, const int64_t* off_a, const uint8_t* blob_b, const int64_t* off_b, int64_t n) {
    pairmap_init(&t->merges, n);
    t->have_merges = 1; t->n_merges = n;
    uint8_t* buf = (uint8_t*)malloc(1 << 16);
    for (int64_t i = 0; i < n; ++i) {
        int64_t la = off_a[i + 1] - off_a[i], lb = off_b[i + 1] - off_b[i];
        uint32_t ia, ib, in;
        if (la + lb > (1 << 16)) { free(buf); return -1; }
        memcpy(buf, blob_a + off_a[i], (size_t)la); memcpy(buf + la, blob_b + off_b[i], (size_t)lb);
        if (!strmap_get(&t->vocab, blob_a + off_a[i], la, &ia) || !strmap_get(&t->vocab, blob_b + off_b[i], lb, &ib) ||
            !strmap_get(&t->vocab, buf, la + lb, &in)) { free(buf); return -1; }   /* MergeTokenOutOfVocabulary */
        pairmap_put(&t->merges, ia, ib, (uint32_t)i, in);
    }
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* splits                                                                                      */
/* ------------------------------------------------------------------------------------------- */
typedef struct { int64_t* s; int64_t* e; int64_t n, cap; } splits;
static void sp_push(splits* sp, int64_t s, int64_t e) {
    if (s == e) return;                            /* empty splits are filtered, pre_tokenizer.rs:90-96 */
    if (sp->n == sp->cap) {
        sp->cap = sp->cap ? sp->cap * 2 : 64;
        sp->s = (int64_t*)realloc(sp->s, sizeof(int64_t) * sp->cap);
        sp->e = (int64_t*)realloc(sp->e, sizeof(int64_t) * sp->cap);
    }
    sp->s[sp->n] = s; sp->e[sp->n] = e; sp->n++;
}

static inline int cls_L(uint32_t cp) { return uc(cp) & UC_ONIG_L; }
static inline int cls_N(uint32_t cp) { return uc(cp) & UC_ONIG_N; }
static inline int cls_S(uint32_t cp) { return uc(cp) & UC_ONIG_S; }

/* match `X+` for a class predicate starting at i; returns end */
#define RUN_WHILE(pred)                                           \
    while (j < n) { int l_; uint32_t c_ = u8dec(s, j, n, &l_); if (!(pred)) break; j += l_; }

/* One leftmost-first match of the GPT-2 regex at position i (byte_level.rs:43-46).  Returns the match end. */
static int64_t gpt2_match(const uint8_t* s, int64_t i, int64_t n) {
    /* 's|'t|'re|'ve|'m|'ll|'d */
    if (s[i] == '\'' && i + 1 < n) {
        uint8_t a = s[i + 1];
        if (a == 's' || a == 't') return i + 2;
        if (i + 2 < n && ((a == 'r' && s[i + 2] == 'e') || (a == 'v' && s[i + 2] == 'e'))) return i + 3;
        if (a == 'm') return i + 2;
        if (i + 2 < n && a == 'l' && s[i + 2] == 'l') return i + 3;
        if (a == 'd') return i + 2;
    }
    int64_t j, k = i;
    int l; uint32_t c;
    /* " ?\p{L}+" */
    if (s[k] == ' ') k++;
    if (k < n) { c = u8dec(s, k, n, &l); if (cls_L(c)) { j = k + l; RUN_WHILE(cls_L(c_)); return j; } }
    /* " ?\p{N}+" */
    k = i; if (s[k] == ' ') k++;
    if (k < n) { c = u8dec(s, k, n, &l); if (cls_N(c)) { j = k + l; RUN_WHILE(cls_N(c_)); return j; } }
    /* " ?[^\s\p{L}\p{N}]+" */
    k = i; if (s[k] == ' ') k++;
    if (k < n) { c = u8dec(s, k, n, &l); if (!cls_S(c) && !cls_L(c) && !cls_N(c)) { j = k + l; RUN_WHILE(!cls_S(c_) && !cls_L(c_) && !cls_N(c_)); return j; } }
    /* "\s+(?!\S)": greedy run of whitespace, backtracking one char at a time until not followed by \S */
    c = u8dec(s, i, n, &l);
    if (cls_S(c)) {
        /* collect run boundaries */
        int64_t ends[4096]; int ne = 0; j = i;
        int64_t run_end;
        while (j < n) { int l2; uint32_t c2 = u8dec(s, j, n, &l2); if (!cls_S(c2)) break; j += l2; if (ne < 4096) ends[ne] = j; ne++; }
        run_end = j;
        if (ne <= 4096) {
            for (int q = ne - 1; q >= 0; --q) {
                int64_t e = ends[q];
                if (e >= n) return e;                            /* end of text: (?!\S) holds */
                int l3; uint32_t c3 = u8dec(s, e, n, &l3);
                if (cls_S(c3)) return e;                         /* followed by whitespace */
            }
        } else {
            /* very long run: the only candidate ends that can satisfy (?!\S) are run_end (if at end of
             * text) and every end strictly inside the run (followed by whitespace): the longest is the
             * last-but-one char boundary */
            if (run_end >= n) return run_end;
            int64_t prev = i, cur = i;
            while (cur < run_end) { prev = cur; cur += u8len(s[cur]); }
            if (prev > i) return prev;
        }
        /* "\s+" */
        return run_end;
    }
    /* unreachable: every code point is covered by one alternative */
    return i + l;
}

static int is_crlf(uint32_t c) { return c == '\r' || c == '\n'; }
static uint32_t fold_ascii(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }

/* One leftmost-first match of the Llama-3 pattern (bindings/python/benches/test_tiktoken.py:38):
 * (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+ */
static int64_t llama3_match(const uint8_t* s, int64_t i, int64_t n) {
    int l; uint32_t c = u8dec(s, i, n, &l);
    int64_t j;
    if (c == '\'' && i + 1 < n) {
        /* (?i:...) uses Unicode simple case folding: U+017F (long s) folds to 's', U+212A (Kelvin) to 'k' */
        int l1; uint32_t a = u8dec(s, i + 1, n, &l1);
        uint32_t af = (a == 0x17F) ? 's' : fold_ascii(a);
        int64_t p2 = i + 1 + l1;
        uint32_t bf = 0; int l2 = 0;
        if (p2 < n) { uint32_t b = u8dec(s, p2, n, &l2); bf = (b == 0x17F) ? 's' : fold_ascii(b); }
        if (af == 's' || af == 't') return p2;
        if (p2 < n && ((af == 'r' && bf == 'e') || (af == 'v' && bf == 'e'))) return p2 + l2;
        if (af == 'm') return p2;
        if (p2 < n && af == 'l' && bf == 'l') return p2 + l2;
        if (af == 'd') return p2;
    }
    /* [^\r\n\p{L}\p{N}]?\p{L}+ */
    {
        int64_t k = i;
        if (!is_crlf(c) && !cls_L(c) && !cls_N(c)) k = i + l;      /* optional prefix char taken greedily */
        if (k < n) {
            int lk; uint32_t ck = u8dec(s, k, n, &lk);
            if (cls_L(ck)) { j = k + lk; RUN_WHILE(cls_L(c_)); return j; }
        }
        if (k != i && cls_L(c)) { /* cannot happen: prefix char is not a letter */ }
        if (cls_L(c)) { j = i + l; RUN_WHILE(cls_L(c_)); return j; }   /* backtrack: no prefix */
    }
    /* \p{N}{1,3} */
    if (cls_N(c)) {
        j = i + l; int cnt = 1;
        while (j < n && cnt < 3) { int l2; uint32_t c2 = u8dec(s, j, n, &l2); if (!cls_N(c2)) break; j += l2; cnt++; }
        return j;
    }
    /* " ?[^\s\p{L}\p{N}]+[\r\n]*" */
    {
        int64_t k = i;
        if (s[k] == ' ') k++;
        if (k < n) {
            int lk; uint32_t ck = u8dec(s, k, n, &lk);
            if (!cls_S(ck) && !cls_L(ck) && !cls_N(ck)) {
                j = k + lk; RUN_WHILE(!cls_S(c_) && !cls_L(c_) && !cls_N(c_));
                while (j < n && (s[j] == '\r' || s[j] == '\n')) j++;
                return j;
            }
        }
    }
    /* "\s*[\r\n]+": greedy \s* then backtrack so that [\r\n]+ matches: ends at the LAST CR/LF of the ws run */
    if (cls_S(c)) {
        int64_t run_end; j = i; RUN_WHILE(cls_S(c_)); run_end = j;
        int64_t last = -1;
        for (int64_t q = i; q < run_end; ++q) if (s[q] == '\r' || s[q] == '\n') last = q;
        if (last >= 0) return last + 1;
        /* "\s+(?!\S)" */
        if (run_end >= n) return run_end;
        int64_t prev = i, cur = i;
        while (cur < run_end) { prev = cur; cur += u8len(s[cur]); }
        if (prev > i) return prev;
        /* "\s+" */
        return run_end;
    }
    return i + l;
}

/* regex find_iter + SplitDelimiterBehavior::Isolated (normalizer.rs:694-783): both patterns cover every byte */
static void split_regex(const uint8_t* s, int64_t n, int which, splits* sp) {
    int64_t i = 0;
    while (i < n) {
        int64_t e = which == PT_GPT2 ? gpt2_match(s, i, n) : llama3_match(s, i, n);
        if (e <= i) e = i + u8len(s[i]);
        sp_push(sp, i, e);
        i = e;
    }
}

/* Whitespace: \w+|[^\w\s]+ matches kept, the rest removed (whitespace.rs:20-29) */
static void split_whitespace(const uint8_t* s, int64_t n, splits* sp) {
    int64_t i = 0;
    while (i < n) {
        int l; uint32_t c = u8dec(s, i, n, &l);
        uint8_t f = uc(c);
        int64_t j = i + l;
        if (f & UC_RX_W) { RUN_WHILE(uc(c_) & UC_RX_W); sp_push(sp, i, j); }
        else if (!(f & UC_RX_S)) { RUN_WHILE(!(uc(c_) & (UC_RX_W | UC_RX_S))); sp_push(sp, i, j); }
        i = j;
    }
}
/* WhitespaceSplit: char::is_whitespace, Removed (whitespace.rs:35-41) */
static void split_wssplit(const uint8_t* s, int64_t n, splits* sp) {
    int64_t i = 0, start = 0;
    while (i < n) {
        int l; uint32_t c = u8dec(s, i, n, &l);
        if (uc(c) & UC_RUST_WS) { sp_push(sp, start, i); start = i + l; }
        i += l;
    }
    sp_push(sp, start, n);
}
/* BertPreTokenizer: whitespace Removed, then punctuation Isolated (bert.rs:14-17) */
static void split_bert(const uint8_t* s, int64_t n, splits* sp) {
    int64_t i = 0, start = 0;
    while (i < n) {
        int l; uint32_t c = u8dec(s, i, n, &l);
        uint8_t f = uc(c);
        if (f & UC_RUST_WS) { sp_push(sp, start, i); start = i + l; }
        else if (f & UC_BERT_P) { sp_push(sp, start, i); sp_push(sp, i, i + l); start = i + l; }
        i += l;
    }
    sp_push(sp, start, n);
}

/* ------------------------------------------------------------------------------------------- */
/* BPE: merge_word + Word::merge_all                                                           */
/* ------------------------------------------------------------------------------------------- */
typedef struct { uint32_t c; int64_t prev, next; int64_t len; } symbol;
typedef struct { int64_t pos; uint32_t rank, new_id; } merge_item;
/* min-heap on (rank, pos): word.rs:28-36 */
static inline int mi_less(const merge_item* a, const merge_item* b) { return a->rank != b->rank ? a->rank < b->rank : a->pos < b->pos; }
typedef struct { merge_item* a; int64_t n, cap; } heap;
static void heap_push(heap* h, merge_item m) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (merge_item*)realloc(h->a, sizeof(merge_item) * h->cap); }
    int64_t i = h->n++;
    h->a[i] = m;
    while (i > 0) { int64_t p = (i - 1) / 2; if (!mi_less(&h->a[i], &h->a[p])) break; merge_item t = h->a[i]; h->a[i] = h->a[p]; h->a[p] = t; i = p; }
}
static int heap_pop(heap* h, merge_item* out) {
    if (!h->n) return 0;
    *out = h->a[0];
    h->a[0] = h->a[--h->n];
    int64_t i = 0;
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < h->n && mi_less(&h->a[l], &h->a[m])) m = l;
        if (r < h->n && mi_less(&h->a[r], &h->a[m])) m = r;
        if (m == i) break;
        merge_item t = h->a[i]; h->a[i] = h->a[m]; h->a[m] = t; i = m;
    }
    return 1;
}

typedef struct { uint32_t* id; int64_t* s; int64_t* e; int64_t n, cap; } toklist;
static void tl_push(toklist* t, uint32_t id, int64_t s, int64_t e) {
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 64;
        t->id = (uint32_t*)realloc(t->id, sizeof(uint32_t) * t->cap);
        t->s = (int64_t*)realloc(t->s, sizeof(int64_t) * t->cap);
        t->e = (int64_t*)realloc(t->e, sizeof(int64_t) * t->cap);
    }
    t->id[t->n] = id; t->s[t->n] = s; t->e[t->n] = e; t->n++;
}

/* Word::merge_all (bpe/word.rs:162-250), dropout = None */
static void merge_all(const oracle_tok* t, symbol* sym, int64_t nsym) {
    heap q = {0};
    for (int64_t i = 0; i + 1 < nsym; ++i) {
        uint32_t r, ni;
        if (pairmap_get(&t->merges, sym[i].c, sym[i + 1].c, &r, &ni)) { merge_item m = {i, r, ni}; heap_push(&q, m); }
    }
    merge_item top;
    while (heap_pop(&q, &top)) {
        if (sym[top.pos].len == 0) continue;
        if (sym[top.pos].next == -1) continue;
        int64_t next_pos = sym[top.pos].next;
        symbol right = sym[next_pos];
        uint32_t r, ni;
        if (!pairmap_get(&t->merges, sym[top.pos].c, right.c, &r, &ni) || ni != top.new_id) continue;   /* stale (word.rs:199-205) */
        sym[top.pos].c = top.new_id;                       /* merge_with */
        sym[top.pos].len += right.len;
        sym[top.pos].next = right.next;
        sym[next_pos].len = 0;
        if (right.next > -1 && right.next < nsym) sym[right.next].prev = top.pos;
        symbol* cur = &sym[top.pos];
        if (cur->prev >= 0) {
            if (pairmap_get(&t->merges, sym[cur->prev].c, cur->c, &r, &ni)) { merge_item m = {cur->prev, r, ni}; heap_push(&q, m); }
        }
        if (cur->next >= 0 && cur->next < nsym) {
            if (pairmap_get(&t->merges, cur->c, sym[cur->next].c, &r, &ni)) { merge_item m = {top.pos, r, ni}; heap_push(&q, m); }
        }
    }
    free(q.a);
}

/* BPE::tokenize on the NORMALIZED pre-token `w` (bpe/model.rs:558-612); token offsets are byte
 * offsets inside `w`.  Supports what the hot path supports: no dropout / prefix / suffix. */
static int bpe_tokenize(const oracle_tok* t, const uint8_t* w, int64_t n, toklist* out) {
    if (n == 0) return 0;
    uint32_t id;
    if (t->ignore_merges && strmap_get(&t->vocab, w, n, &id)) { tl_push(out, id, 0, n); return 0; }
    symbol* sym = (symbol*)malloc(sizeof(symbol) * (size_t)(n + 1));
    int64_t nsym = 0, i = 0;
    int have_unk = 0; uint32_t unk_id = 0; int64_t unk_len = 0;
    while (i < n) {                                        /* merge_word, bpe/model.rs:465-545 */
        int l = u8len(w[i]);
        if (i + l > n) l = (int)(n - i);
        if (strmap_get(&t->vocab, w + i, l, &id)) {
            if (have_unk) { symbol s = {unk_id, nsym - 1, -1, unk_len}; if (nsym) sym[nsym - 1].next = nsym; sym[nsym++] = s; have_unk = 0; }
            symbol s = {id, nsym - 1, -1, l};
            if (nsym) sym[nsym - 1].next = nsym;
            sym[nsym++] = s;
        } else if (t->has_unk) {
            /* fuse_unk=false: emit the previous unk, start a new one (model.rs:524-533) */
            if (have_unk) { symbol s = {unk_id, nsym - 1, -1, unk_len}; if (nsym) sym[nsym - 1].next = nsym; sym[nsym++] = s; }
            have_unk = 1; unk_id = t->unk_id; unk_len = l;
        }   /* no unk_token: the char is silently dropped */
        i += l;
    }
    if (have_unk) { symbol s = {unk_id, nsym - 1, -1, unk_len}; if (nsym) sym[nsym - 1].next = nsym; sym[nsym++] = s; }
    if (t->have_merges) merge_all(t, sym, nsym);
    int64_t pos = 0;
    for (int64_t k = 0; k < nsym; ++k) {                   /* word_to_tokens: running byte sum (word.rs:260-268) */
        if (sym[k].len == 0) continue;
        tl_push(out, sym[k].c, pos, pos + sym[k].len);
        pos += sym[k].len;
    }
    free(sym);
    return 0;
}

/* WordPiece::tokenize (wordpiece/mod.rs:224-283) */
static int wordpiece_tokenize(const oracle_tok* t, const uint8_t* w, int64_t n, toklist* out) {
    int64_t chars = 0;
    for (int64_t i = 0; i < n; i += u8len(w[i])) chars++;
    if (chars > t->max_input_chars) {
        if (!t->has_unk) return ORACLE_ERR_UNK;
        tl_push(out, t->unk_id, 0, n);
        return 0;
    }
    int64_t first = out->n;
    uint8_t* buf = (uint8_t*)malloc((size_t)(n + t->cont_prefix_len + 1));
    int bad = 0;
    int64_t start = 0;
    while (start < n) {
        int64_t end = n; int found = 0; uint32_t id = 0;
        while (start < end) {
            const uint8_t* k = w + start; int64_t kl = end - start;
            if (start > 0) { memcpy(buf, t->cont_prefix, (size_t)t->cont_prefix_len); memcpy(buf + t->cont_prefix_len, k, (size_t)kl); k = buf; kl += t->cont_prefix_len; }
            if (strmap_get(&t->vocab, k, kl, &id)) { found = 1; break; }
            /* end -= len_utf8(last char of substr) */
            int64_t e2 = end - 1;
            while (e2 > start && (w[e2] & 0xC0) == 0x80) e2--;
            end = e2;
        }
        if (!found) { bad = 1; break; }
        tl_push(out, id, start, end);
        start = end;
    }
    free(buf);
    if (bad) {
        out->n = first;
        if (!t->has_unk) return ORACLE_ERR_UNK;
        tl_push(out, t->unk_id, 0, n);
    }
    return 0;
}

/* WordLevel::tokenize (wordlevel/mod.rs:162-178) */
static int wordlevel_tokenize(const oracle_tok* t, const uint8_t* w, int64_t n, toklist* out) {
    uint32_t id;
    if (strmap_get(&t->vocab, w, n, &id)) { tl_push(out, id, 0, n); return 0; }
    if (!t->has_unk) return ORACLE_ERR_UNK;
    tl_push(out, t->unk_id, 0, n);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* one document                                                                                */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t* ids; uint32_t* offs; uint32_t* words; int64_t n, cap;
} enc_out;
static void eo_push(enc_out* o, uint32_t id, int64_t s, int64_t e, uint32_t w) {
    if (o->n == o->cap) {
        o->cap = o->cap ? o->cap * 2 : 256;
        o->ids = (uint32_t*)realloc(o->ids, sizeof(uint32_t) * o->cap);
        o->offs = (uint32_t*)realloc(o->offs, sizeof(uint32_t) * 2 * o->cap);
        o->words = (uint32_t*)realloc(o->words, sizeof(uint32_t) * o->cap);
    }
    o->ids[o->n] = id; o->offs[2 * o->n] = (uint32_t)s; o->offs[2 * o->n + 1] = (uint32_t)e; o->words[o->n] = w; o->n++;
}

/* BertNormalizer on ASCII (normalizers/bert.rs:92-138): drop 0x00-0x1F except \t\n\r and
 * 0x7F; \t\n\r -> ' '; lowercase.  orig[k] = original byte index of normalized byte k. */
static int bert_normalize_ascii(const uint8_t* s, int64_t n, uint8_t* out, int64_t* orig, int64_t* out_n, int lowercase, int clean) {
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint8_t b = s[i];
        if (b >= 0x80) return ORACLE_ERR_UNSUPPORTED;
        if (clean) {
            if (b == '\t' || b == '\n' || b == '\r') b = ' ';
            else if (b < 0x20 || b == 0x7F) continue;
        }
        if (lowercase && b >= 'A' && b <= 'Z') b += 32;
        out[m] = b; orig[m] = i; m++;
    }
    *out_n = m;
    return 0;
}

static int64_t count_chars(const uint8_t* s, int64_t pos) { int64_t c = 0; for (int64_t i = 0; i < pos; ++i) c += (s[i] & 0xC0) != 0x80; return c; }

/* Encode one document.  Offsets are BYTE offsets into the original document (OffsetType::Byte). */
static int encode_doc(const oracle_tok* t, const uint8_t* text, int64_t n, enc_out* out, splits* sp_out) {
    const uint8_t* s = text;
    uint8_t* nbuf = NULL; int64_t* orig = NULL; int64_t nn = n;
    int rc = 0;
    if (t->norm == N_BERT) {
        nbuf = (uint8_t*)malloc((size_t)n + 1); orig = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
        rc = bert_normalize_ascii(text, n, nbuf, orig, &nn, 1, 1);
        if (rc) { free(nbuf); free(orig); return rc; }
        s = nbuf;
    }
    /* ByteLevel add_prefix_space: prepend ' ' when the split does not start with one (byte_level.rs:122-125).
     * The prepended space shares the first original char's alignment (normalizer.rs:503-514). */
    uint8_t* pbuf = NULL; int prefixed = 0;
    if (t->byte_level && t->add_prefix_space && nn > 0 && s[0] != ' ') {
        pbuf = (uint8_t*)malloc((size_t)nn + 2);
        pbuf[0] = ' '; memcpy(pbuf + 1, s, (size_t)nn);
        s = pbuf; nn += 1; prefixed = 1;
    }
    splits sp = {0};
    switch (t->pretok) {
        case PT_GPT2: split_regex(s, nn, PT_GPT2, &sp); break;
        case PT_LLAMA3: split_regex(s, nn, PT_LLAMA3, &sp); break;
        case PT_WS: split_whitespace(s, nn, &sp); break;
        case PT_WSSPLIT: split_wssplit(s, nn, &sp); break;
        case PT_BERT: split_bert(s, nn, &sp); break;
        case PT_BL_NOREGEX: sp_push(&sp, 0, nn); break;
        default: rc = ORACLE_ERR_UNSUPPORTED;
    }
    toklist tl = {0};
    const int64_t doc_first_tok = out->n;
    uint8_t* mapped = NULL; int64_t* mo = NULL;
    for (int64_t k = 0; k < sp.n && !rc; ++k) {
        int64_t a = sp.s[k], b = sp.e[k];
        tl.n = 0;
        if (t->byte_level) {
            /* byte -> BYTES_CHAR[b] (byte_level.rs:132-146); mo[x] = byte index (in s) of mapped byte x */
            mapped = (uint8_t*)realloc(mapped, (size_t)(2 * (b - a) + 2));
            mo = (int64_t*)realloc(mo, sizeof(int64_t) * (size_t)(2 * (b - a) + 2));
            int64_t m = 0;
            for (int64_t x = a; x < b; ++x) {
                uint32_t cp = t->b2c[s[x]];
                if (cp < 0x80) { mapped[m] = (uint8_t)cp; mo[m++] = x; }
                else { mapped[m] = (uint8_t)(0xC0 | (cp >> 6)); mo[m++] = x; mapped[m] = (uint8_t)(0x80 | (cp & 0x3F)); mo[m++] = x; }
            }
            mo[m] = b;
            rc = bpe_tokenize(t, mapped, m, &tl);
            for (int64_t q = 0; q < tl.n && !rc; ++q) {
                /* normalized offsets -> bytes of s: first byte of first mapped char .. one past the last */
                int64_t bs = mo[tl.s[q]], be = (tl.e[q] > 0 ? mo[tl.e[q] - 1] + 1 : bs);
                /* a token covering part of a multi-byte original char reports the whole char (byte_level.rs:135-143) */
                while (bs > 0 && (s[bs] & 0xC0) == 0x80) bs--;
                while (be < nn && (s[be] & 0xC0) == 0x80) be++;
                if (prefixed) {
                    /* s = ' ' + original; the prepended space shares the first original char's alignment
                     * (normalizer.rs:503-514): position 0 maps to [0, len(first char)) */
                    int64_t first_len = u8len(s[1]);
                    bs = bs == 0 ? 0 : bs - 1;
                    be = be <= 1 ? first_len : be - 1;
                }
                if (t->char_offsets) { bs = count_chars(text, bs); be = count_chars(text, be); }
                if (t->trim_offsets) {
                    /* ByteLevel post-processor process_offsets (byte_level.rs:202-234): strip leading / trailing
                     * 'G-dot' (mapped space) chars of the token from its offsets */
                    int64_t ts = mo[tl.s[q]], te = (tl.e[q] > 0 ? mo[tl.e[q] - 1] + 1 : ts);
                    int64_t lead = 0, trail = 0;
                    while (ts + lead < te && s[ts + lead] == ' ') lead++;
                    while (trail < te - ts && s[te - 1 - trail] == ' ') trail++;
                    if (lead > 0) {
                        int is_first = (out->n == doc_first_tok) || bs == 0;
                        if (is_first && t->pp_add_prefix_space && lead == 1) lead = 0;
                        bs = bs + lead < be ? bs + lead : be;
                    }
                    if (trail > 0 && be >= trail) be = be - trail > bs ? be - trail : bs;
                }
                eo_push(out, tl.id[q], bs, be, (uint32_t)k);
            }
        } else {
            const uint8_t* w = s + a; int64_t wl = b - a;
            if (t->model == M_WORDPIECE) rc = wordpiece_tokenize(t, w, wl, &tl);
            else if (t->model == M_WORDLEVEL) rc = wordlevel_tokenize(t, w, wl, &tl);
            else rc = ORACLE_ERR_UNSUPPORTED;
            for (int64_t q = 0; q < tl.n && !rc; ++q) {
                int64_t bs = a + tl.s[q], be = a + tl.e[q];
                if (orig) { int64_t ob = orig[bs]; int64_t oe = orig[be - 1] + 1; bs = ob; be = oe; }
                if (t->char_offsets) { bs = count_chars(text, bs); be = count_chars(text, be); }
                eo_push(out, tl.id[q], bs, be, (uint32_t)k);
            }
        }
    }
    if (sp_out) {
        for (int64_t k = 0; k < sp.n; ++k) {
            int64_t a = sp.s[k], b = sp.e[k];
            if (prefixed) { a = a > 0 ? a - 1 : 0; b = b > 0 ? b - 1 : 0; }
            if (orig) { a = orig[sp.s[k]]; b = orig[sp.e[k] - 1] + 1; }
            if (sp_out->n == sp_out->cap) {
                sp_out->cap = sp_out->cap ? sp_out->cap * 2 : 64;
                sp_out->s = (int64_t*)realloc(sp_out->s, sizeof(int64_t) * sp_out->cap);
                sp_out->e = (int64_t*)realloc(sp_out->e, sizeof(int64_t) * sp_out->cap);
            }
            sp_out->s[sp_out->n] = a; sp_out->e[sp_out->n] = b; sp_out->n++;
        }
    }
    free(tl.id); free(tl.s); free(tl.e); free(mapped); free(mo);
    free(sp.s); free(sp.e); free(pbuf); free(nbuf); free(orig);
    return rc;
}

/* ------------------------------------------------------------------------------------------- */
/* batch API                                                                                   */
/* ------------------------------------------------------------------------------------------- */
typedef struct oracle_batch {
    int64_t n_docs, n_tokens;
    uint32_t* ids; uint32_t* offs; uint32_t* words; int64_t* tok_offsets;
} oracle_batch;

/* TokenizerImpl::encode_batch (tokenizer/mod.rs:1337-1356), serial map over documents. */
int oracle_encode_batch(oracle_tok* t, const uint8_t* text, const int64_t* doc_off, int64_t n_docs, oracle_batch** outp) {
    enc_out eo = {0};
    int64_t* to = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_docs + 1));
    to[0] = 0;
    for (int64_t d = 0; d < n_docs; ++d) {
        int64_t base = eo.n;
        int rc = encode_doc(t, text + doc_off[d], doc_off[d + 1] - doc_off[d], &eo, NULL);
        if (rc) { free(eo.ids); free(eo.offs); free(eo.words); free(to); return rc; }
        (void)base;
        to[d + 1] = eo.n;
    }
    oracle_batch* b = (oracle_batch*)calloc(1, sizeof(oracle_batch));
    b->n_docs = n_docs; b->n_tokens = eo.n; b->ids = eo.ids; b->offs = eo.offs; b->words = eo.words; b->tok_offsets = to;
    *outp = b;
    return 0;
}
int64_t oracle_batch_n_tokens(const oracle_batch* b) { return b->n_tokens; }
const uint32_t* oracle_batch_ids(const oracle_batch* b) { return b->ids; }
const uint32_t* oracle_batch_offsets(const oracle_batch* b) { return b->offs; }
const uint32_t* oracle_batch_words(const oracle_batch* b) { return b->words; }
const int64_t* oracle_batch_tok_offsets(const oracle_batch* b) { return b->tok_offsets; }
void oracle_batch_free(oracle_batch* b) { if (!b) return; free(b->ids); free(b->offs); free(b->words); free(b->tok_offsets); free(b); }

/* pre-tokenizer only: splits of one document as (start,end) byte offsets into the original */
int64_t oracle_pre_tokenize(oracle_tok* t, const uint8_t* text, int64_t n, int64_t* out, int64_t cap) {
    /* run the split stage only: reuse encode_doc with a model-less pass */
    splits sp = {0};
    enc_out eo = {0};
    oracle_tok tmp = *t;
    int rc = encode_doc(&tmp, text, n, &eo, &sp);
    free(eo.ids); free(eo.offs); free(eo.words);
    if (rc && rc != ORACLE_ERR_UNK) { free(sp.s); free(sp.e); return rc; }
    int64_t m = sp.n;
    for (int64_t k = 0; k < m && k < cap; ++k) { out[2 * k] = sp.s[k]; out[2 * k + 1] = sp.e[k]; }
    free(sp.s); free(sp.e);
    return m;
}

/* model only: Model::tokenize on one already-normalized pre-token string (the per-pre-token oracle) */
int64_t oracle_model_tokenize(oracle_tok* t, const uint8_t* w, int64_t n, uint32_t* ids, int64_t* offs, int64_t cap) {
    toklist tl = {0};
    int rc;
    if (t->model == M_BPE) rc = bpe_tokenize(t, w, n, &tl);
    else if (t->model == M_WORDPIECE) rc = wordpiece_tokenize(t, w, n, &tl);
    else rc = wordlevel_tokenize(t, w, n, &tl);
    if (rc) { free(tl.id); free(tl.s); free(tl.e); return rc; }
    int64_t m = tl.n;
    for (int64_t k = 0; k < m && k < cap; ++k) { ids[k] = tl.id[k]; offs[2 * k] = tl.s[k]; offs[2 * k + 1] = tl.e[k]; }
    free(tl.id); free(tl.s); free(tl.e);
    return m;
}
