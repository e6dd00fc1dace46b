// CPU harness for tokenizers_amd/csrc/pretok_l3_core.hpp: runs l3_window_starts -- the exact function every lane of
// k_pretok_llama3_lane executes -- over a whole batch, one 64-byte window per 32 bytes of text, the way the kernel
// tiles it.  Built and driven by tests/test_pretok_core.py (g++, no GPU).
#include <cstdint>
#include <cstring>
#include <vector>

#include "host_model.hpp"
#include "pretok_gpt2_core.hpp"
#include "pretok_l3_core.hpp"
#include "pretok_local_core.hpp"

using namespace tkamd;

extern "C" int l3h_run(const char* json, size_t json_len, const uint8_t* text, int64_t n, const int64_t* doc_off, int64_t n_docs,
                       uint8_t* start_out, uint8_t* unres_out) {
    HostModel hm;
    try {
        hm = HostModel::from_json(json, json_len);
    } catch (const std::exception&) {
        return -1;
    }
    std::vector<uint8_t> docstart((size_t)n + 64, 0);
    for (int64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) docstart[doc_off[d]] = 1;
    // padded copy: the kernel's text carries 64 readable bytes after its end and the first window starts 16 bytes early
    std::vector<uint8_t> buf((size_t)n + 64 + 128, 0);
    uint8_t* t = buf.data() + 64;
    memcpy(t, text, (size_t)n);
    L3Flags lut[256];
    for (uint32_t v = 0; v < 256; ++v) lut[v] = l3_byte_flags(v);
    for (int64_t a = 0; a < n; a += L3W_MAIN) {
        const int64_t base = a - L3W_HALO;
        L3Window w{};
        for (int i = 0; i < 64; ++i) {
            const int64_t g = base + i;
            if (g < 0 || g >= n) continue;
            const L3Flags f = lut[t[g]];
            const uint64_t bit = 1ull << i;
            w.V |= bit;
            if (f.x & 1u) w.L |= bit;
            if (f.x & (1u << 8)) w.N |= bit;
            if (f.x & (1u << 16)) w.W |= bit;
            if (f.x & (1u << 24)) w.R |= bit;
            if (f.y & 1u) w.SP |= bit;
            if (f.y & (1u << 8)) w.C |= bit;
            if (f.y & (1u << 16)) w.AP |= bit;
            if (f.y & (1u << 24)) w.MU |= bit;
            if (docstart[g]) w.D |= bit;
        }
        uint64_t st = 0, un = 0;
        if (split_rule_fast_cs(hm.split_rule)) {
            for (int i = 0; i < 64; ++i)
                if (base + i >= 0 && base + i < n && (t[base + i] & 0x20u)) w.B5 |= 1ull << i;
            l3_window_starts_cs(w, t, base, hm.uc_stage1.data(), hm.uc_stage2.data(), hm.ucc_stage1.data(), hm.ucc_stage2.data(), &st, &un, hm.split_rule);
        } else
        l3_window_starts(w, t, base, hm.uc_stage1.data(), hm.uc_stage2.data(), &st, &un, hm.split_rule);      // (the member of the family the JSON names)
        for (int i = L3W_HALO; i < L3W_HALO + L3W_MAIN; ++i) {
            const int64_t g = base + i;
            if (g < n) { start_out[g] = (st >> i) & 1; unres_out[g] = (un >> i) & 1; }
        }
    }
    return 0;
}

// ---- pretok_local_core.hpp: Whitespace / WhitespaceSplit / BertPreTokenizer, 48 bytes per window ----------------------
template <int KIND>
static void pl_run(const HostModel& hm, const uint8_t* t, int64_t n, const std::vector<uint8_t>& docstart, uint8_t* start_out, uint8_t* end_out) {
    uint32_t lut[256];
    for (uint32_t v = 0; v < 256; ++v) lut[v] = local_byte_flags<KIND>(v);
    for (int64_t a = 0; a <= n; a += PLW_MAIN) {
        const int64_t base = a - PLW_HALO;
        LocalWindow w{};
        for (int i = 0; i < 64; ++i) {
            const int64_t g = base + i;
            if (g == n) w.END = 1ull << i;
            if (g < 0 || g >= n) continue;
            const uint32_t f = lut[t[g]];
            const uint64_t bit = 1ull << i;
            w.V |= bit;
            const bool multi = (f & 0x010101u) == 0x010101u;
            if (multi) w.MU |= bit;
            else {
                if (f & 1u) w.C1 |= bit;
                if (f & (1u << 8)) w.C2 |= bit;
                if (f & (1u << 16)) w.C3 |= bit;
            }
            if (f & (1u << 24)) w.C |= bit;
            if (docstart[g]) w.D |= bit;
        }
        uint64_t st = 0, en = 0;
        local_window_masks<KIND>(w, t, base, hm.uc_stage1.data(), hm.uc_stage2.data(), &st, &en);
        for (int i = PLW_HALO; i < PLW_HALO + PLW_MAIN; ++i) {
            const int64_t g = base + i;
            if (g <= n) { start_out[g] = (st >> i) & 1; end_out[g] = (en >> i) & 1; }
        }
    }
}

extern "C" int plh_run(const char* json, size_t json_len, const uint8_t* text, int64_t n, const int64_t* doc_off, int64_t n_docs,
                       uint8_t* start_out, uint8_t* end_out) {
    HostModel hm;
    try {
        hm = HostModel::from_json(json, json_len);
    } catch (const std::exception&) {
        return -1;
    }
    std::vector<uint8_t> docstart((size_t)n + 64, 0);
    for (int64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) docstart[doc_off[d]] = 1;
    std::vector<uint8_t> buf((size_t)n + 64 + 128, 0);
    uint8_t* t = buf.data() + 64;
    memcpy(t, text, (size_t)n);
    if (hm.pretok == PT_WHITESPACE) pl_run<PT_WHITESPACE>(hm, t, n, docstart, start_out, end_out);
    else if (hm.pretok == PT_WHITESPACE_SPLIT) pl_run<PT_WHITESPACE_SPLIT>(hm, t, n, docstart, start_out, end_out);
    else if (hm.pretok == PT_BERT) pl_run<PT_BERT>(hm, t, n, docstart, start_out, end_out);
    else return -2;
    return 0;
}

// ---- pretok_gpt2_core.hpp: GPT-2 ByteLevel split.  gpt2_lane_starts is everything a lane of k_pretok_gpt2_seq does (loads,
// flag deposit, masks, algebra); the loop below adds what the kernel adds: the LUT, the document-start bitmask, and the
// "four lanes' 48 bits are three 64-bit words" assembly (__shfl_down by one lane).
extern "C" int g2h_run(const char* json, size_t json_len, const uint8_t* text, int64_t n, const int64_t* doc_off, int64_t n_docs,
                       uint8_t* start_out) {
    HostModel hm;
    try {
        hm = HostModel::from_json(json, json_len);
    } catch (const std::exception&) {
        return -1;
    }
    const int64_t n_words = (n >> 6) + 1;
    std::vector<uint64_t> docmask((size_t)n_words + 1, 0), startmask((size_t)n_words + 1, 0);
    for (int64_t d = 0; d < n_docs; ++d)
        if (doc_off[d] < n) docmask[doc_off[d] >> 6] |= 1ull << (doc_off[d] & 63);
    std::vector<uint8_t> buf((size_t)n + 64 + 128, 0xEE);          // garbage before the text, zero pad after it is NOT assumed either
    uint8_t* t = buf.data() + 64;
    memcpy(t, text, (size_t)n);
    Gpt2Flags lut[256];
    for (uint32_t v = 0; v < 256; ++v) lut[v] = gpt2_byte_flags(v);
    const int64_t n_lanes = ((n + 1 + 256 * G2W_MAIN - 1) / (256 * G2W_MAIN)) * 256;       // the kernel's grid
    std::vector<uint64_t> out((size_t)n_lanes + 1, 0);
    for (int64_t lane = 0; lane < n_lanes; ++lane)
        out[lane] = gpt2_lane_starts(t, n, n_words, docmask.data(), lut, lane, hm.uc_stage1.data(), hm.uc_stage2.data());
    for (int64_t lane = 0; lane < n_lanes; ++lane) {
        const int q = (int)(lane & 3);
        if (q == 3) continue;
        const int64_t word = 3 * (lane >> 2) + q;
        const uint64_t nxt = (lane & 63) == 63 ? 0 : out[lane + 1];                         // __shfl_down stays inside the wavefront
        if (word < n_words) startmask[word] = (out[lane] >> (16 * q)) | (nxt << (G2W_MAIN - 16 * q));
    }
    for (int64_t g = 0; g < n; ++g) start_out[g] = (startmask[g >> 6] >> (g & 63)) & 1;
    return 0;
}
