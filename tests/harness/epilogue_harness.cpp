// CPU harness for tokenizers_amd/csrc/kernels/{scan_util,epilogue}.hip: the epilogue kernels and their launchers -- special tokens,
// truncation with its overflowing encodings, padding, pairs -- compiled for the host, UNCHANGED, under the SIMT shim of
// tests/harness/simt/ and driven in the order csrc/capi.cpp drives them (run_pipeline: add_specials / finalize / finalize_pairs).
// Built and used by tests/test_epilogue_core.py: input = the wheel's plain encodings, expected output = the wheel's truncated /
// padded / overflowing / pair encodings.  Test infrastructure; nothing in the product includes this.
#include <hip/hip_runtime.h>   // the shim (-I tests/harness/simt)

#include <cstdint>
#include <vector>

#include "device_utils.hpp"
#include "kernels.hpp"
#include "overflow_core.hpp"
#include "tables.hpp"

namespace tkamd {
static inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }
#include "kernels/scan_util.hip"
#include "kernels/epilogue.hip"
}  // namespace tkamd

using namespace tkamd;

namespace {
struct Result {
    std::vector<int64_t> tok_offsets2;
    std::vector<uint32_t> ids2, offsets2, word_ids2, pad_count, enc_doc, enc_parts;
    std::vector<uint8_t> type_ids2, seq_ids2;
    int64_t n_enc = 0, n_tok = 0;
    int err = 0;
} R;
constexpr int GRID = 3;         // workgroups of the copy kernels (the product uses 8 per CU)
}  // namespace

extern "C" {

// params[]: 0 add_special, 1 trunc_on, 2 trunc_max_length, 3 trunc_stride, 4 trunc_left, 5 trunc_strategy (0 LongestFirst 1 OnlyFirst 2 OnlySecond),
//           6 pad_on, 7 pad_fixed, 8 pad_length, 9 pad_multiple, 10 pad_left, 11 pad_id, 12 pad_type_id, 13 want_overflow
// single sequences: capi.cpp run_pipeline, lambdas add_specials / finalize
int epi_single(const int64_t* tok_offsets, int64_t n_docs, const uint32_t* ids, const uint32_t* offsets, const uint32_t* word_ids,
               const uint32_t* prefix, int32_t n_prefix, const uint32_t* suffix, int32_t n_suffix, const uint32_t* params) {
    R = Result{};
    const bool add_special = params[0] && (n_prefix || n_suffix), trunc_on = params[1], pad_on = params[6];
    const int64_t T = tok_offsets[n_docs];
    if (!trunc_on && !pad_on) {                              // specials only
        const size_t T2 = (size_t)T + 4 + (size_t)(n_docs + 1) * (size_t)(n_prefix + n_suffix);
        R.tok_offsets2.assign((size_t)n_docs + 2, 0); R.ids2.assign(T2, 0xDEADBEEFu); R.offsets2.assign(2 * T2, 0xDEADBEEFu); R.word_ids2.assign(T2, 0xDEADBEEFu);
        SpecialArgs sa{};
        sa.tok_offsets = tok_offsets; sa.n_docs = n_docs; sa.ids = ids; sa.offsets = offsets; sa.word_ids = word_ids;
        sa.prefix = prefix; sa.suffix = suffix; sa.n_prefix = add_special ? n_prefix : 0; sa.n_suffix = add_special ? n_suffix : 0;
        sa.tok_offsets2 = R.tok_offsets2.data(); sa.ids2 = R.ids2.data(); sa.offsets2 = R.offsets2.data(); sa.word_ids2 = R.word_ids2.data(); sa.n_tok2 = &R.n_tok;
        launch_add_specials(nullptr, GRID, sa);
        R.n_enc = n_docs;
        return 0;
    }
    const uint32_t n_add = add_special ? (uint32_t)(n_prefix + n_suffix) : 0u;
    uint32_t target = 0;
    FinalArgs fa{};
    fa.tok_offsets = tok_offsets; fa.n_docs = n_docs; fa.ids = ids; fa.offsets = offsets; fa.word_ids = word_ids;
    fa.prefix = prefix; fa.suffix = suffix; fa.n_prefix = add_special ? n_prefix : 0; fa.n_suffix = add_special ? n_suffix : 0;
    fa.trunc_len = 0xFFFFFFFFu;
    if (trunc_on) fa.trunc_len = (n_add && params[2] < n_add) ? 0xFFFFFFFFu : params[2] - n_add;
    fa.trunc_left = params[4]; fa.trunc_needs_pair = (trunc_on && params[5] == 2) ? 1u : 0u; fa.trunc_stride = params[3];
    fa.pad_on = pad_on; fa.pad_fixed = params[7]; fa.pad_length = params[8]; fa.pad_multiple = params[9]; fa.pad_left = params[10]; fa.pad_id = params[11];
    std::vector<uint32_t> bsum((size_t)(n_docs + 1) / 256 + 2), parts, len1, fin, enc_start, enc_cnt;
    std::vector<int64_t> enc_base;
    fa.bsum = bsum.data(); fa.target = &target; fa.n_tok2 = &R.n_tok; fa.err = &R.err;
    int64_t n_enc = n_docs;
    const bool overflow = params[13] && trunc_on;
    if (overflow) {
        parts.assign((size_t)n_docs + 2, 0); enc_base.assign((size_t)n_docs + 2, 0);
        fa.ovf_parts = parts.data(); fa.enc_base = enc_base.data();
        launch_overflow_count(nullptr, fa, &n_enc);
        R.enc_doc.assign((size_t)n_enc + 2, 0xDEADBEEFu); enc_start.assign((size_t)n_enc + 2, 0); enc_cnt.assign((size_t)n_enc + 2, 0);
        fa.enc_doc = R.enc_doc.data(); fa.enc_start = enc_start.data(); fa.enc_cnt = enc_cnt.data();
        bsum.assign((size_t)(n_enc + 1) / 256 + 2, 0);
        fa.bsum = bsum.data();
    }
    len1.assign((size_t)n_enc + 2, 0); fin.assign((size_t)n_enc + 2, 0); R.tok_offsets2.assign((size_t)n_enc + 2, 0);
    if (pad_on) R.pad_count.assign((size_t)n_enc + 2, 0);
    fa.len1 = len1.data(); fa.fin = fin.data(); fa.tok_offsets2 = R.tok_offsets2.data(); fa.pad_count = pad_on ? R.pad_count.data() : nullptr;
    if (overflow) { launch_overflow_ranges(nullptr, fa); fa.n_docs = n_enc; }
    else launch_final_lens(nullptr, fa);
    launch_final_offsets(nullptr, fa);
    const size_t T2 = (size_t)R.n_tok + 4;
    R.ids2.assign(T2, 0xDEADBEEFu); R.offsets2.assign(2 * T2, 0xDEADBEEFu); R.word_ids2.assign(T2, 0xDEADBEEFu);
    fa.ids2 = R.ids2.data(); fa.offsets2 = R.offsets2.data(); fa.word_ids2 = R.word_ids2.data();
    launch_finalize(nullptr, GRID, fa);
    R.n_enc = n_enc;
    if (!overflow) R.enc_doc.clear();
    return R.err;
}

// pairs: documents 2i / 2i + 1 are sequence A / B (capi.cpp run_pipeline, lambda finalize_pairs); tpl = [n_tpl][3] kind, id, type id
int epi_pair(const int64_t* tok_offsets, int64_t n_pairs, const uint32_t* ids, const uint32_t* offsets, const uint32_t* word_ids,
             const uint32_t* tpl, int32_t n_tpl, const uint32_t* params) {
    R = Result{};
    uint32_t n_special = 0, target = 0;
    for (int k = 0; k < n_tpl; ++k) n_special += tpl[3 * k] == 2u;
    PairArgs pa{};
    pa.tok_offsets = tok_offsets; pa.n_pairs = n_pairs; pa.ids = ids; pa.offsets = offsets; pa.word_ids = word_ids;
    pa.tpl = tpl; pa.n_tpl = n_tpl; pa.n_special = n_special;
    pa.trunc_on = params[1]; pa.trunc_max = params[2]; pa.trunc_left = params[4]; pa.trunc_strategy = params[5]; pa.trunc_stride = params[3];
    pa.pad_on = params[6]; pa.pad_fixed = params[7]; pa.pad_length = params[8]; pa.pad_multiple = params[9]; pa.pad_left = params[10]; pa.pad_id = params[11];
    pa.pad_type_id = params[12];
    pa.ovf_ty_tpl = params[14];
    std::vector<uint32_t> keep((size_t)2 * n_pairs + 2), len1((size_t)n_pairs + 2), fin, bsum((size_t)(n_pairs + 1) / 256 + 2), parts, enc_win;
    std::vector<int64_t> enc_base;
    pa.keep = keep.data(); pa.bsum = bsum.data(); pa.target = &target; pa.n_tok2 = &R.n_tok; pa.err = &R.err;
    for (int k = 0; k < n_tpl; ++k) if (tpl[3 * k] < 2u) { pa.first_is_b = tpl[3 * k] == 1u; break; }
    const bool overflow = params[13] && pa.trunc_on;
    int64_t n_enc = n_pairs;
    if (overflow) {
        parts.assign((size_t)n_pairs + 2, 0); enc_base.assign((size_t)n_pairs + 2, 0);
        pa.ovf_parts = parts.data(); pa.enc_base = enc_base.data();
    } else pa.len1 = len1.data();
    launch_pair_lens(nullptr, pa);
    if (overflow) {
        launch_pair_overflow_scan(nullptr, pa, &n_enc);
        R.enc_doc.assign((size_t)n_enc + 2, 0xDEADBEEFu); R.enc_parts.assign(2 * ((size_t)n_enc + 2), 0xDEADBEEFu); enc_win.assign(4 * ((size_t)n_enc + 2), 0);
        len1.assign((size_t)n_enc + 2, 0); bsum.assign((size_t)(n_enc + 1) / 256 + 2, 0);
        pa.enc_doc = R.enc_doc.data(); pa.enc_idx = R.enc_parts.data(); pa.enc_win = enc_win.data(); pa.len1 = len1.data(); pa.bsum = bsum.data();
    }
    fin.assign((size_t)n_enc + 2, 0);
    R.tok_offsets2.assign((size_t)n_enc + 2, 0);
    if (pa.pad_on) R.pad_count.assign((size_t)n_enc + 2, 0);
    pa.fin = fin.data(); pa.tok_offsets2 = R.tok_offsets2.data(); pa.pad_count = pa.pad_on ? R.pad_count.data() : nullptr;
    if (overflow) launch_pair_ranges(nullptr, pa);
    FinalArgs fa{};
    fa.n_docs = n_enc; fa.len1 = pa.len1; fa.fin = pa.fin; fa.bsum = pa.bsum; fa.target = pa.target; fa.tok_offsets2 = pa.tok_offsets2; fa.n_tok2 = pa.n_tok2;
    fa.pad_on = pa.pad_on; fa.pad_fixed = pa.pad_fixed; fa.pad_length = pa.pad_length; fa.pad_multiple = pa.pad_multiple;
    launch_final_offsets(nullptr, fa);
    const size_t T2 = (size_t)R.n_tok + 4;
    R.ids2.assign(T2, 0xDEADBEEFu); R.offsets2.assign(2 * T2, 0xDEADBEEFu); R.word_ids2.assign(T2, 0xDEADBEEFu); R.type_ids2.assign(T2, 0xEE); R.seq_ids2.assign(T2, 0xEE);
    pa.ids2 = R.ids2.data(); pa.offsets2 = R.offsets2.data(); pa.word_ids2 = R.word_ids2.data(); pa.type_ids2 = R.type_ids2.data(); pa.seq_ids2 = R.seq_ids2.data();
    if (overflow) pa.n_pairs = n_enc;
    launch_pair_finalize(nullptr, GRID, pa);
    R.n_enc = n_enc;
    if (!overflow) { R.enc_doc.clear(); R.enc_parts.clear(); }
    return R.err;
}

int64_t epi_n_enc(void) { return R.n_enc; }
int64_t epi_n_tok(void) { return R.n_tok; }
const int64_t* epi_tok_offsets(void) { return R.tok_offsets2.data(); }
const uint32_t* epi_ids(void) { return R.ids2.data(); }
const uint32_t* epi_offsets(void) { return R.offsets2.data(); }
const uint32_t* epi_word_ids(void) { return R.word_ids2.data(); }
const uint32_t* epi_pad_count(void) { return R.pad_count.empty() ? nullptr : R.pad_count.data(); }
const uint32_t* epi_enc_doc(void) { return R.enc_doc.empty() ? nullptr : R.enc_doc.data(); }
const uint32_t* epi_enc_parts(void) { return R.enc_parts.empty() ? nullptr : R.enc_parts.data(); }
const uint8_t* epi_type_ids(void) { return R.type_ids2.empty() ? nullptr : R.type_ids2.data(); }
const uint8_t* epi_seq_ids(void) { return R.seq_ids2.empty() ? nullptr : R.seq_ids2.data(); }

}  // extern "C"
