// HIP kernels of the encode_batch hot path for gfx950 (MI355X, CDNA4, wave64).
//
// Flat, structure-of-arrays pipeline over ONE concatenated UTF-8 buffer (no per-document objects):
//
//   text bytes ──documents─────► doc-start bitmask (+ added-token matches, prefix space)          kernels/documents.hip
//        │──────bert_norm──────► normalised text X (BertNormalizer)                               kernels/bert_norm.hip
//        │──────pretok_*───────► pre-token start (/ end) bitmask: per-lane 64-bit mask algebra    kernels/pretok_{gpt2,llama3,local}.hip
//        │──────scan_emit──────► pt_start[P+1]   (byte offset of every pre-token = "split")       kernels/scan_emit.hip
//        │──────lookup─────────► straight from the bitmasks: whole-word hit -> 1 token (LDS hot    kernels/lookup.hip
//        │                       table, then the perfect hash in HBM); misses queued by length class
//        │──────bpe────────────► min-rank merge loop, one lane per queued pre-token (Word in LDS)  kernels/bpe.hip, bpe_huge.hip
//        │──────word_models────► WordPiece trie walk of the queued words                          kernels/word_models.hip
//        │──────scan_emit──────► pt_start[P+1] only when offsets / word ids are requested          kernels/scan_emit.hip
//        │──────output─────────► ids[T] (single-pass look-back compaction), per-document CSR,      kernels/output.hip, results.hip
//        │                       offsets, word ids, specials
//   ids ────────decode─────────► text (decode_batch)                                              kernels/decode.hip
//
// Every kernel cites the reference code it replaces.  All results are bit-exact integers.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <cstdint>

#include "device_utils.hpp"
#include "kernels.hpp"
#include "overflow_core.hpp"
#include "bert_norm_core.hpp"
#include "pretok_gpt2_core.hpp"
#include "pretok_l3_core.hpp"
#include "pretok_local_core.hpp"
#include "tables.hpp"

namespace tkamd {

// ---- small helpers shared by several kernels ----
__device__ __forceinline__ uint32_t uc_flags(uint32_t cp, const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2) {
    if (cp >= 0x110000u) return 0;
    return uc2[((uint32_t)uc1[cp >> 8] << 8) | (cp & 255u)];
}

struct __attribute__((packed, aligned(1))) Unaligned4 { uint32_t v; };
struct __attribute__((packed, aligned(1))) Unaligned16 { uint32_t a, b, c, d; };      // 16-byte global access at any alignment
// Streaming accesses (`nt` on the instruction): what a kernel reads or writes ONCE should not push what it keeps probing -- the word
// table, the claims -- out of the L2.  (The host build of the tests has no such thing: plain accesses.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ Unaligned16 load_nt16(const uint8_t* p) { const nt_u32x4 v = __builtin_nontemporal_load((const nt_u32x4*)p); return Unaligned16{v.x, v.y, v.z, v.w}; }
template <class T> __device__ __forceinline__ T load_nt(const T* p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void store_nt(T* p, T v) { __builtin_nontemporal_store(v, p); }
typedef uint32_t nt_a_u32x4 __attribute__((ext_vector_type(4)));          // (HIP's uint4 / uint2 are structs: the builtin wants vectors)
typedef uint32_t nt_a_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 load_nt(const uint4* p) { const nt_a_u32x4 v = __builtin_nontemporal_load((const nt_a_u32x4*)p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 load_nt(const uint2* p) { const nt_a_u32x2 v = __builtin_nontemporal_load((const nt_a_u32x2*)p); return make_uint2(v.x, v.y); }
__device__ __forceinline__ void store_nt(uint2* p, uint2 v) { nt_a_u32x2 x; x.x = v.x; x.y = v.y; __builtin_nontemporal_store(x, (nt_a_u32x2*)p); }
#else
__device__ __forceinline__ Unaligned16 load_nt16(const uint8_t* p) { return *(const Unaligned16*)p; }
template <class T> __device__ __forceinline__ T load_nt(const T* p) { return *p; }
template <class T> __device__ __forceinline__ void store_nt(T* p, T v) { *p = v; }
#endif
// decode the code point whose lead byte is text[i] (text has TKAMD_TEXT_PAD readable slack)
__device__ __forceinline__ uint32_t utf8_global(const uint8_t* __restrict__ text, int64_t i, uint32_t* len) {
    uint32_t w = ((const Unaligned4*)(text + i))->v;
    uint32_t b0 = w & 0xFFu, b1 = (w >> 8) & 0x3Fu, b2 = (w >> 16) & 0x3Fu, b3 = (w >> 24) & 0x3Fu;
    if (b0 < 0x80u) { *len = 1; return b0; }
    if (b0 < 0xE0u) { *len = 2; return ((b0 & 0x1Fu) << 6) | b1; }
    if (b0 < 0xF0u) { *len = 3; return ((b0 & 0x0Fu) << 12) | (b1 << 6) | b2; }
    *len = 4;
    return ((b0 & 0x07u) << 18) | (b1 << 12) | (b2 << 6) | b3;
}
static inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }   // grid size

// ---- the kernels, by pipeline stage (one translation unit; every file below is a plain slice of it) ----
#include "kernels/results.hip"
#include "kernels/scan_util.hip"
#include "kernels/documents.hip"
#include "kernels/pretok_gpt2.hip"
#include "kernels/pretok_llama3.hip"
#include "kernels/pretok_local.hip"
#include "kernels/bert_norm.hip"
#include "kernels/scan_emit.hip"
#include "kernels/bpe.hip"
#include "kernels/lookup.hip"
#include "kernels/word_models.hip"
#include "kernels/bpe_huge.hip"
#include "kernels/output.hip"
#include "kernels/epilogue.hip"
#include "kernels/decode.hip"
#include "kernels/launch.hip"

}  // namespace tkamd
