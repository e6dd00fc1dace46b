// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Token compaction, offsets / word ids, special tokens.

// =================================================================================================
// Token compaction: tok0[P] (+ rows) -> ids[T], the running token offset of every pre-token, the total.
// Replaces: PreTokenizedString::into_encoding + Encoding::from_iter (tokenizer/pre_tokenizer.rs:198-263,
// tokenizer/encoding.rs:541-562) for the whole batch at once.
// ONE pass over the data with decoupled look-back (results.hip), software-pipelined so that nobody waits for it:
//   front(c)   load the chunk's tok0 words and result rows, count, scan, PUBLISH the chunk's total, and assemble the chunk's
//              ids (and chunk-local token offsets) in LDS -- none of which needs the tokens in front of the chunk;
//   back(c)    resolve the look-back (by now the predecessors have long published), then LDS -> ids[] / pt_tokoff[] as whole
//              wavefront-wide stores.
// A workgroup runs front(c_next) before back(c): the look-back round trips hide behind a whole chunk of work (measured: the
// un-pipelined kernel spent 45 % of its time in them).  Two LDS buffers alternate.  A chunk whose tokens do not fit the buffer
// (> CP_STAGE: only text made of many-token pre-tokens) is scattered straight from its rows in back().
// Chunks go round robin over the grid (chunk = blockIdx + k * gridDim): neighbouring chunks belong to neighbouring workgroups, which
// run in step, so a look-back rarely finds anything unpublished.  Forward progress does NOT depend on that, nor on which workgroups
// the hardware keeps resident -- the reference's encode_batch is &self + Send + Sync (tokenizer/mod.rs:1328-1348): any number of
// callers, two compactions on two streams each holding half the chip included.  front() waits for nothing; back(c) needs the TOTALS
// of the chunks in front of c, and a total is a pure function of tok0 and the rows, which are final before this kernel starts: a
// wavefront that has polled an unpublished predecessor lb `patience` times (its owner may not have been scheduled yet, and may
// not be until this workgroup gets out of the way) computes that chunk's total itself and publishes it on the owner's behalf
// (results.hip lb_resolve; the owner later stores the same value and does its own copy-out, which nobody waits for).  Every missing
// total is therefore finite work for whoever waits for it: every wait ends, at any grid and any residency.
// (Round 4 first handed the chunks out by a global ticket, which gives the same guarantee by construction: an atomic on one address
// resolves at ~12 ns apiece at the memory side, 19 k tickets a C2 batch, and the kernel went from 0.14 to 0.26 ms -- profiles/r4a_*.)
// =================================================================================================
// Two shapes: 8 pre-tokens per lane (chunks of 2048, 56 KB of LDS: two workgroups per CU) and 4 (chunks of 1024, 28 KB and half the
// registers: five workgroups per CU -- more chunks in flight to hide the load -> rows -> look-back chain of each); the host picks one
// (TKAMD_CP_ITEMS) and k_doc_first_pretok is told the chunk size.
constexpr int CP_NT = 256;
template <int CP_ITEMS> struct CpShape {
    static constexpr int CHUNK = CP_NT * CP_ITEMS;     // pre-tokens per chunk (the host sizes the state array by COMPACT_CHUNK_MIN)
    static constexpr int STAGE = CP_ITEMS * 640;       // tokens of a chunk assembled in LDS (2.5 per pre-token; more: scattered from the rows)
    static_assert(CP_ITEMS == 2 || CP_ITEMS == 4 || CP_ITEMS == 8, "shapes the launcher knows");
};

template <int CP_ITEMS> struct CpRows {
    uint4 row[CP_ITEMS];
    uint32_t cnt[CP_ITEMS];
};
// the two halves of a chunk's loads: a lane's CP_ITEMS consecutive tok0 words, then -- once those are here -- the rows they name
template <int CP_ITEMS> struct CpTok0 { uint32_t w[CP_ITEMS]; };
template <int CP_ITEMS>
__device__ __forceinline__ void cp_load_tok0(const uint32_t* __restrict__ tok0, int64_t p0, int64_t P, CpTok0<CP_ITEMS>& f) {
    static_assert(CP_ITEMS == 2 || CP_ITEMS == 4 || CP_ITEMS == 8, "one 8-byte, one or two 16-byte loads of tok0 per lane");
    uint32_t* const first = f.w;
    // (tok0 is read once: non-temporal, so that the rows the kernel gathers stay in the L2)
    // The wide load is UNCONDITIONAL -- a lane whose words reach beyond P (the batch's last chunk only) loads the array's first words
    // instead and then fetches its words one by one.  Written as `if (whole) wide load else word loads`, the two branches fill the same
    // registers and the compiler waits for the wide load right behind it (a write-after-write on the registers): the load a chunk
    // AHEAD -- the whole point of CpAhead -- was a round trip at the top of every iteration (s_waitcnt vmcnt(0) seven instructions
    // behind the load in the ISA of rounds 3 to 5).
    const bool whole = p0 + CP_ITEMS <= P;
    const int64_t pl = whole ? p0 : 0;
    if (CP_ITEMS == 2) {
        const uint2 a = load_nt((const uint2*)(tok0 + pl));
        first[0] = a.x; first[CP_ITEMS - 1] = a.y;
    } else {
        const uint4 a = load_nt((const uint4*)(tok0 + pl));
        first[0] = a.x; first[1] = a.y; first[CP_ITEMS > 2 ? 2 : 0] = a.z; first[CP_ITEMS > 3 ? 3 : 0] = a.w;
    }
    if (CP_ITEMS == 8) {
        const uint4 b = load_nt((const uint4*)(tok0 + pl + 4));
        first[CP_ITEMS - 4] = b.x; first[CP_ITEMS - 3] = b.y; first[CP_ITEMS - 2] = b.z; first[CP_ITEMS - 1] = b.w;
    }
    if (!whole) {
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) first[k] = (p0 + k < P) ? tok0[p0 + k] : 0u;
    }
}
template <int CP_ITEMS> struct CpAhead { CpTok0<CP_ITEMS> f; uint32_t dlo, dhi; };     // what k_compact loads a chunk ahead
// the gathers of a lane's rows, issued (cp_issue_rows) and turned into rows and counts (cp_settle_rows) -- two steps, so that the DEEP
// shape of the kernel can put a whole chunk of work between them
template <int CP_ITEMS> struct CpGather { uint4 ld[CP_ITEMS]; };
template <int CP_ITEMS>
__device__ __forceinline__ void cp_issue_rows(const CpTok0<CP_ITEMS>& f, const uint4* __restrict__ rows, const uint4* __restrict__ crows, CpGather<CP_ITEMS>& g) {
    const uint32_t* const first = f.w;
    // EVERY word loads a row -- one that names none (seven of eight at C2) loads row 0 and drops it.  A load under `if (names a row)`
    // whose other branch fills the same registers with the inline token makes the compiler wait for the load before that branch may
    // write them (a write-after-write on the registers, whatever the lanes): up to round 5 the CP_ITEMS gathers of a lane went out one
    // after the other, a memory round trip each (s_waitcnt vmcnt(0) behind every one of them in the ISA), although a wavefront all of
    // whose 64 lanes skip a given load is one in thousands.  Unconditional, they are all in flight together (0.140 -> 0.126 ms at C2,
    // profiles/r5l_ab_c2.txt).
    // (Also tried there: one BYTE per claimed slot with the row's token count, read ahead of the row so that the chunk's total is
    // published while the slots' rows are still in flight -- the rows then need registers of their own until the copy-out, the kernel
    // spills at five workgroups per CU, and it measured 0.151 ms.)
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        const uint4* const src = ((first[k] & TOK_SLOT) == TOK_SLOT) ? crows : rows;
        g.ld[k] = src[(first[k] & TOK_ROW) ? (first[k] & TOK_REF_MASK) : 0u];
    }
}
template <int CP_ITEMS>
__device__ __forceinline__ uint32_t cp_settle_rows(const CpTok0<CP_ITEMS>& f, const CpGather<CP_ITEMS>& g, CpRows<CP_ITEMS>& r) {
    const uint32_t* const first = f.w;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        r.row[k] = (first[k] & TOK_ROW) ? g.ld[k] : make_uint4((first[k] & TOK_ID_MASK) | (((first[k] & TOK_ONE) ? 1u : 0u) << ROW_CNT_SHIFT), 0u, 0u, 0u);
        r.cnt[k] = row_count(r.row[k]);
        v += r.cnt[k];
    }
    return v;
}
template <int CP_ITEMS>
__device__ __forceinline__ uint32_t cp_load_rows(const CpTok0<CP_ITEMS>& f, const uint4* __restrict__ rows, const uint4* __restrict__ crows, CpRows<CP_ITEMS>& r) {
    CpGather<CP_ITEMS> g;
    cp_issue_rows<CP_ITEMS>(f, rows, crows, g);
    return cp_settle_rows<CP_ITEMS>(f, g, r);
}
template <int CP_ITEMS>
__device__ __forceinline__ uint32_t cp_load(const uint32_t* __restrict__ tok0, const uint4* __restrict__ rows, const uint4* __restrict__ crows, int64_t p0, int64_t P, CpRows<CP_ITEMS>& r) {
    CpTok0<CP_ITEMS> f;
    cp_load_tok0<CP_ITEMS>(tok0, p0, P, f);
    return cp_load_rows<CP_ITEMS>(f, rows, crows, r);
}
// (words 1..3 of a row go out WITH their top byte -- the boundary in front of the token, results.hip row_boundary -- when DST is the
// LDS stage: the copy-out splits them into ids and tok_b8; the direct scatter of an oversized chunk splits them here, B8 null: ids only)
#define TKAMD_CP_SCATTER(DST, R, O, RAW, B8)                                                                  \
    _Pragma("unroll") for (int k = 0; k < CP_ITEMS; ++k) {                                                    \
        const uint32_t c = (R).cnt[k];                                                                        \
        if (c) {                                                                                              \
            const bool more = ((R).row[k].x >> ROW_CNT_SHIFT) == ROW_CNT_MORE;                                \
            const uint32_t keep_ = (RAW) ? 0xFFFFFFFFu : TOK_ID_MASK;                                         \
            (DST)[(O)] = ((R).row[k].x & ROW_ID_MASK) | ((RAW) && (B8) ? B8_FIRST << ROW_B8_SHIFT : 0u);      \
            if (!(RAW) && (B8)) (B8)[(O)] = (uint8_t)B8_FIRST;                                                \
            if (!more) {                                                                                      \
                if (c > 1) { (DST)[(O) + 1] = (R).row[k].y & keep_; if (!(RAW) && (B8)) (B8)[(O) + 1] = (uint8_t)((R).row[k].y >> ROW_B8_SHIFT); } \
                if (c > 2) { (DST)[(O) + 2] = (R).row[k].z & keep_; if (!(RAW) && (B8)) (B8)[(O) + 2] = (uint8_t)((R).row[k].z >> ROW_B8_SHIFT); } \
                if (c > 3) { (DST)[(O) + 3] = (R).row[k].w & keep_; if (!(RAW) && (B8)) (B8)[(O) + 3] = (uint8_t)((R).row[k].w >> ROW_B8_SHIFT); } \
            } else {                                                                                          \
                if (!(RAW) && (B8)) for (uint32_t j = 1; j < c; ++j) (B8)[(O) + j] = 0;                       \
                /* ids 1.. of a longer run: tmp_ids[s + 1 ..], four to a load (16 bytes at any alignment; the  */ \
                /* buffer extends four words past the text, so the last load may overshoot the run)           */ \
                const uint32_t* const src_ = tmp_ids + (R).row[k].y;                                          \
                for (uint32_t j = 1; j < c; j += 4) {                                                         \
                    const Unaligned16 v_ = *(const Unaligned16*)(src_ + j);                                   \
                    (DST)[(O) + j] = v_.a;                                                                    \
                    if (j + 1 < c) (DST)[(O) + j + 1] = v_.b;                                                 \
                    if (j + 2 < c) (DST)[(O) + j + 2] = v_.c;                                                 \
                    if (j + 3 < c) (DST)[(O) + j + 3] = v_.d;                                                 \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        (O) += c;                                                                                             \
    }

// The token CSR of the documents (Encoding per document, tokenizer/mod.rs:1345-1348) leaves this kernel too: the documents whose first
// pre-token lies in a chunk (chunk_lo / doc_pt of k_doc_first_pretok; an empty document starts at its successor's) get
// tok_offsets[d] = the token offset of that pre-token -- so the ids-only path never writes the P-sized pt_tokoff (null then).
// (PROF: the diagnostic instantiation -- TKAMD_PHASES, tkamd_debug_phases -- stamps the shader clock per phase into phases[workgroup][8]:
// 0 loads + scan + publish of front(), 1 its LDS scatter, 2 the look-back wait, 3 the copy-out, 7 the whole kernel)
// (min. wavefronts per SIMD: 5 -- five workgroups per CU, <= 96 VGPRs -- for the shapes of 2 and 4 pre-tokens per lane, whose 28 KB of LDS
// allow it; the helper of the look-back must not cost the main path its occupancy)
// (Session Q also ran a shape with the rows of a chunk gathered a chunk ahead -- issued at the top of an iteration for the chunk after
// next, settled at the top of the next one, 94 VGPRs: level, 0.1266 against 0.1250 ms -- the front's round trip is not what a chunk waits for.)
// EARLY: the look-back's first read -- the 128 chunks in front -- goes out at the TOP of the iteration and is looked at behind front()
// of the next chunk (results.hip lb_prefetch / lb_resolve_pre).
template <int CP_ITEMS, bool PROF = false, bool EARLY = true>
__global__ __launch_bounds__(CP_NT, CP_ITEMS == 8 ? 2 : 5) void k_compact(const uint32_t* __restrict__ tok0, const uint4* __restrict__ rows, const uint4* __restrict__ crows,
                                                   const uint32_t* __restrict__ tmp_ids, const int64_t* __restrict__ n_pretok,
                                                   unsigned long long* __restrict__ state,
                                                   int64_t* __restrict__ n_tok, uint32_t* __restrict__ pt_tokoff, uint32_t* __restrict__ ids,
                                                   const uint32_t* __restrict__ chunk_lo, const uint32_t* __restrict__ doc_pt, int64_t n_docs,
                                                   int64_t* __restrict__ tok_offsets, unsigned long long* __restrict__ phases, uint32_t patience,
                                                   uint8_t* __restrict__ tok_b8) {
    // tok_b8 (with offsets only, else null): per token, the boundary byte in front of it that its row carried (results.hip row_boundary)
    constexpr int CP_CHUNK = CpShape<CP_ITEMS>::CHUNK, CP_STAGE = CpShape<CP_ITEMS>::STAGE;
    unsigned long long ph_t = 0ull, ph_t0 = 0ull, ph_acc[4] = {0ull, 0ull, 0ull, 0ull};
    auto tick = [&](int k) {
        if (PROF && threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph_acc[k] += now - ph_t; ph_t = now; }
    };
    if (PROF && threadIdx.x == 0) ph_t = ph_t0 = __builtin_amdgcn_s_memtime();
    __shared__ uint32_t sm[4];
    __shared__ uint32_t s_dlo[2], s_dhi[2];              // documents [dlo, dhi) start in the chunk
    __shared__ uint32_t s_docpt[2][CP_NT];               // doc_pt of the first CP_NT of them, loaded with the chunk
    __shared__ uint32_t s_stage[2][CP_STAGE];
    __shared__ uint32_t s_loc[2][CP_CHUNK];              // chunk-local token offset of every pre-token
    __shared__ uint32_t s_tot[2];
    __shared__ unsigned long long s_lbw;
    const int64_t P = uniform_i64(*n_pretok);           // (scalar registers: the chunk count and every chunk's bounds follow it)
    const int64_t n_chunks = (P + CP_CHUNK - 1) / CP_CHUNK;
    const int tid = (int)threadIdx.x;
    if (n_chunks == 0) {                                  // no pre-token at all: every document is empty
        if (blockIdx.x == 0)
            for (int64_t d = tid; d <= n_docs; d += CP_NT) tok_offsets[d] = 0;
        return;
    }
    // front half of a chunk into LDS buffer b.  49 % of the kernel's time was the wait in here (profiles/r4b_ab_c2.txt): TWO chains of
    // two dependent round trips each -- tok0, then the rows it names; chunk_lo, then the doc_pt entries it names.  The heads of both
    // are loaded a chunk AHEAD (`a`, behind the previous chunk's work), so front() starts with the second halves, and the doc_pt
    // values are only parked in LDS behind the scan: the chunk's total is published as soon as the ROWS are in.
    // (The rows and the doc_pt value a chunk ahead as well, tok0 two ahead -- 91 VGPRs, still five workgroups per CU -- measured level:
    // 0.1408 against 0.1414 ms, profiles/r4h_ab_c2.txt.  The wait is not these loads' latency alone.)
    auto front = [&](int64_t ch, int b, const CpAhead<CP_ITEMS>& a) {
        // (the last chunk also takes the documents that start behind the last pre-token: trailing empty ones and the closing entry)
        const uint32_t dlo = a.dlo, dhi = a.dhi;
        CpRows<CP_ITEMS> r;
        const bool has_doc = dlo + (uint32_t)tid < dhi;
        uint32_t my_docpt = 0u;
        const uint32_t v = cp_load_rows<CP_ITEMS>(a.f, rows, crows, r);
        if (has_doc) my_docpt = doc_pt[dlo + (uint32_t)tid];
        uint32_t tot;
        const uint32_t ex = block256_excl_scan(v, sm, &tot);
        if (tid == 0) { lb_publish(state, ch, (unsigned long long)tot); s_tot[b] = tot; }
        if (has_doc) s_docpt[b][tid] = my_docpt;
        if (tid == 0) { s_dlo[b] = dlo; s_dhi[b] = dhi; }
        tick(0);
        uint32_t acc = ex;
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) { s_loc[b][tid * CP_ITEMS + k] = acc; acc += r.cnt[k]; }
        if (tot <= (uint32_t)CP_STAGE) {
            uint32_t o = ex;
            uint32_t* const dst = s_stage[b];
            TKAMD_CP_SCATTER(dst, r, o, true, tok_b8)
        }
        tick(1);
    };
    // the total of a chunk some other workgroup owns, by ONE wavefront (the look-back's way out of waiting: see the top of this section)
    auto chunk_total = [&](int64_t hc) -> unsigned long long {
        uint32_t v = 0;
#pragma unroll 1
        for (int j = 0; j < CP_NT / 64; ++j) {
            CpRows<CP_ITEMS> r;
            v += cp_load<CP_ITEMS>(tok0, rows, crows, hc * CP_CHUNK + (int64_t)(j * 64 + (tid & 63)) * CP_ITEMS, P, r);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
        return (unsigned long long)v;
    };
    auto ahead_of = [&](int64_t c, CpAhead<CP_ITEMS>& a) {     // (a chunk beyond the end: zeros, never used)
        a.dlo = a.dhi = 0u;
        if (c < n_chunks) {
            cp_load_tok0<CP_ITEMS>(tok0, c * CP_CHUNK + (int64_t)tid * CP_ITEMS, P, a.f);
            a.dlo = chunk_lo[c];
            a.dhi = c == n_chunks - 1 ? (uint32_t)n_docs + 1u : chunk_lo[c + 1];
        } else {
#pragma unroll
            for (int k = 0; k < CP_ITEMS; ++k) a.f.w[k] = 0u;
        }
    };
    int b = 0;
    CpAhead<CP_ITEMS> aa;                                  // the NEXT chunk's tok0 words and document range
    {
        CpAhead<CP_ITEMS> a0;
        ahead_of(blockIdx.x, a0);
        ahead_of((int64_t)blockIdx.x + gridDim.x, aa);
        if ((int64_t)blockIdx.x < n_chunks) front(blockIdx.x, 0, a0);
    }
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x, b ^= 1) {
        const int64_t nxt = ch + gridDim.x;
        // EARLY: wavefront 0 asks for the states of the 128 chunks in front of `ch` NOW and looks at them behind front(nxt).  Every
        // workgroup's look-back finds its predecessors' prefixes just in time -- the workgroups fell into step at the kernel's start, 64
        // of them a round trip behind the 64 before -- so the read that starts when back() starts always costs its whole round trip
        // (31..36 % of the kernel, three wavefronts at a barrier meanwhile); the prefixes some 64 to 128 chunks back are older:
        // a read two windows wide often finds one of them, every total in between has long been published, and the answer is there when
        // front(nxt) is done (compact 0.1245 -> 0.115 ms at C2, profiles/r5r_*, r5t_*).  What it does not settle, lb_resolve_pre goes on polling for like lb_resolve.
        // (behind ahead_of: the compiler drains every memory operation of the last iteration in there -- the registers of `an` are the
        // loop's own -- and a read issued in front of that would be waited for at once; from here it returns with front(nxt)'s gathers)
        LbPre pre;
        CpAhead<CP_ITEMS> an;                              // ... and those of the one after it: in flight while front(nxt) works
        ahead_of(nxt + gridDim.x, an);
        if (EARLY && tid < 64) lb_prefetch(state, ch, pre);
        if (nxt < n_chunks) front(nxt, b ^ 1, aa);        // (its two barriers also order this chunk's LDS writes before the reads below)
        else __syncthreads();
        aa = an;
        const uint32_t tot = s_tot[b];
        // the chunk's place in the token stream: wavefront 0 looks back (results.hip), the others wait at the barrier
        if (tid < 64) {
            const unsigned long long r = EARLY ? lb_resolve_pre(state, ch, (unsigned long long)tot, patience, chunk_total, pre)
                                               : lb_resolve(state, ch, (unsigned long long)tot, patience, chunk_total);
            if (tid == 0) s_lbw = r;
        }
        __syncthreads();
        const unsigned long long base = s_lbw;
        tick(2);
        if (ch == n_chunks - 1 && tid == 0) *n_tok = (int64_t)(base + tot);
        const int64_t pc = ch * CP_CHUNK;
        if (pt_tokoff) {
#pragma unroll
            for (int i = 0; i < CP_ITEMS; ++i) {
                const int q = i * CP_NT + tid;
                if (pc + q < P) pt_tokoff[pc + q] = (uint32_t)base + s_loc[b][q];
            }
        }
        {
            const uint32_t dlo = s_dlo[b], dhi = s_dhi[b];
            for (uint32_t d = dlo + (uint32_t)tid; d < dhi; d += CP_NT) {
                // (the LDS copy read by every lane, the array in memory only by the lanes beyond it -- through a volatile access: `in LDS ?
                // s_docpt[..] : doc_pt[..]`, and two plain loads in two branches, which the compiler folds back into it, is ONE load
                // through a flat pointer, which waits for everything in flight)
                uint32_t dp = s_docpt[b][(d - dlo) & (uint32_t)(CP_NT - 1)];
                if (d - dlo >= (uint32_t)CP_NT) dp = *(const volatile uint32_t*)(doc_pt + d);
                const uint32_t local = dp - (uint32_t)pc;
                tok_offsets[d] = (int64_t)(base + (local < (uint32_t)CP_CHUNK ? s_loc[b][local] : tot));
            }
        }
        if (tot <= (uint32_t)CP_STAGE) {
            if (tok_b8) {                                  // (wavefront-uniform) the staged words carry the boundary bytes: split them
                for (uint32_t i = (uint32_t)tid; i < tot; i += CP_NT) {
                    const uint32_t w = s_stage[b][i];
                    store_nt(ids + base + i, w & TOK_ID_MASK);
                    tok_b8[base + i] = (uint8_t)(w >> ROW_B8_SHIFT);
                }
            } else {
                for (uint32_t i = (uint32_t)tid; i < tot; i += CP_NT) store_nt(ids + base + i, s_stage[b][i]);
            }
        } else {                                           // rare: too many tokens for the buffer -- scatter from the rows
            const int64_t p0 = ch * CP_CHUNK + (int64_t)tid * CP_ITEMS;
            CpRows<CP_ITEMS> r;
            cp_load<CP_ITEMS>(tok0, rows, crows, p0, P, r);
            uint32_t o = s_loc[b][tid * CP_ITEMS];
            uint32_t* const dst = ids + base;
            uint8_t* const b8 = tok_b8 ? tok_b8 + base : nullptr;
            TKAMD_CP_SCATTER(dst, r, o, false, b8)
        }
        __syncthreads();                                   // buffer b is free for front() of the chunk after next
        tick(3);
    }
    if (PROF && tid == 0 && phases) {
        unsigned long long* const o = phases + (size_t)blockIdx.x * 8;
        for (int k = 0; k < 4; ++k) o[k] += ph_acc[k];
        o[7] += ph_t - ph_t0;
    }
}
#undef TKAMD_CP_SCATTER

// is_pretokenized inputs (InputSequence::PreTokenized, tokenizer/mod.rs:782-795): every word of a sequence went through the pipeline
// as a document of its own -- the reference encodes each word separately and merges the encodings -- and sequence s owns the tokens
// of the words [seq_off[s], seq_off[s + 1]).  (seq_off is the validated copy: launch_validate_csr.)
__global__ void k_seq_tok_offsets(const int64_t* __restrict__ seq_off, int64_t n_seqs, const int64_t* __restrict__ word_tok_off,
                                  int64_t* __restrict__ seq_tok_off) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= n_seqs) seq_tok_off[s] = word_tok_off[seq_off[s]];
}
// Encoding.word_ids of a pre-tokenized sequence: the index of the word in its sequence (do_tokenize's word_idx, mod.rs:1178-1200)
// first_tok (process_offsets needs "token 0 of the encoding", byte_level.rs:213-216): the index of the first token of the word's sequence
__global__ void k_word_index(const int64_t* __restrict__ seq_off, int64_t n_seqs, int64_t n_words, uint32_t* __restrict__ widx,
                             const int64_t* __restrict__ seq_tok_off, int64_t* __restrict__ first_tok) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    int64_t lo = 0, hi = n_seqs;                          // last s with seq_off[s] <= w
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (seq_off[mid] <= w) lo = mid; else hi = mid; }
    if (widx) widx[w] = (uint32_t)(w - seq_off[lo]);
    if (first_tok) first_tok[w] = seq_tok_off[lo];
}

// =================================================================================================
// K_token_meta: per-token (start, end) offsets and word ids.
// Replaces the per-token half of PreTokenizedString::into_encoding (tokenizer/pre_tokenizer.rs:231-256):
//   offsets = split.offsets_original().0 + convert_offsets(Normalized(token.offsets))   (:237-241)
//   word    = index of the split inside the document                                     (:252-256)
// plus BytesToCharOffsetConverter for OffsetType::Char (:329-364) and the ByteLevel post-processor's
// process_offsets (pre_tokenizers/byte_level.rs:202-234) when trim_offsets is set.
// Byte-level rule (byte_level.rs:135-143, tests/offsets.rs:47-57): a token that covers only part of a
// multi-byte char reports the whole char, so starts snap back and ends snap forward to char boundaries.
// One lane per pre-token; the document of a pre-token is found by binary search over doc_pt.
// =================================================================================================
__device__ __forceinline__ uint32_t lead_rank(const unsigned long long* __restrict__ leadmask, const uint32_t* __restrict__ lprefix, uint32_t pos) {
    unsigned long long m = leadmask[pos >> 6];
    uint32_t b = pos & 63u;
    return lprefix[pos >> 6] + (uint32_t)__popcll(m & ((1ull << b) - 1ull));
}

// lead-byte bitmask of the text (char offsets: BytesToCharOffsetConverter, pre_tokenizer.rs:329-364, as ranks of lead bytes).  A lane takes
// sixteen bytes -- one load -- and folds "not a continuation byte" of each into four bits per word with a multiply (bits 0 / 8 / 16 / 24 of
// a word land on bits 24..27, no carries); four lanes make one mask word.  (Rounds 1-5: a lane per byte and a ballot, 0.215 ms on 120 MB.)
__global__ __launch_bounds__(256) void k_leadmask(const uint8_t* __restrict__ text, int64_t n_bytes, unsigned long long* __restrict__ leadmask) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;          // 16-byte group
    const int64_t i = q * 16;
    uint32_t bits = 0u;
    if (i < n_bytes) {                                                   // (the text is readable TEXT_PAD bytes beyond its end)
        const uint4 v = *(const uint4*)(text + i);
        auto four = [](uint32_t w) -> uint32_t {
            const uint32_t cont = (w >> 7) & (~w >> 6) & 0x01010101u;    // byte is 10xxxxxx
            return (((cont ^ 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
        };
        bits = four(v.x) | (four(v.y) << 4) | (four(v.z) << 8) | (four(v.w) << 12);
        const int64_t left = n_bytes - i;
        if (left < 16) bits &= (1u << left) - 1u;
    }
    unsigned long long m = (unsigned long long)bits << (16 * (threadIdx.x & 3));
    m |= __shfl_xor(m, 1, 64);
    m |= __shfl_xor(m, 2, 64);
    if ((threadIdx.x & 3) == 0 && i <= n_bytes) leadmask[i >> 6] = m;
}

// end of the original byte range of x byte k.  Behind BertNormalizer (norig_e null) every byte of a normalised char carries the range of
// its SOURCE char -- (i, i + that char's UTF-8 length), a verbatim byte (i, i + 1) -- so the end follows from the start and one byte
// of the original text (a continuation byte, as the last byte of a verbatim char is, counts 1); the prefix-space copy keeps its
// per-byte ends in memory.
__device__ __forceinline__ uint32_t norig_end(const MetaArgs& a, uint32_t k) {
    if (a.norig_e) return a.norig_e[k];
    const uint32_t i = a.norig[k], b = a.text[i];
    return i + (b < 0xC0u ? 1u : b < 0xE0u ? 2u : b < 0xF0u ? 3u : 4u);
}

// offsets / word id of ONE token: token j of pre-token p (document d; [s, e) in x space; its tokens are o .. o + c), covering
// [s + rel, s + rel_end) of the x text.  Everything of into_encoding / process_offsets that is per token.
// snapped: the caller knows the edges already snapped to char boundaries (the boundary bytes the rows carried, results.hip): bs / be.
// (the outputs are written once and read by nobody here: non-temporal stores -- token_meta 0.4005 -> 0.391 ms, profiles/r6i_*)
__device__ __forceinline__ void meta_one_token(const MetaArgs& a, int64_t p, int64_t d, uint32_t s, uint32_t e, bool is_match, uint32_t o, uint32_t j,
                                               uint32_t rel, uint32_t rel_end, uint32_t xdoc, uint32_t odoc, uint32_t word,
                                               bool snapped = false, uint32_t snap_bs = 0u, uint32_t snap_be = 0u) {
    if (a.want_words) store_nt(a.word_ids + o + j, word);
    if (!a.want_offsets) return;
    uint32_t ts = s + rel, te = s + rel_end;              // token bytes in x space
    uint32_t bs = ts, be = te;
    if (snapped) { bs = snap_bs; be = snap_be; }
    else if (a.snap_chars && !is_match) {                 // snap to char boundaries inside the pre-token (its own edges are char boundaries)
        while (bs > s && (a.x_text[bs] & 0xC0u) == 0x80u) --bs;
        while (be < e && (a.x_text[be] & 0xC0u) == 0x80u) ++be;
    }
    // x space -> original text
    uint32_t os, oe;
    const uint8_t* ttext = a.x_text;                      // text the trimming below reads, and the token's span in it
    uint32_t tts = ts, tte = te;
    if (is_match) {
        const uint32_t ml = a.tmp_end[s];
        os = a.norig ? a.norig[s] : s - xdoc + odoc;
        if (ml & MATCH_LEN_ORIG) { oe = os + (ml & ~MATCH_LEN_ORIG); ttext = a.text; tts = os; tte = oe; }
        else { tte = s + ml; oe = a.norig ? norig_end(a, tte - 1u) : tte - xdoc + odoc; }
    }
    else if (a.norig) { os = a.norig[bs]; oe = norig_end(a, be - 1u); }
    else { os = bs - xdoc + odoc; oe = be - xdoc + odoc; }
    if (a.char_mode) {
        uint32_t base = lead_rank(a.leadmask, a.lprefix, odoc);
        os = lead_rank(a.leadmask, a.lprefix, os) - base;
        oe = lead_rank(a.leadmask, a.lprefix, oe) - base;
    } else { os -= odoc; oe -= odoc; }
    if (a.trim_offsets) {                                 // process_offsets, byte_level.rs:202-234
        uint32_t lead_sp = 0, trail_sp = 0;
        if (is_match) {
            // an added token's text is the raw slice: its leading / trailing chars are tested with char::is_whitespace
            uint32_t q = tts;
            while (q < tte) { uint32_t l; const uint32_t cp = utf8_global(ttext, q, &l); if (cp != 0x120u && !(uc_flags(cp, a.uc1, a.uc2) & UC_RUST_WS)) break; ++lead_sp; q += l; }
            q = tte;
            while (q > tts) {
                uint32_t r = q - 1;
                while (r > tts && (ttext[r] & 0xC0u) == 0x80u) --r;
                uint32_t l;
                const uint32_t cp = utf8_global(ttext, r, &l);
                if (cp != 0x120u && !(uc_flags(cp, a.uc1, a.uc2) & UC_RUST_WS)) break;   // (*c == 'Ġ' || c.is_whitespace())
                ++trail_sp;
                q = r;
            }
        } else if (!a.trim_matches_only) {
            while (ts + lead_sp < te && a.x_text[ts + lead_sp] == 0x20u) ++lead_sp;
            while (trail_sp < te - ts && a.x_text[te - 1 - trail_sp] == 0x20u) ++trail_sp;
        }
        bool took_one = false;
        const uint32_t os0 = os, oe0 = oe;
        if (lead_sp) {
            // (token 0 of the encoding, or offsets that start at 0 -- every word of a pre-tokenized sequence, byte_level.rs:213-216;
            // `word` is no test for it: all pre-tokens of a sequence's word 0 carry word id 0; first_tok: pre-tokenized input)
            bool is_first = (a.first_tok ? (int64_t)(o + j) == a.first_tok[d] : (p == (int64_t)a.doc_pt[d] && j == 0)) || os == 0;
            if (is_first && a.pp_add_prefix_space && lead_sp == 1) lead_sp = 0;
            took_one = lead_sp == 1 && a.pp_add_prefix_space && os < oe;
            os = min(os + lead_sp, oe);
        }
        if (trail_sp && oe >= trail_sp) oe = max(oe - trail_sp, os);
        // (2: as token 0 of an encoding its END would differ too -- the token is nothing but that one space, so the trailing
        // trim stops at a start that lies one further left)
        if (a.trim1) a.trim1[o + j] = !took_one ? 0 : (((trail_sp && oe0 >= trail_sp) ? max(oe0 - trail_sp, os0) : oe0) != oe) ? 2 : 1;
    }
    store_nt((uint2*)(a.offsets + 2 * (size_t)(o + j)), make_uint2(os, oe));
}

// One lane per PRE-TOKEN, its tokens in a loop: the shape of rounds 1-5, kept for the one configuration whose token edges depend on the
// tokens in front of them -- BPE over characters without an unk_token, where dropped chars move every later edge (`running` below).
__global__ __launch_bounds__(256) void k_token_meta_seq(MetaArgs a) {
    const int64_t P = *a.n_pretok;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
        const uint32_t o = a.pt_tokoff[p];
        const uint32_t c = ((p + 1 < P) ? a.pt_tokoff[p + 1] : (uint32_t)*a.n_tok) - o;
        if (!c) continue;
        const uint32_t s = a.pt_start[p], e = a.pt_end ? a.pt_end[p] : a.pt_start[p + 1];
        // document of this pre-token: last d with doc_pt[d] <= p
        int64_t lo = 0, hi = a.n_docs;
        while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if ((int64_t)a.doc_pt[mid] <= p) lo = mid; else hi = mid; }
        const int64_t d = lo;
        const uint32_t word = a.word_of_doc ? a.word_of_doc[d] : (uint32_t)(p - a.doc_pt[d]);
        const uint32_t xdoc = (uint32_t)a.x_doc_off[d];
        const uint32_t odoc = (uint32_t)a.doc_off[d];
        uint32_t rel = 0;
        // an added-token match is one token whose own length is on record (k_scatter_matches): a later, overlapping match may have
        // cut it short in the start mask, and its text is the raw slice (trimmed by real whitespace chars)
        const bool is_match = a.matchmask && a.tmp_end && ((a.matchmask[s >> 6] >> (s & 63)) & 1ull);
        uint32_t s_ends = s;                                          // whose token ends: the pre-token's own, or the claimant's of its word
        if (a.claim_pos && c > 1) {
            const uint32_t t0 = a.tok0[p];
            if ((t0 & TOK_SLOT) == TOK_SLOT) s_ends = a.claim_pos[t0 & TOK_REF_MASK];
        }
        // BPE over characters without an unk_token: a char the vocabulary lacks leaves no symbol, and the reference's token offsets are
        // running sums of the symbols' lengths (word.rs:260-268) -- every edge behind a dropped char moves up by its bytes.  The model
        // kernels report edges as positions; the dropped bytes in front of each edge are counted here, walking the word once.
        // (not on a whole-word hit of ignore_merges: vocab.get(sequence) reports (0, sequence.len()), bpe/model.rs:559-567)
        bool whole_word = false;
        if (a.ww_tok0 && c == 1) {
            const uint32_t t0 = a.ww_tok0[p];
            if ((t0 & TOK_SLOT) == TOK_ONE) whole_word = true;
            else if (t0 & TOK_ROW) whole_word = (((t0 & TOK_ONE) ? (const uint4*)a.ww_crows : (const uint4*)a.ww_rows)[t0 & TOK_REF_MASK]).y == ROW_WHOLE_WORD;
        }
        uint32_t dq = s, dropped = 0u;
        auto running = [&](uint32_t rel_pos) -> uint32_t {
            if (!a.char_id || whole_word) return rel_pos;
            const uint32_t target = s + rel_pos;
            while (dq < target) {
                uint32_t l;
                const uint32_t cp = utf8_global(a.x_text, dq, &l);
                l = min(l, e - dq);
                const uint32_t var = ((dq != s && (a.cb & CB_PREFIX)) ? 1u : 0u) | ((dq + l >= e && (a.cb & CB_SUFFIX)) ? 2u : 0u);
                if (cp >= 0x110000u || a.char_id[(cp << 2) | var] == CHAR_NONE) dropped += l;
                dq += l;
            }
            return rel_pos - dropped;
        };
        for (uint32_t j = 0; j < c; ++j) {
            uint32_t rel_end = running((c == 1 || !a.tmp_end) ? (e - s) : a.tmp_end[s_ends + j]);      // (no token ends without offsets: word ids only)
            meta_one_token(a, p, d, s, e, is_match, o, j, rel, rel_end, xdoc, odoc, word);
            rel = rel_end;
        }
    }
}

// Round 6's shape.  A workgroup takes tiles of TM_TILE consecutive pre-tokens.  Into LDS, with coalesced loads: their token offsets, starts
// and ends; the entries of the documents around them (first pre-token, start in both texts: a table of TM_DOCS documents from the one
// the compaction's chunk_lo names -- no search); the tile's tokens' boundary bytes (tok_b8, dense per token next to the ids:
// results.hip row_boundary; the byte of a pre-token's FIRST token is a marker the compaction puts there).  The documents that start
// inside the tile are counted into its pre-tokens and a scan turns the counts into every pre-token's document.  Then A LANE PER TOKEN,
// in output order: the tile's tokens are one dense range [T0, T1) of the output arrays, so every store of the kernel is a coalesced one,
// and the loop reads nothing but LDS.  Which pre-token a token belongs to is a RANK: the markers make a bit mask over the tile's tokens
// (a compare of sixteen bytes a lane, four lanes a mask word -- LDS atomics on one mask word from the 50 lanes whose pre-tokens share it
// serialised: 0.09 ms of a 0.43 ms kernel, profiles/r6n_*), a scan over the mask's words gives the set bits in front of each, and token
// t's pre-token is prefix[t / 64] + popcount(mask word below bit t).  (A tile with more tokens than the mask holds, or with a pre-token
// of no token at all -- BPE over characters dropping every char of a word -- takes a binary search over the tile's token offsets and
// reads its bytes from memory.)  Rows that carry no boundary bytes (pre-tokens of > 32 bytes or > 4 tokens) take the old way: tmp_end at
// the pre-token's byte position, or behind tok0 -> claim_pos.
// (Rounds 1-5: a lane per pre-token with its tokens in a loop and a twenty-step binary search over doc_pt each: 1.14 ms on C2's
// 22.7 M tokens.  A lane per token with a binary search over the tile's token offsets and every token through tmp_end: 0.52 ms.  Four
// pre-tokens a lane with their tokens in a loop: 0.73 ms -- a wavefront ran the loop as often as its longest word has tokens.  Single-token
// pre-tokens written by their lane, the others listed and dealt a token a lane: 0.40 ms, 0.39 with the boundary bytes dense, 0.29 with a
// grid of what is resident (profiles/r6[a-k]_*): four dependent round trips a tile, and stores that left holes for the second phase to
// fill -- 400 MB written for 272.)
constexpr int TM_TILE = 1024;
constexpr int TM_TOKCAP = 4096;                           // tokens of a tile the rank mask and the LDS copy of their bytes hold (four per pre-token)
constexpr int TM_DOCS = 256;                              // documents of a tile whose entries sit in LDS
constexpr int TM_EXC = 512;                               // tokens of a tile on the list of those that need memory
// (software-pipelined: a tile's loads -- everything above -- are issued a tile AHEAD into registers, behind the parking of the current
// tile's into LDS, and the three scalars they depend on -- the tile's first and last token, its first document -- two tiles ahead: the
// two dependent round trips a tile cost, 0.09 ms of the kernel's 0.29 with nothing else to do, fly during the previous tile's work.
// Loads are unconditional with clamped addresses: a load under a condition whose other branch fills the same registers is waited for
// at once -- kernels/output.hip cp_load_tok0.)
struct TmScal { uint32_t T0, T1, lw, s_end; int64_t d0; };      // lw: the 64-byte word of the text that holds the tile's first pre-token's start (CHARS / MASKS)
template <bool HAS_END, bool CHARS, bool MASKS, bool NORIG> struct TmAhead {
    uint32_t tokoff[4], start[MASKS ? 1 : 4], end[HAS_END && !MASKS ? 4 : 1], dpt, dxo, dod;
    Unaligned16 b8;
    unsigned long long lm[CHARS ? 1 : 0]; uint32_t lp[CHARS ? 1 : 0];
    unsigned long long sm[MASKS ? 1 : 0]; uint32_t wp[MASKS ? 1 : 0];      // MASKS: word lw + lane of the start mask, the starts in front of it
    unsigned long long em[MASKS && HAS_END ? 1 : 0];                       // ... of the end mask ("Removed" pre-tokenizers)
    unsigned long long mm[NORIG ? 1 : 0];                                  // ... of the added-token matches (NORIG)
};
constexpr int TM_LEADW = 256;                             // char mode: 64-byte words of the text, from the tile's first pre-token on, whose lead-byte mask and prefix sit in LDS
// SIMPLE: what most tokenizers are -- no normalizer's alignment map, no added-token matches, no trim_offsets, documents that are not the
// words of pre-tokenized sequences: a token's offsets are its (snapped) edges minus its document's start, and the general path's flag
// tests (a thousand scalar instructions in meta_one_token, taken or not) are not compiled in.
// CHARS (with SIMPLE): char offsets, ranks from the LDS window of the lead-byte mask (an instantiation of its own: the byte-offset kernel
// measured 6 % slower carrying the window's registers and LDS, profiles/r7d_*)
// MASKS (pre-tokenizers without an end mask): the pre-tokens' starts are read off the START MASK -- the 256 words from the one that holds
// the tile's first pre-token on (tile_w, a by-product of the mask scan), every lane walking its word's bits into the tile's LDS array by
// rank; more windows if the tile's text is longer (pre-tokens of hundreds of bytes).  pt_start -- 4 bytes a pre-token written by
// k_emit_pretok and read back here, 0.06 ms of launch on C2 -- does not exist then.
// NORIG (with SIMPLE and MASKS, round 6's last step): behind BertNormalizer -- the x text is the normalised text, a token's offsets
// are its edges through the alignment map (norig; the end of a byte's range from its start and one byte of the original text) -- with
// the general path's flag tests compiled out like SIMPLE's.  Added-token matches are a property of the TILE there: the match mask
// rides in the mask window, a tile that holds a match bit takes the general path whole.
template <bool HAS_END, bool SIMPLE, bool CHARS, bool MASKS, bool NORIG = false>
// (five wavefronts a SIMD where the kernel fits 96 registers without a spill -- the instantiations without end mask and char window --,
// four elsewhere: a spilled value's reload in the token loop is a wait for the last iteration's stores)
__global__ __launch_bounds__(256, MASKS ? 5 : 4) void k_token_meta(MetaArgs a) {
    static_assert(SIMPLE || !CHARS, "the LDS window is the SIMPLE path's");
    static_assert(!NORIG || (SIMPLE && MASKS && !CHARS), "the alignment-map path reads the match mask with the mask window; char ranks from memory");
    __shared__ uint2 s_ts[TM_TILE + 1];                   // pre-token i: first token, first byte (one 16-byte read gives i and i + 1)
    __shared__ uint32_t s_end[HAS_END ? TM_TILE : 1];
    __shared__ uint32_t s_doc[TM_TILE];                   // documents starting AT pre-token i, then (scanned) the document of pre-token i
    __shared__ unsigned long long s_tmask[TM_TOKCAP / 64];    // bit t: token T0 + t is the first token of a pre-token
    __shared__ uint32_t s_tpre[TM_TOKCAP / 64];           // set bits in front of word w
    __shared__ __attribute__((aligned(16))) uint8_t s_b8[TM_TOKCAP + 16];
    __shared__ uint4 s_dtab[TM_DOCS];                     // document dbase + k: first pre-token, start in the x text, in the original, (char mode) lead bytes in front of it
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_before, s_slow, s_nexc, s_cov, s_match;
    __shared__ uint16_t s_exc[TM_EXC];                    // tokens of the tile that need memory (see the token loop)
    // char mode (SIMPLE: the x text IS the original text): lead-byte mask and lead bytes in front of the TM_LEADW words from the tile's
    // first pre-token on -- a tile of prose spans about a hundred; a token that ends beyond them takes the general path
    __shared__ unsigned long long s_lm[CHARS ? TM_LEADW : 1];
    __shared__ uint32_t s_lp[CHARS ? TM_LEADW : 1];
    static_assert(TM_LEADW == 256, "a word a lane");
    static_assert(TM_TOKCAP == 256 * 16, "sixteen boundary bytes a lane");
    // (scalar registers: what is loaded through a pointer lives in vector registers, and so does everything derived from it -- the tile
    // count, every tile's base, the addresses of the per-tile scalars, which then are vector loads too)
    const int64_t P = uniform_i64(*a.n_pretok);
    const uint32_t n_tok = (uint32_t)uniform_i64(*a.n_tok);
    const int tid = (int)threadIdx.x;
    const int64_t n_tiles = (P + TM_TILE - 1) / TM_TILE;
    if (n_tiles == 0) return;
    const int64_t G = gridDim.x;
    constexpr bool chars = CHARS;
    const uint32_t lw_max = chars ? (uint32_t)uniform_i64(a.doc_off[a.n_docs] >> 6) : 0u;      // the last word of the lead-byte mask
    const uint32_t x_len = MASKS ? (uint32_t)uniform_i64(a.x_len_dev ? *a.x_len_dev : a.x_len_host) : 0u;
    const uint32_t n_mw = MASKS ? (uint32_t)min(a.n_mask_words, (int64_t)(x_len >> 6) + 1) : 0u;     // words of the start mask that hold bits
    auto scal_of = [&](int64_t tile, TmScal& sc) {        // (a tile beyond the end: the last one's, never used)
        const int64_t base = min(tile, n_tiles - 1) * TM_TILE;
        const int64_t pe = min(base + TM_TILE, P);
        sc.T0 = a.pt_tokoff[base];
        sc.lw = MASKS ? a.tile_w[base / TM_TILE] : CHARS ? a.pt_start[base] >> 6 : 0u;
        sc.T1 = pe < P ? a.pt_tokoff[pe] : n_tok;
        sc.s_end = MASKS ? 0u : a.pt_start[pe];           // (pt_start[P] is the sentinel k_emit_pretok writes: the text's length)
        sc.d0 = (int64_t)a.chunk_lo[base / a.chunk];      // chunk_lo[c]: the first d with doc_pt[d] >= c * chunk
    };
    auto ahead_of = [&](int64_t tile, const TmScal& sc, TmAhead<HAS_END, CHARS, MASKS, NORIG>& h) {
        const int64_t base = min(tile, n_tiles - 1) * TM_TILE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t p = min(base + tid + 256 * q, P - 1);
            h.tokoff[q] = a.pt_tokoff[p];
            if constexpr (!MASKS) h.start[q] = a.pt_start[p];
            if constexpr (HAS_END && !MASKS) h.end[q] = a.pt_end[p];
        }
        const int64_t d = min((sc.d0 > 0 ? sc.d0 - 1 : 0) + tid, a.n_docs - 1);
        h.dpt = a.doc_pt[d];
        if constexpr (!SIMPLE || NORIG) h.dxo = ((const uint32_t*)a.x_doc_off)[2 * d];    // (the low words: a batch is < 4 GiB; SIMPLE without a map: x text == text, one CSR)
        else h.dxo = 0u;
        h.dod = ((const uint32_t*)a.doc_off)[2 * d];
        if (a.tok_b8) h.b8 = *(const Unaligned16*)(a.tok_b8 + min(sc.T0 + 16u * (uint32_t)tid, n_tok));     // (readable 64 bytes beyond the tokens)
        else h.b8 = Unaligned16{0u, 0u, 0u, 0u};
        if constexpr (CHARS) { const uint32_t w = min(sc.lw + (uint32_t)tid, lw_max); h.lm[0] = a.leadmask[w]; h.lp[0] = a.lprefix[w]; }
        if constexpr (MASKS) { const uint32_t w = min(sc.lw + (uint32_t)tid, n_mw - 1u); h.sm[0] = a.startmask[w]; h.wp[0] = a.wprefix[w]; if constexpr (HAS_END) h.em[0] = a.endmask[w]; if constexpr (NORIG) h.mm[0] = a.matchmask ? a.matchmask[w] : 0ull; }
    };
    TmScal sc0, sc1;
    TmAhead<HAS_END, CHARS, MASKS, NORIG> h;
    scal_of(blockIdx.x, sc0);
    scal_of((int64_t)blockIdx.x + G, sc1);
    ahead_of(blockIdx.x, sc0, h);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += G) {
        const int64_t base = tile * TM_TILE;
        const int np = (int)min((int64_t)TM_TILE, P - base);
        TmScal sc2;
        scal_of(tile + 2 * G, sc2);
        const int64_t d0 = sc0.d0;
        const int64_t dbase = d0 > 0 ? d0 - 1 : 0;
        const uint32_t T0 = sc0.T0, nt = sc0.T1 - T0;
        bool slow = nt > (uint32_t)TM_TOKCAP || !a.tok_b8;
        if (NORIG && tid == 0) s_match = 0u;              // (its last reader was the previous tile's second barrier)
        __syncthreads();                                  // (the previous tile's readers are done)
        // ---- this tile's loads, asked for a tile ago: into LDS
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q;
            if constexpr (MASKS) s_ts[i].x = h.tokoff[q];     // (the starts: the walk over the mask words below)
            else s_ts[i] = make_uint2(h.tokoff[q], h.start[q]);    // (beyond np: the clamped loads' values, read by nobody)
            if constexpr (HAS_END && !MASKS) s_end[i] = h.end[q];
            s_doc[i] = 0u;
        }
        // documents from the one in front of the first of the compaction chunk that holds `base` on (the tile's first pre-tokens may belong
        // to that one): their entries go to the table
        s_dtab[tid] = make_uint4(h.dpt, (!SIMPLE || NORIG) ? h.dxo : h.dod, h.dod, 0u);
        if constexpr (CHARS) { s_lm[tid] = h.lm[0]; s_lp[tid] = h.lp[0]; }
        // (char mode: the lead bytes in front of every listed document -- asked for in FRONT of the next tile's loads: loads return in order)
        uint32_t dlead_now = 0u;
        if (CHARS && dbase + tid < a.n_docs) dlead_now = lead_rank(a.leadmask, a.lprefix, h.dod);
        if (tid == 0) { s_before = 0u; s_slow = 0u; s_nexc = 0u; }
        // MASKS: the starts of the pre-tokens base .. base + np (the last one: the end of pre-token np - 1), by rank, from a window of 256
        // mask words; `wp` counts the starts in front of the word -- ranks in front of the tile wrap around and fail the test
        auto walk_starts = [&](unsigned long long m, uint32_t wp, uint32_t word) {
            uint32_t r = wp - (uint32_t)base;
            for (; m; m &= m - 1ull, ++r)
                if (r <= (uint32_t)np) s_ts[r].y = (word << 6) + (uint32_t)(__ffsll((unsigned long long)m) - 1);
        };
        // (an end bit closes the last start in front of it: the end of pre-token [starts in front of the word + starts below the bit] - 1.
        // Every pre-token of such a pre-tokenizer has one, in front of the next start: the windows that hold the starts hold the ends)
        auto walk_ends = [&](unsigned long long sm, unsigned long long em, uint32_t wp, uint32_t word) {
            for (; em; em &= em - 1ull) {
                const uint32_t b = (uint32_t)(__ffsll((unsigned long long)em) - 1);
                const uint32_t r = wp + (uint32_t)__popcll(sm & ((1ull << b) - 1ull)) - 1u - (uint32_t)base;
                if (r < (uint32_t)np) s_end[r] = (word << 6) + b;
            }
        };
        if constexpr (MASKS) {
            const uint32_t word = sc0.lw + (uint32_t)tid;
            walk_starts(word < n_mw ? h.sm[0] : 0ull, h.wp[0], word);
            if constexpr (HAS_END) walk_ends(word < n_mw ? h.sm[0] : 0ull, word < n_mw ? h.em[0] : 0ull, h.wp[0], word);
            if (tid == 255) s_cov = h.wp[0] + (uint32_t)__popcll(h.sm[0]) - (uint32_t)base;      // ranks below this are placed (a clamped word: every rank there is)
            if constexpr (NORIG) { if (h.mm[0]) s_match = 1u; }
        }
        if (!slow) {
            // the tile's boundary bytes, sixteen a lane, and the mask of the FIRST markers among them (four lanes a word)
            const Unaligned16 v = h.b8;
            *(uint4*)(s_b8 + 16 * tid) = make_uint4(v.a, v.b, v.c, v.d);
            auto four = [](uint32_t w) -> uint32_t {      // bit k: byte k of w is B8_FIRST
                const uint32_t x = w ^ (B8_FIRST * 0x01010101u);
                const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
                return (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
            };
            uint32_t bits = four(v.a) | (four(v.b) << 4) | (four(v.c) << 8) | (four(v.d) << 12);
            const uint32_t left = nt - min(nt, 16u * (uint32_t)tid);
            if (left < 16u) bits &= (1u << left) - 1u;
            unsigned long long m = (unsigned long long)bits << (16 * (tid & 3));
            m |= __shfl_xor(m, 1, 64);
            m |= __shfl_xor(m, 2, 64);
            if ((tid & 3) == 0) s_tmask[tid >> 2] = m;
        }
        ahead_of(tile + G, sc1, h);                       // the next tile's loads: in flight from here to the top of the next iteration
        __syncthreads();
        if constexpr (MASKS) {
            // (rare: the tile's text is longer than a window -- more windows, loaded here and waited for)
            uint32_t cov = s_cov, wnext = sc0.lw + 256u;
            while (cov <= (uint32_t)np && wnext < n_mw) {
                __syncthreads();                          // (everyone has read s_cov)
                const uint32_t word = wnext + (uint32_t)tid, wc = min(word, n_mw - 1u);
                const unsigned long long m = a.startmask[wc];
                const uint32_t wp = a.wprefix[wc];
                walk_starts(word < n_mw ? m : 0ull, wp, word);
                if constexpr (HAS_END) walk_ends(word < n_mw ? m : 0ull, word < n_mw ? a.endmask[wc] : 0ull, wp, word);
                if constexpr (NORIG) { if (a.matchmask && a.matchmask[wc]) s_match = 1u; }
                if (tid == 255) s_cov = wp + (uint32_t)__popcll(m) - (uint32_t)base;
                __syncthreads();
                cov = s_cov;
                wnext += 256u;
            }
            if (tid == 0) { s_ts[np].x = sc0.T1; if (base + np == P) s_ts[np].y = x_len; }     // (the sentinel: the text's length)
        } else if (tid == 0) s_ts[np] = make_uint2(sc0.T1, sc0.s_end);        // (behind the barrier: lane np & 255 wrote a clamped value there)
        if (CHARS) s_dtab[tid].w = dlead_now;
        // the documents in front of the tile are counted, the ones inside it add to their first pre-token (an empty document to the next one's)
        for (int64_t d = d0 + tid; d < a.n_docs; d += 256) {
            const int64_t k = d - dbase;
            const int64_t r = (int64_t)(k < TM_DOCS ? s_dtab[k].x : a.doc_pt[d]) - base;
            if (r >= np) break;
            atomicAdd(r < 0 ? &s_before : &s_doc[r], 1u);
        }
        if (!slow) {
            // (a pre-token of no token: ranks do not count it.  Lane np's pair uses the sentinel: by value, tid 0's store may not be there yet)
            for (int i = tid; i < np; i += 256)
                if ((i + 1 == np ? sc0.T1 : s_ts[i + 1].x) == s_ts[i].x) s_slow = 1u;
        }
        __syncthreads();
        {   // inclusive scan over the tile, four pre-tokens a lane: the document of pre-token i = the last d with doc_pt[d] <= base + i
            uint32_t v[4], sum = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = (4 * tid + q < np) ? s_doc[4 * tid + q] : 0u; sum += v[q]; }
            uint32_t tot;
            uint32_t run = block256_excl_scan(sum, s_scan, &tot) + (uint32_t)d0 + s_before - 1u;
#pragma unroll
            for (int q = 0; q < 4; ++q) { run += v[q]; if (4 * tid + q < np) s_doc[4 * tid + q] = run; }
        }
        slow = slow || s_slow != 0u || (NORIG && s_match != 0u);
        if (!slow && tid < 64) {                          // set bits in front of every mask word: sixty-four words, one wavefront, no barrier of its own
            static_assert(TM_TOKCAP / 64 == 64, "a word a lane of wavefront 0");
            const uint32_t v = (uint32_t)__popcll(s_tmask[tid]);
            s_tpre[tid] = wave_incl_scan(v) - v;
        }
        __syncthreads();
        // ---- a lane per token, in output order.  The loop of the tokens whose every need is in LDS holds NO load from memory, on any
        // branch: a load that fills the registers an LDS read fills elsewhere makes the compiler wait for everything in flight in front of
        // that read (s_waitcnt vmcnt(0): the next tile's loads, and the last iteration's STORES -- a round trip an iteration, 0.26 ms for
        // this kernel against 0.2x).  The tokens that need memory -- a row without boundary bytes, a document beyond the table -- go on
        // a list and through `general` behind the loop; so does every token of a tile without a rank mask.
        auto general = [&](uint32_t t, int i) {
            const uint32_t tt = T0 + t;
            uint32_t v0 = 0u, v1 = 0u;                    // the boundary bytes in front of the token and behind it
            if (a.tok_b8) { v0 = a.tok_b8[tt]; v1 = a.tok_b8[tt + 1u]; }
            const uint2 ts0 = s_ts[i], ts1 = s_ts[i + 1];
            const uint32_t o = ts0.x, c = ts1.x - o, j = tt - o;
            const uint32_t s = ts0.y, e = HAS_END ? s_end[i] : ts1.y;
            const int64_t p = base + i, d = (int64_t)s_doc[i];
            uint32_t word = a.word_of_doc ? a.word_of_doc[d] : (uint32_t)(p - (int64_t)a.doc_pt[d]);
            const uint32_t xdoc = (uint32_t)a.x_doc_off[d], odoc = (uint32_t)a.doc_off[d];
            uint32_t rel = 0u, rel_end = e - s;           // (no token ends without offsets: word ids only)
            bool snapped = c == 1u;                       // (one token: the pre-token's own edges, char boundaries both)
            uint32_t bs = s, be = e;
            if (c > 1u && a.tok_b8) {
                // its token ends: carried by its row -- the byte in front of token 1 says so -- or, the old way, in tmp_end: its own, or
                // behind tok0 -> claim_pos its claimant's
                const uint32_t carried = j == 1u ? v0 : j == 0u ? v1 : (uint32_t)a.tok_b8[o + 1u];
                if (carried) {
                    snapped = true;
                    if (j) { rel = v0 & 31u; bs = s + rel - b8_back(v0); }
                    if (j + 1u < c) { rel_end = v1 & 31u; be = s + rel_end + b8_fwd(v1); }
                } else if (a.tmp_end) {
                    uint32_t se = s;
                    if (a.claim_pos) {
                        const uint32_t t0 = a.tok0[p];
                        if ((t0 & TOK_SLOT) == TOK_SLOT) se = a.claim_pos[t0 & TOK_REF_MASK];
                    }
                    if (j) rel = a.tmp_end[se + j - 1u];
                    rel_end = a.tmp_end[se + j];
                }
            }
            const bool is_match = a.matchmask && a.tmp_end && ((a.matchmask[s >> 6] >> (s & 63)) & 1ull);
            meta_one_token(a, p, d, s, e, is_match, o, j, rel, rel_end, xdoc, odoc, word, snapped && !is_match, bs, be);
        };
        if (slow) {                                       // (workgroup-uniform)
            for (uint32_t t = (uint32_t)tid; t < nt; t += 256u) {
                int lo = 0, hi = np;                      // the last i with tokoff[i] <= T0 + t (its successor's lies beyond: it has tokens)
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_ts[mid].x <= T0 + t) lo = mid; else hi = mid; }
                general(t, lo);
            }
        } else {
            const uint32_t dbase32 = (uint32_t)dbase, lw0 = sc0.lw;
            for (uint32_t t = (uint32_t)tid; t < nt; t += 256u) {
                // (branch-free up to the stores: every branch of a wavefront is a dozen scalar instructions whether taken or not)
                const uint32_t tt = T0 + t, w = t >> 6;
                const int i = (int)(s_tpre[w] + (uint32_t)__popcll(s_tmask[w] & ((2ull << (t & 63u)) - 1ull))) - 1;
                const uint32_t b0 = s_b8[t], b1 = s_b8[t + 1u];
                const uint2 ts0 = s_ts[i], ts1 = s_ts[i + 1];
                const uint32_t o = ts0.x, c = ts1.x - o, j = tt - o;
                const uint32_t s = ts0.y, e = HAS_END ? s_end[i] : ts1.y;
                const uint32_t dk = s_doc[i] - dbase32;
                const uint32_t carried = (uint32_t)s_b8[o + 1u - T0] | (c == 1u ? 1u : 0u);      // (the byte in front of token 1: 0 = a row without boundary bytes)
                // the boundary in front of the token (none in front of token 0) and behind it (none behind the last one: the pre-token's end)
                const uint32_t v0 = j ? b0 : 0u, v1 = j + 1u < c ? b1 : 0u;
                const uint32_t rel = v0 & 31u, rel_end = v1 ? (v1 & 31u) : e - s;
                const uint32_t bs = s + rel - b8_back(v0), be = s + rel_end + b8_fwd(v1);
                if (dk >= (uint32_t)TM_DOCS || !carried || (chars && (be >> 6) - lw0 >= (uint32_t)TM_LEADW)) {
                    const uint32_t k = atomicAdd(&s_nexc, 1u);
                    if (k < (uint32_t)TM_EXC) s_exc[k] = (uint16_t)t;
                    continue;
                }
                const uint4 de = s_dtab[dk];
                const uint32_t word = (uint32_t)(base + i) - de.x, xdoc = de.y, odoc = de.z;
                if (SIMPLE) {
                    if (a.want_words) store_nt(a.word_ids + tt, word);
                    if (a.want_offsets) {
                        uint32_t os = bs - xdoc, oe = be - xdoc;
                        if constexpr (NORIG) {            // x space -> original text through the alignment map (meta_one_token's way, no match in this tile)
                            const uint32_t i1 = a.norig[be - 1u];
                            os = a.norig[bs];
                            const uint32_t b = a.text[i1];
                            oe = i1 + (b < 0xC0u ? 1u : b < 0xE0u ? 2u : b < 0xF0u ? 3u : 4u);
                            if (a.char_mode) {
                                const uint32_t lead0 = lead_rank(a.leadmask, a.lprefix, odoc);
                                os = lead_rank(a.leadmask, a.lprefix, os) - lead0;
                                oe = lead_rank(a.leadmask, a.lprefix, oe) - lead0;
                            } else { os -= odoc; oe -= odoc; }
                        }
                        if (chars) {                      // (x text == original text here: launch_token_meta)
                            const uint32_t ws = (bs >> 6) - lw0, we = (be >> 6) - lw0;
                            os = s_lp[ws] + (uint32_t)__popcll(s_lm[ws] & ((1ull << (bs & 63u)) - 1ull)) - de.w;
                            oe = s_lp[we] + (uint32_t)__popcll(s_lm[we] & ((1ull << (be & 63u)) - 1ull)) - de.w;
                        }
                        store_nt((uint2*)(a.offsets + 2 * (size_t)tt), make_uint2(os, oe));
                    }
                } else {
                    const int64_t p = base + i, d = (int64_t)dk + dbase;
                    const bool is_match = a.matchmask && a.tmp_end && ((a.matchmask[s >> 6] >> (s & 63)) & 1ull);
                    meta_one_token(a, p, d, s, e, is_match, o, j, rel, rel_end, xdoc, odoc, a.word_of_doc ? a.word_of_doc[d] : word, !is_match, bs, be);
                }
            }
            __syncthreads();
            const uint32_t nexc = s_nexc;
            if (nexc > (uint32_t)TM_EXC) {                // (more than the list holds: every token asked again)
                for (uint32_t t = (uint32_t)tid; t < nt; t += 256u) {
                    const uint32_t w = t >> 6;
                    const int i = (int)(s_tpre[w] + (uint32_t)__popcll(s_tmask[w] & ((2ull << (t & 63u)) - 1ull))) - 1;
                    const uint32_t o = s_ts[i].x, c = s_ts[i + 1].x - o;
                    bool exc = (int64_t)s_doc[i] - dbase >= TM_DOCS || (c > 1u && !s_b8[o + 1u - T0]);
                    if (!exc && chars) {                  // (the token's end, as the loop above computed it)
                        const uint32_t j = T0 + t - o, b1 = s_b8[t + 1u], v1 = j + 1u < c ? b1 : 0u;
                        const uint32_t s = s_ts[i].y, e = HAS_END ? s_end[i] : s_ts[i + 1].y;
                        const uint32_t be = s + (v1 ? (v1 & 31u) : e - s) + b8_fwd(v1);
                        exc = (be >> 6) - lw0 >= (uint32_t)TM_LEADW;
                    }
                    if (exc) general(t, i);
                }
            } else {
                for (uint32_t k = (uint32_t)tid; k < nexc; k += 256u) {
                    const uint32_t t = s_exc[k], w = t >> 6;
                    general(t, (int)(s_tpre[w] + (uint32_t)__popcll(s_tmask[w] & ((2ull << (t & 63u)) - 1ull))) - 1);
                }
            }
        }
        sc0 = sc1; sc1 = sc2;
    }
}
