// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  How the model kernels hand their tokens to the compaction.

// =================================================================================================
// Result layout between Model::tokenize and into_encoding (replaces the Vec<Token> every Split carries,
// tokenizer/pre_tokenizer.rs:22-47):
//   tok0[p]   one word per pre-token:  TOK_ONE | id            exactly one token (settled by the lookup kernel)
//                                      TOK_ROW | row            the tokens are in rows[row]        (30 bits of row)
//                                      TOK_SLOT | slot          the tokens are in the row of a claimed / cached slot (both flag bits; kernels/lookup.hip)
//   rows[r]   16 bytes per QUEUED pre-token, named by the queue position the lookup kernel gave it (so the model kernels
//             never touch the P-sized arrays):   { id0 | count << 28, id1, id2, id3 }                 count <= 4
//                                                { id0 | 15 << 28, s, count, 0 }                      any count: ids 1.. in
//             tmp_ids[s + j] -- s is the pre-token's first byte; a pre-token of L bytes has at most L tokens, so the slots
//             s+1 .. s+L-1 of a 4-byte-per-text-byte array belong to it alone.
//   tmp_end[s + j]  (only when offsets are requested) end of token j, in bytes from the pre-token start.
// Token ids are < 2^24 (checked at load), so the flag / count bits never collide with an id.
// =================================================================================================
constexpr uint32_t TOK_ROW = 0x80000000u;
constexpr uint32_t TOK_ONE = 0x40000000u;
constexpr uint32_t TOK_ID_MASK = 0x00FFFFFFu;
constexpr uint32_t TOK_SLOT = TOK_ROW | TOK_ONE, TOK_REF_MASK = 0x3FFFFFFFu;
constexpr uint32_t ROW_CNT_SHIFT = 28, ROW_CNT_MORE = 15u, ROW_ID_MASK = 0x0FFFFFFFu;
constexpr uint32_t ROW_WHOLE_WORD = 0xFFFFFFFFu;      // word 1 of a ONE-token row (unused otherwise): a whole-word vocabulary hit of k_long_vocab, not a merge result (k_token_meta)

// one queued pre-token: first byte and length (the model kernels need nothing else).  Bit 31 of the length: this entry HOLDS the in-batch
// claim of its word (kernels/lookup.hip) -- the model kernel that finishes it also copies its row to the claimed slot's row, where
// the compaction finds it for the word's other occurrences.  Every reader takes the length through qitem_len().
struct __attribute__((aligned(8))) QItem { uint32_t s, len; };
constexpr uint32_t QLEN_CLAIM = 0x80000000u;
__device__ __forceinline__ uint32_t qitem_len(uint32_t len_word) { return len_word & ~QLEN_CLAIM; }

// A work queue is NSQ sub-queues of sq_cap entries each.  Atomics on ONE address serialise at ~10 ns each on MI355X (device-scope
// atomics are resolved at the memory side), and a shared fill counter would take one per workgroup, tile and queue -- tens of
// thousands per batch.  Instead lookup workgroup b owns sub-queue b outright and writes its fill once, at the end; the
// consumers see the sub-queues laid end to end (qview_*).  Position p = sub-queue * sq_cap + index names the queue entry and,
// with row_base, the result row.
// (struct QView / NSQ / QCNT_STRIDE: kernels.hpp)
// all threads of the workgroup: loads the fill counts, leaves their prefix sums in s_pre[NSQ + 1], returns the total
__device__ __forceinline__ uint32_t qview_prefix(const QView& v, uint32_t* s_pre) {
    constexpr int PER = NSQ / 64;                           // one wavefront scans the counts, PER consecutive ones per lane
    static_assert(NSQ % 64 == 0, "sub-queues per lane");
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { c[k] = min(v.counts[(threadIdx.x * PER + k) * QCNT_STRIDE], v.sq_cap); sum += c[k]; }
        uint32_t run = wave_incl_scan(sum) - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) { run += c[k]; s_pre[threadIdx.x * PER + k + 1] = run; }
        if (threadIdx.x == 0) s_pre[0] = 0u;
    }
    __syncthreads();
    return s_pre[NSQ];
}
// the queue position of the item-th entry of the concatenated sub-queues (item < total)
__device__ __forceinline__ uint32_t qview_pos(const uint32_t* s_pre, uint32_t sq_cap, uint32_t item) {
    uint32_t lo = 0, hi = NSQ;                              // invariant: s_pre[lo] <= item < s_pre[hi]
#pragma unroll
    for (int it = 0; it < 10; ++it) {
        static_assert(NSQ <= 1024, "binary search depth");
        const uint32_t mid = (lo + hi) >> 1;
        if (hi - lo > 1) { if (s_pre[mid] <= item) lo = mid; else hi = mid; }
    }
    return lo * sq_cap + (item - s_pre[lo]);
}

__device__ __forceinline__ uint32_t row_count(const uint4& row) {
    const uint32_t cf = row.x >> ROW_CNT_SHIFT;
    return cf < ROW_CNT_MORE ? cf : row.z;
}
// up to four tokens held in registers; beyond four the caller has written ids 1.. to tmp_ids[s + j] itself
// (round 6) Token ENDS in the row.  With offsets requested, a row of two to four tokens of a pre-token of <= 32 bytes carries the boundary
// in front of token j (1..3) in the top byte of its word j (ids are < 2^24): position in the pre-token (1..31, five bits) and -- byte-level
// tokens may cut a multi-byte char, whose whole range both neighbours then report (byte_level.rs:135-143) -- how far the token that
// starts there snaps back and the one that ends there snaps forward (three bits: 0 = a char boundary; 1..6 = (back, fwd) of a cut at
// byte k of a char of L bytes: (1,1) (1,2) (2,1) (1,3) (2,2) (3,1)).  The compaction hands the byte on, one per token, next to the ids
// (tok_b8: a dense array), and k_token_meta reads its neighbours' instead of a sparse array indexed by byte position, a claimant's
// place behind tok0 -> claim_pos and the text at every cut (kernels/output.hip).  0 = not carried: token_meta takes the old way.
// The byte of token 0 is B8_FIRST, put there by the compaction: the set of them is token_meta's map from tokens to pre-tokens.
constexpr uint32_t ROW_B8_SHIFT = 24;
constexpr uint32_t B8_FIRST = 0xE0u;                  // the byte of a pre-token's FIRST token in tok_b8 (no boundary looks like it: position 0, code 7) -- the compaction's marker
__device__ __forceinline__ uint32_t row_boundary(const uint8_t* __restrict__ text, uint32_t s, uint32_t pos, bool snap) {
    uint32_t code = 0u;
    if (snap && (text[s + pos] & 0xC0u) == 0x80u) {       // (the pre-token starts and ends on char boundaries: the walks stay inside it)
        uint32_t back = 1u, fwd = 1u;
        while (back < 3u && (text[s + pos - back] & 0xC0u) == 0x80u) ++back;
        while (fwd < 3u && (text[s + pos + fwd] & 0xC0u) == 0x80u) ++fwd;
        code = back == 1u ? (fwd == 1u ? 1u : fwd == 2u ? 2u : 4u) : back == 2u ? (fwd == 1u ? 3u : 5u) : 6u;
    }
    return (pos | (code << 5)) << ROW_B8_SHIFT;
}
// (two bits a code, looked up in a constant: code 0 -> 0; (back, fwd) of codes 1..6 as listed above)
__device__ __forceinline__ uint32_t b8_back(uint32_t v) { return (0x3994u >> ((v >> 5) * 2u)) & 3u; }
__device__ __forceinline__ uint32_t b8_fwd(uint32_t v) { return (0x1B64u >> ((v >> 5) * 2u)) & 3u; }

__device__ __forceinline__ uint4 make_row(uint32_t count, uint32_t s, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3) {
    return count <= 4u ? make_uint4(id0 | (count << ROW_CNT_SHIFT), id1, id2, id3) : make_uint4(id0 | (ROW_CNT_MORE << ROW_CNT_SHIFT), s, count, 0u);
}

// =================================================================================================
// Single-pass prefix sums (decoupled look-back): chunk c publishes its own total as soon as it knows it, then adds up the
// published totals of its predecessors until it meets one that already carries a full prefix.  One 64-bit word per chunk
// holds the state and the value together, so a reader never sees one without the other; all accesses are device-scope
// atomics (the L2 of one XCD is not coherent with the others').  A wait never depends on another workgroup being scheduled: a
// chunk's total is a pure function of data that is final before the kernel starts, so a wavefront that has polled an unpublished
// predecessor `patience` times computes that total itself (`help`) and publishes it on the owner's behalf -- compare-and-swap from
// "nothing", so an owner that got there first wins, and either way the word holds the same value.
// =================================================================================================
constexpr unsigned long long LB_AGG = 1ull << 62, LB_PREFIX = 2ull << 62, LB_FLAGS = 3ull << 62, LB_VALUE = ~LB_FLAGS;

__device__ __forceinline__ void lb_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lb_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Two halves, so that a workgroup can publish a chunk's total early and come back for its prefix after other work:
//   lb_publish(state, ch, total)   one lane: total of chunk ch is known
//   lb_resolve(state, ch, total, patience, help)
//                                  ONE whole wavefront: returns, in every lane, the sum of all chunks before `ch`, and publishes the
//                                  inclusive prefix of `ch`.  help(c): called by the whole wavefront, returns chunk c's total in every lane.
__device__ __forceinline__ void lb_publish(unsigned long long* __restrict__ state, int64_t ch, unsigned long long total) {
    lb_store(&state[ch], (ch == 0 ? LB_PREFIX : LB_AGG) | total);
}
constexpr uint32_t LB_PATIENCE = 1024;              // polls of an unpublished predecessor before its total is computed here (about half a millisecond; TKAMD_LB_PATIENCE)
// (lb_resolve_from: the walk from chunk i downwards with `run` already summed -- lb_resolve starts it at ch - 1 with nothing)
template <class Help>
__device__ __forceinline__ unsigned long long lb_resolve_from(unsigned long long* __restrict__ state, int64_t ch, int64_t i, unsigned long long run, unsigned long long total,
                                                              uint32_t patience, Help&& help) {
    const int lane = lane_id();
    uint32_t polls = 0u;
    while (true) {
        const int64_t idx = i - lane;
        unsigned long long st;
        uint64_t empty, pref;
        do {                                                        // wait until the window holds no unpublished chunk in front of the first full prefix
            st = idx >= 0 ? lb_load(&state[idx]) : LB_PREFIX;
            empty = __ballot((st & LB_FLAGS) == 0ull);
            pref = __ballot((st & LB_FLAGS) == LB_PREFIX);
            if (pref) {
                const int first = __ffsll((unsigned long long)pref) - 1;          // nearest predecessor with a full prefix
                empty &= (first >= 63) ? ~0ull : ((2ull << first) - 1ull);
            }
            if (empty && ++polls > patience) {                      // (wavefront-uniform) out of patience: one of the missing totals is computed here
                // (different workgroups pick different ones; having helped once, this wait goes on helping without the long pause)
                int k = (int)((blockIdx.x + polls) % (uint32_t)__popcll((unsigned long long)empty));
                uint64_t e = empty;
                for (; k > 0; --k) e &= e - 1ull;
                const int64_t hc = i - (int64_t)(__ffsll((unsigned long long)e) - 1);
                const unsigned long long t = help(hc);
                if (lane == 0) atomicCAS(&state[hc], 0ull, (hc == 0 ? LB_PREFIX : LB_AGG) | t);
                patience = 8u;
                polls = 0u;
            }
        } while (empty);
        const int first = pref ? __ffsll((unsigned long long)pref) - 1 : 63;
        unsigned long long v = (lane <= first) ? (st & LB_VALUE) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        run += v;
        if (pref) break;
        i -= 64;
    }
    if (lane == 0) lb_store(&state[ch], LB_PREFIX | (run + total));
    return run;
}
template <class Help>
__device__ __forceinline__ unsigned long long lb_resolve(unsigned long long* __restrict__ state, int64_t ch, unsigned long long total, uint32_t patience, Help&& help) {
    if (ch == 0) return 0ull;
    return lb_resolve_from(state, ch, ch - 1, 0ull, total, patience, help);
}

// The same look-back with its first read taken EARLY: lb_prefetch asks for the states of the LB_PRE_W * 64 = 128 chunks in front of `ch` (one
// load per window and lane, all in flight together, no wait), lb_resolve_pre looks at them -- window by window, nearest first -- and
// returns at the first full prefix if every chunk in front of it had published; anything else (an unpublished chunk, no prefix in
// reach) goes on from there through lb_resolve's own loop, fresh reads and helping included.  A state word only ever goes from
// nothing to a total to a prefix, so a value read early is still true when it is used.
// (How wide: sessions S / T, profiles/r5s_*, r5t_*: two, three and four windows level -- compact 0.1152 .. 0.1178 ms against 0.1245 without
// the early read --, eight windows slower than none, 0.1375: eight device-scope loads a chunk are traffic of their own.)
constexpr int LB_PRE_W = 2;
struct LbPre { unsigned long long st[LB_PRE_W]; };
__device__ __forceinline__ void lb_prefetch(const unsigned long long* __restrict__ state, int64_t ch, LbPre& p) {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < LB_PRE_W; ++r) {
        const int64_t idx = ch - 1 - 64 * r - lane;
        p.st[r] = idx >= 0 ? lb_load(&state[idx]) : LB_PREFIX;
    }
}
template <class Help>
__device__ __forceinline__ unsigned long long lb_resolve_pre(unsigned long long* __restrict__ state, int64_t ch, unsigned long long total, uint32_t patience, Help&& help,
                                                             const LbPre& p) {
    const int lane = lane_id();
    if (ch == 0) return 0ull;
    // every window's share of this lane first, ONE sum over the lanes behind them (the branches are wavefront-uniform)
    unsigned long long acc = 0ull;
    bool done = false;
    int used = 0;                                                   // windows taken whole (or up to their first full prefix)
#pragma unroll
    for (int r = 0; r < LB_PRE_W; ++r) {
        if (!done && used == r) {
            const unsigned long long st = p.st[r];
            uint64_t empty = __ballot((st & LB_FLAGS) == 0ull);
            const uint64_t pref = __ballot((st & LB_FLAGS) == LB_PREFIX);
            const int first = pref ? __ffsll((unsigned long long)pref) - 1 : 63;
            if (pref) empty &= (first >= 63) ? ~0ull : ((2ull << first) - 1ull);
            if (!empty) {                                           // (an unpublished chunk in front of the prefix: fresh reads from this window on)
                acc += (lane <= first) ? (st & LB_VALUE) : 0ull;
                used = r + 1;
                done = pref != 0ull;
            }
        }
    }
    unsigned long long run = acc;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) run += __shfl_xor(run, d, 64);
    if (done) {
        if (lane == 0) lb_store(&state[ch], LB_PREFIX | (run + total));
        return run;
    }
    return lb_resolve_from(state, ch, ch - 1 - 64 * (int64_t)used, run, total, patience, help);
}

// (Round 4 also tried the look-back by the whole 256-lane workgroup, four windows of 64 predecessors a round: two barriers a round
// cost more than the shorter walk saved -- k_compact 0.152 ms against 0.143 ms on C2, profiles/r4g_ab_c2.txt.)
