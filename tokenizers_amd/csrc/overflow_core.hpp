// Encoding::truncate (tokenizer/encoding.rs:307-395) in closed form: which token ranges of a sequence of n tokens become the
// truncated encoding (part 0) and its `overflowing` encodings (parts 1..), for a window of max_len tokens that advances by
// max_len - stride.  One host+device function pair, so that the CPU tests run exactly what the kernels run
// (tests/test_abi.py replays it against the reference wheel's Encoding.truncate).
#pragma once
#include <cstdint>

#include "tables.hpp"

namespace tkamd {

// Number of encodings a sequence of n tokens leaves: 1 (nothing cut, encoding.rs:309-311), 2 when max_len == 0 (the empty
// encoding + the whole sequence as its one overflowing piece, :313-317), else the window positions of :327-356 --
//   Right: starts 0, off, 2 off, ... until start + max_len >= n;   Left: stops n, n - off, ... until stop <= max_len
// both = ceil((n - max_len) / off) + 1 with off = max_len - stride.  stride >= max_len is the reference's assert (:319): the
// caller reports it; 0 is returned.
TK_HD uint32_t ovf_parts(uint64_t n, uint32_t max_len, uint32_t stride) {
    if ((uint64_t)max_len >= n) return 1u;
    if (max_len == 0u) return 2u;
    if (stride >= max_len) return 0u;
    const uint64_t off = (uint64_t)(max_len - stride);
    const uint64_t p = (n - max_len + off - 1) / off + 1;
    return p > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)p;
}

// Token range [start, start + count) of part p (p < ovf_parts(...)) inside the sequence.
TK_HD void ovf_part_range(uint64_t n, uint32_t max_len, uint32_t stride, bool left, uint32_t p, uint64_t* start, uint64_t* count) {
    if ((uint64_t)max_len >= n) { *start = 0; *count = n; return; }
    if (max_len == 0u) { *start = 0; *count = p ? n : 0; return; }
    const uint64_t off = (uint64_t)(max_len - stride);
    if (!left) {
        const uint64_t s = (uint64_t)p * off;
        const uint64_t e = s + max_len < n ? s + max_len : n;
        *start = s;
        *count = e - s;
    } else {
        const uint64_t e = n - (uint64_t)p * off;                 // stop of the p-th window counted from the end
        const uint64_t s = e > max_len ? e - max_len : 0;          // stop.saturating_sub(max_len)
        *start = s;
        *count = e - s;
    }
}

}  // namespace tkamd
