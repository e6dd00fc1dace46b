/*
 * _marshal: list[str] -> one UTF-8 buffer + int64 CSR offsets, without a Python-level loop.
 *
 * Host-side counterpart of the extraction loop in PyTokenizer::encode_batch
 * (bindings/python/src/tokenizer.rs:1320-1327: every item is extracted into an owned Rust String while
 * the GIL is held).  Here each str is asked for its cached UTF-8 representation
 * (PyUnicode_AsUTF8AndSize; zero-copy for ASCII strs) and memcpy'd into the batch buffer.
 * Errors mirror the reference: a non-str item raises TypeError("TextInputSequence must be str")
 * (tokenizer.rs:274); tuples / lists (pair or pre-tokenized inputs) are reported with a distinct
 * message so the caller can raise UnsupportedError.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#define TEXT_PAD 64

/* The list's str objects are scattered over the heap, so marshalling is bound by cache misses on the object headers and on the
 * character data, not by the copy.  Both passes therefore run on several threads: pass A reads every item's header (type, kind,
 * length) and sizes its UTF-8; the main thread turns the per-thread totals into base offsets and provides the destination; pass C
 * writes the CSR offsets and the bytes.  The GIL is held by the calling thread throughout -- the helper threads touch no Python
 * state, they only READ the (immutable) headers and character data of objects the sequence keeps alive, and nothing can mutate the
 * sequence or its strs while this thread holds the GIL.  A str that is not ASCII is encoded straight from its UCS1 / UCS2 / UCS4
 * code units into the batch buffer (PyUnicode_AsUTF8AndSize would first materialise a second, cached copy inside every such str);
 * the rare items that need the interpreter (a non-str: the exception; a str with lone surrogates: the UnicodeEncodeError; a
 * legacy non-compact str) are left to the main thread, in index order, so the first bad item raises what the sequential loop would. */
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <unistd.h>

typedef struct { const void* p; int64_t len; } span_t;     /* len >= 0: UTF-8 bytes at p; len | SPAN_ENCODE: p is the str, encoded from its code units */
#define SPAN_ENCODE ((int64_t)1 << 62)
#define SPAN_LEN(sp) ((sp)->len & ~SPAN_ENCODE)

#define PACK_PREFETCH 12
#define PACK_MAX_THREADS 32
#define PACK_DEFAULT_THREADS 24   /* measured on the MI355X box's 256 cores (profiles/r5f_list_leg.txt): 24 threads pack fastest, 32 lose to them */
#define SPAN_SLOW (-1)         /* needs the interpreter (main thread) */

static inline int64_t utf8_size_ucs1(const Py_UCS1* s, int64_t n) { int64_t b = n; for (int64_t i = 0; i < n; ++i) b += s[i] >> 7; return b; }
static inline int64_t utf8_size_ucs2(const Py_UCS2* s, int64_t n) {
    int64_t b = 0;
    for (int64_t i = 0; i < n; ++i) { const unsigned c = s[i]; if (c - 0xD800u < 0x800u) return -1; b += c < 0x80u ? 1 : c < 0x800u ? 2 : 3; }
    return b;
}
static inline int64_t utf8_size_ucs4(const Py_UCS4* s, int64_t n) {
    int64_t b = 0;
    for (int64_t i = 0; i < n; ++i) { const unsigned c = s[i]; if (c - 0xD800u < 0x800u || c > 0x10FFFFu) return -1; b += c < 0x80u ? 1 : c < 0x800u ? 2 : c < 0x10000u ? 3 : 4; }
    return b;
}
static inline char* utf8_put(char* d, unsigned c) {
    if (c < 0x80u) { *d++ = (char)c; }
    else if (c < 0x800u) { *d++ = (char)(0xC0u | (c >> 6)); *d++ = (char)(0x80u | (c & 0x3Fu)); }
    else if (c < 0x10000u) { *d++ = (char)(0xE0u | (c >> 12)); *d++ = (char)(0x80u | ((c >> 6) & 0x3Fu)); *d++ = (char)(0x80u | (c & 0x3Fu)); }
    else { *d++ = (char)(0xF0u | (c >> 18)); *d++ = (char)(0x80u | ((c >> 12) & 0x3Fu)); *d++ = (char)(0x80u | ((c >> 6) & 0x3Fu)); *d++ = (char)(0x80u | (c & 0x3Fu)); }
    return d;
}
static void span_write(const span_t* sp, char* d) {
    if (!(sp->len & SPAN_ENCODE)) { memcpy(d, sp->p, (size_t)sp->len); return; }
    PyObject* it = (PyObject*)sp->p;                              /* a compact, non-ASCII str without a cached UTF-8 form */
    const int64_t n = (int64_t)PyUnicode_GET_LENGTH(it);
    const void* data = (const void*)(((const PyCompactUnicodeObject*)it) + 1);
    const int kind = (int)PyUnicode_KIND(it);
    if (kind == PyUnicode_1BYTE_KIND) { const Py_UCS1* s = (const Py_UCS1*)data; for (int64_t i = 0; i < n; ++i) d = utf8_put(d, s[i]); }
    else if (kind == PyUnicode_2BYTE_KIND) { const Py_UCS2* s = (const Py_UCS2*)data; for (int64_t i = 0; i < n; ++i) d = utf8_put(d, s[i]); }
    else { const Py_UCS4* s = (const Py_UCS4*)data; for (int64_t i = 0; i < n; ++i) d = utf8_put(d, s[i]); }
}

/* pass A for one item, without touching interpreter state: the item's UTF-8 source and size, or SPAN_SLOW */
static inline void span_of(PyObject* it, span_t* sp) {
    if (!PyUnicode_Check(it)) { sp->p = NULL; sp->len = SPAN_SLOW; return; }
    const int64_t n = (int64_t)PyUnicode_GET_LENGTH(it);
    if (PyUnicode_IS_COMPACT_ASCII(it)) { sp->p = (const void*)(((PyASCIIObject*)it) + 1); sp->len = n; return; }      /* (ready by construction) */
    sp->p = NULL; sp->len = SPAN_SLOW;
    if (!PyUnicode_IS_READY(it) || !PyUnicode_IS_COMPACT(it)) return;
    const PyCompactUnicodeObject* cu = (const PyCompactUnicodeObject*)it;
    if (cu->utf8) { sp->p = cu->utf8; sp->len = (int64_t)cu->utf8_length; return; }              /* already cached: plain bytes */
    const void* data = (const void*)(cu + 1);
    const int kind = (int)PyUnicode_KIND(it);
    int64_t b = kind == PyUnicode_1BYTE_KIND ? utf8_size_ucs1((const Py_UCS1*)data, n)
              : kind == PyUnicode_2BYTE_KIND ? utf8_size_ucs2((const Py_UCS2*)data, n) : utf8_size_ucs4((const Py_UCS4*)data, n);
    if (b < 0) return;                                                                           /* lone surrogate: the interpreter raises */
    sp->p = it; sp->len = b | SPAN_ENCODE;
}

typedef struct pack_ctx {
    PyObject** items; Py_ssize_t n; span_t* sp; int64_t* off;
    int nt; pthread_barrier_t bar;
    int64_t total[PACK_MAX_THREADS], base[PACK_MAX_THREADS];
    Py_ssize_t first_slow[PACK_MAX_THREADS];          /* first item of the thread's range that needs the interpreter, or -1 */
    char* dst;                                        /* set by the main thread between the passes; NULL: do not copy */
    int go;                                           /* start gate of the helpers: 0 wait, 1 go (nt is final), -1 leave */
} pack_ctx;
typedef struct { pack_ctx* c; int t; } pack_arg;

static void pass_a(pack_ctx* c, int t) {
    const Py_ssize_t lo = c->n * t / c->nt, hi = c->n * (t + 1) / c->nt;
    int64_t total = 0;
    Py_ssize_t slow = -1;
    for (Py_ssize_t i = lo; i < hi; ++i) {
        if (i + PACK_PREFETCH < hi) __builtin_prefetch(c->items[i + PACK_PREFETCH]);
        span_of(c->items[i], &c->sp[i]);
        if (c->sp[i].len == SPAN_SLOW) { if (slow < 0) slow = i; }
        else total += SPAN_LEN(&c->sp[i]);
    }
    c->total[t] = total;
    c->first_slow[t] = slow;
}
static void pass_c(pack_ctx* c, int t) {
    const Py_ssize_t lo = c->n * t / c->nt, hi = c->n * (t + 1) / c->nt;
    int64_t at = c->base[t];
    for (Py_ssize_t i = lo; i < hi; ++i) {
        if (c->dst && i + 8 < hi) __builtin_prefetch(c->sp[i + 8].p);
        c->off[i] = at;
        if (c->dst) span_write(&c->sp[i], c->dst + at);
        at += SPAN_LEN(&c->sp[i]);
    }
}
static void* pack_worker(void* arg) {
    pack_ctx* c = ((pack_arg*)arg)->c;
    const int t = ((pack_arg*)arg)->t;
    int go;
    while ((go = __atomic_load_n(&c->go, __ATOMIC_ACQUIRE)) == 0) sched_yield();
    if (go < 0) return NULL;
    pass_a(c, t);
    pthread_barrier_wait(&c->bar);                    /* main: slow items, base offsets, destination */
    pthread_barrier_wait(&c->bar);
    if (c->base[0] >= 0) pass_c(c, t);                /* (base[0] < 0: an exception is pending, nothing to write) */
    return NULL;
}

/* the items the helper threads could not size: the interpreter's own conversion, in index order (the first failing item raises) */
static int resolve_slow(pack_ctx* c) {
    for (int t = 0; t < c->nt; ++t) {
        if (c->first_slow[t] < 0) continue;
        const Py_ssize_t hi = c->n * (t + 1) / c->nt;
        for (Py_ssize_t i = c->first_slow[t]; i < hi; ++i) {
            if (c->sp[i].len != SPAN_SLOW) continue;
            PyObject* it = c->items[i];
            if (!PyUnicode_Check(it)) {
                if (PyTuple_Check(it) || PyList_Check(it))
                    PyErr_SetString(PyExc_NotImplementedError, "a pair or a list of words among single sequences: a batch holds one kind of input (encode_batch splits a batch that mixes them; lists of words need is_pretokenized=True)");
                else
                    PyErr_SetString(PyExc_TypeError, "TextInputSequence must be str");
                return -1;
            }
            Py_ssize_t len;
            const char* s = PyUnicode_AsUTF8AndSize(it, &len);      /* e.g. lone surrogates: UnicodeEncodeError */
            if (!s) return -1;
            c->sp[i].p = s; c->sp[i].len = len;
            c->total[t] += len;
        }
    }
    return 0;
}

/* list[str] -> CSR offsets in off[0..n] + (if `provide` hands out a destination for the total) the UTF-8 bytes and TEXT_PAD zero
 * bytes.  Returns the total or -1 with an exception set. */
typedef char* (*dst_fn)(void* user, int64_t total);
static int64_t pack_core(PyObject** items, Py_ssize_t n, int64_t* off, dst_fn provide, void* user) {
    pack_ctx* c = (pack_ctx*)calloc(1, sizeof(pack_ctx));
    span_t* sp = (span_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(span_t));
    if (!c || !sp) { free(c); free(sp); PyErr_NoMemory(); return -1; }
    static int max_threads = 0;
    if (!max_threads) {
        const char* e = getenv("TKAMD_PACK_THREADS");
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        long want = e ? atol(e) : (ncpu > PACK_DEFAULT_THREADS ? PACK_DEFAULT_THREADS : ncpu);
        max_threads = (int)(want < 1 ? 1 : (want > PACK_MAX_THREADS ? PACK_MAX_THREADS : want));
    }
    int nt = n < 16384 ? 1 : max_threads;
    c->items = items; c->n = n; c->sp = sp; c->off = off;
    pthread_t th[PACK_MAX_THREADS];
    pack_arg args[PACK_MAX_THREADS];
    int started = 0;
    /* the helpers wait at a gate until it is known how many of them could be created: the ranges and the barrier are sized for that */
    for (int t = 1; t < nt; ++t) {
        args[t].c = c; args[t].t = t;
        if (pthread_create(&th[t], NULL, pack_worker, &args[t]) != 0) break;
        started = t;
    }
    nt = started + 1;
    c->nt = nt;
    if (nt > 1 && pthread_barrier_init(&c->bar, NULL, (unsigned)nt) != 0) {
        __atomic_store_n(&c->go, -1, __ATOMIC_RELEASE);                      /* helpers leave at once */
        for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
        nt = 1; c->nt = 1; started = 0;
    }
    __atomic_store_n(&c->go, 1, __ATOMIC_RELEASE);
    int64_t total = -1;
    pass_a(c, 0);
    if (nt > 1) pthread_barrier_wait(&c->bar);
    int rc = resolve_slow(c);
    if (rc == 0) {
        int64_t acc = 0;
        for (int t = 0; t < nt; ++t) { c->base[t] = acc; acc += c->total[t]; }
        total = acc;
        c->dst = provide(user, total);                /* NULL: offsets only (the caller's buffer is too small, or allocation failed) */
        if (!c->dst && PyErr_Occurred()) { rc = -1; total = -1; }
    }
    if (rc != 0) c->base[0] = -1;
    if (nt > 1) pthread_barrier_wait(&c->bar);
    if (rc == 0) pass_c(c, 0);
    for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
    if (nt > 1) pthread_barrier_destroy(&c->bar);
    if (rc == 0) {
        off[n] = total;
        if (c->dst) memset(c->dst + total, 0, TEXT_PAD);
    }
    free(sp);
    free(c);
    return total;
}

static char* provide_bytearray(void* user, int64_t total) {
    PyObject* buf = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)total + TEXT_PAD);
    *(PyObject**)user = buf;
    return buf ? PyByteArray_AS_STRING(buf) : NULL;
}
static PyObject* pack(PyObject* self, PyObject* arg) {
    PyObject* seq = PySequence_Fast(arg, "encode_batch expects a sequence of str");
    if (!seq) return NULL;
    Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject* offs = PyByteArray_FromStringAndSize(NULL, (n + 1) * (Py_ssize_t)sizeof(int64_t));
    if (!offs) { Py_DECREF(seq); return NULL; }
    PyObject* buf = NULL;
    int64_t total = pack_core(PySequence_Fast_ITEMS(seq), n, (int64_t*)PyByteArray_AS_STRING(offs), provide_bytearray, &buf);
    Py_DECREF(seq);
    if (total < 0 || !buf) { Py_XDECREF(buf); Py_DECREF(offs); return NULL; }
    PyObject* r = PyTuple_Pack(2, buf, offs);
    Py_DECREF(buf);
    Py_DECREF(offs);
    return r;
}

/* pack_into(seq, text_addr, text_capacity, off_addr) -> total bytes.  Writes the CSR offsets (len(seq) + 1 int64) to
 * off_addr and, if total + 64 <= text_capacity, the UTF-8 bytes + 64 zero bytes to text_addr; otherwise nothing is
 * copied and the caller retries with a buffer of at least the returned size + 64.  The destination is the tokenizer
 * handle's reusable host staging (tkamd_host_staging): no allocation and no first-touch page faults per batch. */
typedef struct { char* addr; unsigned long long cap; } fixed_dst;
static char* provide_fixed(void* user, int64_t total) {
    fixed_dst* f = (fixed_dst*)user;
    return ((unsigned long long)total + TEXT_PAD <= f->cap) ? f->addr : NULL;
}
static PyObject* pack_into(PyObject* self, PyObject* args) {
    PyObject* arg;
    unsigned long long text_addr, text_cap, off_addr;
    if (!PyArg_ParseTuple(args, "OKKK", &arg, &text_addr, &text_cap, &off_addr)) return NULL;
    PyObject* seq = PySequence_Fast(arg, "encode_batch expects a sequence of str");
    if (!seq) return NULL;
    fixed_dst f = {(char*)(uintptr_t)text_addr, text_cap};
    int64_t total = pack_core(PySequence_Fast_ITEMS(seq), PySequence_Fast_GET_SIZE(seq), (int64_t*)(uintptr_t)off_addr, provide_fixed, &f);
    Py_DECREF(seq);
    return total < 0 ? NULL : PyLong_FromLongLong(total);
}

/* line_offsets(addr, n) -> bytearray of int64 CSR offsets of the '\n'-terminated lines of the n bytes at addr, every line
 * keeping its terminator -- the reference's own line reader, Lines / lines_with_ending (tokenizers/src/utils/iter.rs:64-100,
 * used by train_from_files, tokenizer/mod.rs:1432-1444: "we want to keep the \n and potential \r").  A last line without a
 * terminator is a line too; an empty buffer has no lines.  The file's bytes ARE the batch text: nothing is copied. */
static PyObject* line_offsets(PyObject* self, PyObject* args) {
    unsigned long long addr, n;
    if (!PyArg_ParseTuple(args, "KK", &addr, &n)) return NULL;
    const char* p = (const char*)(uintptr_t)addr;
    Py_ssize_t lines = 0;
    Py_BEGIN_ALLOW_THREADS
    for (const char* q = p, *e = p + n; q < e;) {
        const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
        ++lines;
        if (!nl) break;
        q = nl + 1;
    }
    Py_END_ALLOW_THREADS
    PyObject* offs = PyByteArray_FromStringAndSize(NULL, (lines + 1) * (Py_ssize_t)sizeof(int64_t));
    if (!offs) return NULL;
    int64_t* off = (int64_t*)PyByteArray_AS_STRING(offs);
    Py_ssize_t k = 0;
    off[0] = 0;
    for (const char* q = p, *e = p + n; q < e;) {
        const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
        q = nl ? nl + 1 : e;
        off[++k] = (int64_t)(q - p);
    }
    return offs;
}

/* ---- pack_encode: list[str] -> tkamd_encode_batch_paced, the packing of the batch's tail behind the H2D copy and the kernels of its head ----
 * pack_encode(seq, text_addr, text_capacity, off_addr, fn_addr, tok_addr, flags) -> (total, status, batch_addr)
 *   fn_addr: the address of tkamd_encode_batch_paced (include/tokenizers_amd.h), tok_addr: the tokenizer handle.
 * Pass A (sizes) and the CSR offsets are finished first, as in pack_into; if total + 64 > text_capacity nothing else happens and
 * (total, None, None) tells the caller to grow its buffer.  Otherwise the helper threads copy the bytes STRIPE by stripe (4 MB of
 * text, every helper its share of the stripe's items) and announce every finished stripe in `ready`, while this thread is inside
 * the library call, which waits for `ready` before it reads a slice.  The GIL is held until the library reports that it has seen the
 * whole text announced (tkamd_pace.consumed): until then the helpers read the strs' character data, and nothing may mutate the list
 * or drop a str; from there on the call only waits for the GPU, and other Python threads run. */
typedef struct tkamd_pace_c { const int64_t* ready_bytes; void (*consumed)(void*); void* user; } tkamd_pace_c;
typedef int (*paced_fn)(void* tok, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, uint32_t flags, const tkamd_pace_c* pace, void** out);
static int64_t stripe_bytes(void) {      /* 4 MB of text a stripe (TKAMD_PACK_STRIPE_KB: tests run the striped copy on small batches) */
    static int64_t v = 0;
    if (!v) { const char* e = getenv("TKAMD_PACK_STRIPE_KB"); long kb = e ? atol(e) : 4096; v = (int64_t)(kb < 4 ? 4 : kb) << 10; }
    return v;
}
#define STRIPE_BYTES (stripe_bytes())
typedef struct stripe_ctx {
    const span_t* sp; const int64_t* off; char* dst; Py_ssize_t n;
    int nh;                              /* helper threads */
    Py_ssize_t* bound; int n_stripes;    /* stripe s = items [bound[s], bound[s + 1]) */
    int* done;                           /* helpers that finished stripe s */
    int64_t ready;                       /* bytes announced */
} stripe_ctx;
static void stripe_copy(stripe_ctx* c, int h) {
    for (int s = 0; s < c->n_stripes; ++s) {
        const Py_ssize_t a = c->bound[s], b = c->bound[s + 1];
        const Py_ssize_t lo = a + (b - a) * h / c->nh, hi = a + (b - a) * (h + 1) / c->nh;
        for (Py_ssize_t i = lo; i < hi; ++i) {
            if (i + 8 < hi) __builtin_prefetch(c->sp[i + 8].p);
            span_write(&c->sp[i], c->dst + c->off[i]);
        }
        if (__atomic_add_fetch(&c->done[s], 1, __ATOMIC_ACQ_REL) == c->nh)         /* the last helper of the stripe: everything below its end is packed */
            __atomic_store_n(&c->ready, c->off[b], __ATOMIC_RELEASE);
    }
}
/* one set of helper threads for the whole of pack_encode: sizes (pass A), offsets (pass C without a destination), then -- if the main
 * thread says so -- the striped copy, which the main thread does not take part in (it is inside the library call by then) */
typedef struct pe_ctx { pack_ctx pc; stripe_ctx sc; int stripe_go; } pe_ctx;
typedef struct { pe_ctx* c; int t; } pe_arg;
static void* pe_worker(void* arg) {
    pe_ctx* c = ((pe_arg*)arg)->c;
    const int t = ((pe_arg*)arg)->t;
    int go;
    while ((go = __atomic_load_n(&c->pc.go, __ATOMIC_ACQUIRE)) == 0) sched_yield();
    if (go < 0) return NULL;
    pass_a(&c->pc, t);
    pthread_barrier_wait(&c->pc.bar);                 /* main: slow items, base offsets */
    pthread_barrier_wait(&c->pc.bar);
    if (c->pc.base[0] >= 0) pass_c(&c->pc, t);        /* (dst NULL: the offsets) */
    pthread_barrier_wait(&c->pc.bar);                 /* main: capacity, stripes */
    pthread_barrier_wait(&c->pc.bar);
    if (c->stripe_go) stripe_copy(&c->sc, t - 1);     /* helpers 1 .. nt - 1 are stripe workers 0 .. nt - 2 */
    return NULL;
}
static void pace_consumed(void* user) { *(PyThreadState**)user = PyEval_SaveThread(); }

static PyObject* pack_encode(PyObject* self, PyObject* args) {
    PyObject* arg;
    unsigned long long text_addr, text_cap, off_addr, fn_addr, tok_addr, flags;
    if (!PyArg_ParseTuple(args, "OKKKKKK", &arg, &text_addr, &text_cap, &off_addr, &fn_addr, &tok_addr, &flags)) return NULL;
    PyObject* seq = PySequence_Fast(arg, "encode_batch expects a sequence of str");
    if (!seq) return NULL;
    PyObject** items = PySequence_Fast_ITEMS(seq);
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    int64_t* off = (int64_t*)(uintptr_t)off_addr;
    char* dst = (char*)(uintptr_t)text_addr;
    pe_ctx* c = (pe_ctx*)calloc(1, sizeof(pe_ctx));
    span_t* sp = (span_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(span_t));
    if (!c || !sp) { free(c); free(sp); Py_DECREF(seq); return PyErr_NoMemory(); }
    c->pc.items = items; c->pc.n = n; c->pc.sp = sp; c->pc.off = off; c->pc.dst = NULL;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    const char* e = getenv("TKAMD_PACK_THREADS");
    long want = e ? atol(e) : (ncpu > PACK_DEFAULT_THREADS ? PACK_DEFAULT_THREADS : ncpu);
    int max_threads = (int)(want < 1 ? 1 : (want > PACK_MAX_THREADS ? PACK_MAX_THREADS : want));
    int nt = n < 16384 ? 1 : max_threads;
    pthread_t th[PACK_MAX_THREADS];
    pe_arg pargs[PACK_MAX_THREADS];
    int started = 0;
    for (int t = 1; t < nt; ++t) {
        pargs[t].c = c; pargs[t].t = t;
        if (pthread_create(&th[t], NULL, pe_worker, &pargs[t]) != 0) break;
        started = t;
    }
    nt = started + 1;
    c->pc.nt = nt;
    if (nt > 1 && pthread_barrier_init(&c->pc.bar, NULL, (unsigned)nt) != 0) {
        __atomic_store_n(&c->pc.go, -1, __ATOMIC_RELEASE);
        for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
        nt = 1; c->pc.nt = 1; started = 0;
    }
    __atomic_store_n(&c->pc.go, 1, __ATOMIC_RELEASE);
    pass_a(&c->pc, 0);
    if (nt > 1) pthread_barrier_wait(&c->pc.bar);
    int rc = resolve_slow(&c->pc);
    int64_t total = -1;
    if (rc == 0) {
        int64_t acc = 0;
        for (int t = 0; t < nt; ++t) { c->pc.base[t] = acc; acc += c->pc.total[t]; }
        total = acc;
    } else c->pc.base[0] = -1;
    if (nt > 1) pthread_barrier_wait(&c->pc.bar);
    if (rc == 0) pass_c(&c->pc, 0);
    if (nt > 1) pthread_barrier_wait(&c->pc.bar);     /* the offsets are complete */
    /* the stripes (the helpers wait at the next barrier for the decision) */
    const int fits = rc == 0 && (unsigned long long)total + TEXT_PAD <= text_cap;
    const int nh = nt - 1;
    Py_ssize_t* bound = NULL;
    int* done = NULL;
    int striped = 0;
    if (fits) {
        off[n] = total;
        if (nh > 0 && total >= 2 * STRIPE_BYTES) {
            const int n_stripes = (int)(total / STRIPE_BYTES) + 1;
            bound = (Py_ssize_t*)malloc((size_t)(n_stripes + 1) * sizeof(Py_ssize_t));
            done = (int*)calloc((size_t)n_stripes, sizeof(int));
            if (bound && done) {
                bound[0] = 0;
                for (int s = 1; s < n_stripes; ++s) {               /* first item at or behind s stripes of text (binary search over the offsets) */
                    const int64_t target = (int64_t)s * STRIPE_BYTES;
                    Py_ssize_t lo = bound[s - 1], hi = n;
                    while (lo < hi) { const Py_ssize_t mid = lo + (hi - lo) / 2; if (off[mid] < target) lo = mid + 1; else hi = mid; }
                    bound[s] = lo;
                }
                bound[n_stripes] = n;
                c->sc.sp = sp; c->sc.off = off; c->sc.dst = dst; c->sc.n = n; c->sc.nh = nh;
                c->sc.bound = bound; c->sc.n_stripes = n_stripes; c->sc.done = done;
                striped = 1;
            }
        }
    }
    c->stripe_go = striped;
    if (nt > 1) pthread_barrier_wait(&c->pc.bar);     /* helpers: copy the stripes, or leave */
    PyObject* result = NULL;
    if (rc != 0) {
        /* (the exception is set) */
    } else if (!fits) {
        result = Py_BuildValue("LOO", (long long)total, Py_None, Py_None);          /* the caller grows its buffer and comes back */
    } else {
        if (!striped) {                                                            /* a small batch: the copy is done here, before the call */
            for (Py_ssize_t i = 0; i < n; ++i) span_write(&sp[i], dst + off[i]);
            c->sc.ready = total;
        }
        memset(dst + total, 0, TEXT_PAD);
        PyThreadState* ts = NULL;
        tkamd_pace_c pace = {&c->sc.ready, pace_consumed, &ts};
        void* batch = NULL;
        const int status = ((paced_fn)(uintptr_t)fn_addr)((void*)(uintptr_t)tok_addr, (const uint8_t*)dst, off, (int64_t)n, (uint32_t)flags, &pace, &batch);
        for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
        started = 0;
        if (ts) PyEval_RestoreThread(ts);
        result = Py_BuildValue("LiK", (long long)total, status, (unsigned long long)(uintptr_t)batch);
    }
    for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
    if (nt > 1) pthread_barrier_destroy(&c->pc.bar);
    free(bound);
    free(done);
    free(sp);
    free(c);
    Py_DECREF(seq);
    return result;
}

static PyMethodDef methods[] = {
    {"pack_encode", pack_encode, METH_VARARGS, "pack_encode(seq_of_str, text_addr, text_capacity, off_addr, fn_addr, tok_addr, flags) -> (total, status | None, batch_addr | None)"},
    {"pack", pack, METH_O, "pack(seq_of_str) -> (bytearray utf8 + 64 zero bytes, bytearray int64 offsets[n+1])"},
    {"line_offsets", line_offsets, METH_VARARGS, "line_offsets(addr, n) -> bytearray int64 offsets of the lines (terminators kept)"},
    {"pack_into", pack_into, METH_VARARGS, "pack_into(seq_of_str, text_addr, text_capacity, off_addr) -> total bytes (copied iff it fits)"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef mod = {PyModuleDef_HEAD_INIT, "_marshal", "list[str] -> UTF-8 CSR marshalling", -1, methods};
PyMODINIT_FUNC PyInit__marshal(void) { return PyModule_Create(&mod); }
