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

/* The list's str objects are scattered over the heap, so both passes are bound by cache misses on the object headers
 * and on the character data, not by the copy: pass 1 prefetches the headers a few items ahead and records (pointer,
 * length) of every item's UTF-8; pass 2 copies from those pointers on several threads.  The GIL is held throughout --
 * the helper threads touch no Python state, they only read the (immutable) character data of objects the sequence
 * keeps alive, and nothing can mutate the sequence while this thread holds the GIL. */
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>

typedef struct { const char* p; int64_t len; } span_t;
typedef struct { const span_t* sp; const int64_t* off; char* dst; Py_ssize_t lo, hi; } copy_job_t;

static void* copy_worker(void* arg) {
    const copy_job_t* j = (const copy_job_t*)arg;
    for (Py_ssize_t i = j->lo; i < j->hi; ++i) {
        if (i + 8 < j->hi) __builtin_prefetch(j->sp[i + 8].p);
        memcpy(j->dst + j->off[i], j->sp[i].p, (size_t)j->sp[i].len);
    }
    return NULL;
}

#define PACK_PREFETCH 12
#define PACK_MAX_THREADS 16

/* pass 1: type checks, (pointer, length) of every item's UTF-8, CSR offsets.  Returns the total or -1 with an exception set. */
static int64_t measure(PyObject** items, Py_ssize_t n, span_t* sp, int64_t* off) {
    int64_t total = 0;
    off[0] = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (i + PACK_PREFETCH < n) __builtin_prefetch(items[i + PACK_PREFETCH]);
        PyObject* it = items[i];
        if (!PyUnicode_Check(it)) {
            if (PyTuple_Check(it) || PyList_Check(it))
                PyErr_SetString(PyExc_NotImplementedError, "pair / pre-tokenized inputs are outside the MI355X hot path");
            else
                PyErr_SetString(PyExc_TypeError, "TextInputSequence must be str");
            return -1;
        }
        Py_ssize_t len;
        const char* s = PyUnicode_AsUTF8AndSize(it, &len);      /* materialises the cached UTF-8 of a non-ASCII str */
        if (!s) return -1;                                      /* e.g. lone surrogates */
        sp[i].p = s;
        sp[i].len = len;
        total += len;
        off[i + 1] = total;
    }
    return total;
}

/* pass 2: copy, on several threads for big batches; zero the TEXT_PAD bytes after the text */
static void copy_all(const span_t* sp, const int64_t* off, Py_ssize_t n, int64_t total, char* dst) {
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = (int)(ncpu < 1 ? 1 : (ncpu > PACK_MAX_THREADS ? PACK_MAX_THREADS : ncpu));
    if (n < 65536 || total < (4 << 20)) nt = 1;
    copy_job_t jobs[PACK_MAX_THREADS];
    pthread_t th[PACK_MAX_THREADS];
    int started = 0;
    for (int t = 0; t < nt; ++t) {
        jobs[t].sp = sp; jobs[t].off = off; jobs[t].dst = dst;
        jobs[t].lo = n * t / nt; jobs[t].hi = n * (t + 1) / nt;
    }
    for (int t = 1; t < nt; ++t) {
        if (pthread_create(&th[t], NULL, copy_worker, &jobs[t]) != 0) break;
        started = t;
    }
    copy_worker(&jobs[0]);
    for (int t = started + 1; t < nt; ++t) copy_worker(&jobs[t]);          /* threads that could not be created: do their share here */
    for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
    memset(dst + total, 0, TEXT_PAD);
}

static PyObject* pack(PyObject* self, PyObject* arg) {
    PyObject* seq = PySequence_Fast(arg, "encode_batch expects a sequence of str");
    if (!seq) return NULL;
    Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject* offs = PyByteArray_FromStringAndSize(NULL, (n + 1) * (Py_ssize_t)sizeof(int64_t));
    span_t* sp = (span_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(span_t));
    if (!offs || !sp) { Py_XDECREF(offs); free(sp); Py_DECREF(seq); return sp ? NULL : PyErr_NoMemory(); }
    int64_t* off = (int64_t*)PyByteArray_AS_STRING(offs);
    int64_t total = measure(PySequence_Fast_ITEMS(seq), n, sp, off);
    PyObject* buf = total < 0 ? NULL : PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)total + TEXT_PAD);
    if (!buf) { Py_DECREF(offs); free(sp); Py_DECREF(seq); return NULL; }
    copy_all(sp, off, n, total, PyByteArray_AS_STRING(buf));
    free(sp);
    Py_DECREF(seq);
    PyObject* r = PyTuple_Pack(2, buf, offs);
    Py_DECREF(buf);
    Py_DECREF(offs);
    return r;
}

/* pack_into(seq, text_addr, text_capacity, off_addr) -> total bytes.  Writes the CSR offsets (len(seq) + 1 int64) to
 * off_addr and, if total + 64 <= text_capacity, the UTF-8 bytes + 64 zero bytes to text_addr; otherwise nothing is
 * copied and the caller retries with a buffer of at least the returned size + 64.  The destination is the tokenizer
 * handle's reusable host staging (tkamd_host_staging): no allocation and no first-touch page faults per batch. */
static PyObject* pack_into(PyObject* self, PyObject* args) {
    PyObject* arg;
    unsigned long long text_addr, text_cap, off_addr;
    if (!PyArg_ParseTuple(args, "OKKK", &arg, &text_addr, &text_cap, &off_addr)) return NULL;
    PyObject* seq = PySequence_Fast(arg, "encode_batch expects a sequence of str");
    if (!seq) return NULL;
    Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    span_t* sp = (span_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(span_t));
    if (!sp) { Py_DECREF(seq); return PyErr_NoMemory(); }
    int64_t* off = (int64_t*)(uintptr_t)off_addr;
    int64_t total = measure(PySequence_Fast_ITEMS(seq), n, sp, off);
    if (total >= 0 && (unsigned long long)total + TEXT_PAD <= text_cap) copy_all(sp, off, n, total, (char*)(uintptr_t)text_addr);
    free(sp);
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

static PyMethodDef methods[] = {
    {"pack", pack, METH_O, "pack(seq_of_str) -> (bytearray utf8 + 64 zero bytes, bytearray int64 offsets[n+1])"},
    {"line_offsets", line_offsets, METH_VARARGS, "line_offsets(addr, n) -> bytearray int64 offsets of the lines (terminators kept)"},
    {"pack_into", pack_into, METH_VARARGS, "pack_into(seq_of_str, text_addr, text_capacity, off_addr) -> total bytes (copied iff it fits)"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef mod = {PyModuleDef_HEAD_INIT, "_marshal", "list[str] -> UTF-8 CSR marshalling", -1, methods};
PyMODINIT_FUNC PyInit__marshal(void) { return PyModule_Create(&mod); }
