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

static PyObject* pack(PyObject* self, PyObject* arg) {
    PyObject* seq = PySequence_Fast(arg, "encode_batch expects a sequence of str");
    if (!seq) return NULL;
    Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject** items = PySequence_Fast_ITEMS(seq);
    PyObject* offs = PyByteArray_FromStringAndSize(NULL, (n + 1) * (Py_ssize_t)sizeof(int64_t));
    if (!offs) { Py_DECREF(seq); return NULL; }
    int64_t* off = (int64_t*)PyByteArray_AS_STRING(offs);
    /* pass 1: sizes (also materialises the cached UTF-8 of every str) */
    int64_t total = 0;
    off[0] = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* it = items[i];
        if (!PyUnicode_Check(it)) {
            if (PyTuple_Check(it) || PyList_Check(it))
                PyErr_SetString(PyExc_NotImplementedError, "pair / pre-tokenized inputs are outside the MI355X hot path");
            else
                PyErr_SetString(PyExc_TypeError, "TextInputSequence must be str");
            Py_DECREF(offs); Py_DECREF(seq);
            return NULL;
        }
        Py_ssize_t len;
        if (!PyUnicode_AsUTF8AndSize(it, &len)) { Py_DECREF(offs); Py_DECREF(seq); return NULL; }   /* e.g. lone surrogates */
        total += len;
        off[i + 1] = total;
    }
    PyObject* buf = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)total + TEXT_PAD);
    if (!buf) { Py_DECREF(offs); Py_DECREF(seq); return NULL; }
    char* dst = PyByteArray_AS_STRING(buf);
    /* pass 2: copy */
    for (Py_ssize_t i = 0; i < n; ++i) {
        Py_ssize_t len;
        const char* s = PyUnicode_AsUTF8AndSize(items[i], &len);
        memcpy(dst + off[i], s, (size_t)len);
    }
    memset(dst + total, 0, TEXT_PAD);
    Py_DECREF(seq);
    PyObject* r = PyTuple_Pack(2, buf, offs);
    Py_DECREF(buf);
    Py_DECREF(offs);
    return r;
}

static PyMethodDef methods[] = {
    {"pack", pack, METH_O, "pack(seq_of_str) -> (bytearray utf8 + 64 zero bytes, bytearray int64 offsets[n+1])"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef mod = {PyModuleDef_HEAD_INIT, "_marshal", "list[str] -> UTF-8 CSR marshalling", -1, methods};
PyMODINIT_FUNC PyInit__marshal(void) { return PyModule_Create(&mod); }
