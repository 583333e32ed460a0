/* Host-side result building for a batch of searches (CPython extension, no GPU code): the hits of B queries are k row
 * numbers each; the reference's answer carries, per query, the ids, documents and metadata dicts of those rows
 * (embedder.py:604-609 reshapes Chroma's lists of lists).  In Python this is B*k list indexings, dict copies and list
 * appends per table -- 2 ms per 256-query call, all of it under the interpreter lock, which is what concurrent callers
 * of the service contend for.  Here it is one pass in C: the same objects, the same order. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* gather(rows, k, ids, docs, metas) -> (ids_ll, docs_ll | None, metas_ll | None)
 * rows: C-contiguous int64 buffer of n*k row numbers, every one a hit (0 <= row < len(table));
 * ids / docs / metas: the index's row tables (lists); docs and metas may be None (column not asked for).
 * metadata dicts are COPIED (a caller may change what it gets), ids and documents are shared (immutable). */
static PyObject *gather(PyObject *self, PyObject *args) {
    PyObject *rows_obj, *ids_t, *docs_t, *metas_t;
    Py_ssize_t k;
    if (!PyArg_ParseTuple(args, "OnOOO", &rows_obj, &k, &ids_t, &docs_t, &metas_t)) return NULL;
    if (k <= 0 || !PyList_Check(ids_t) || (docs_t != Py_None && !PyList_Check(docs_t)) ||
        (metas_t != Py_None && !PyList_Check(metas_t))) {
        PyErr_SetString(PyExc_TypeError, "gather(rows, k > 0, ids: list, docs: list | None, metas: list | None)");
        return NULL;
    }
    Py_buffer view;
    if (PyObject_GetBuffer(rows_obj, &view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return NULL;
    PyObject *out_ids = NULL, *out_docs = NULL, *out_metas = NULL, *res = NULL;
    if (view.itemsize != 8 || view.len % (8 * k) != 0) {
        PyErr_SetString(PyExc_ValueError, "rows must be a contiguous int64 buffer of n * k entries");
        goto done;
    }
    {
        const int64_t *rows = (const int64_t *)view.buf;
        const Py_ssize_t n = view.len / (8 * k);
        const Py_ssize_t n_rows = PyList_GET_SIZE(ids_t);
        if ((docs_t != Py_None && PyList_GET_SIZE(docs_t) < n_rows) ||
            (metas_t != Py_None && PyList_GET_SIZE(metas_t) < n_rows)) {
            PyErr_SetString(PyExc_ValueError, "row tables of different lengths");
            goto done;
        }
        out_ids = PyList_New(n);
        out_docs = docs_t != Py_None ? PyList_New(n) : NULL;
        out_metas = metas_t != Py_None ? PyList_New(n) : NULL;
        if (!out_ids || (docs_t != Py_None && !out_docs) || (metas_t != Py_None && !out_metas)) goto fail;
        for (Py_ssize_t b = 0; b < n; ++b) {
            PyObject *li = PyList_New(k), *ld = out_docs ? PyList_New(k) : NULL, *lm = out_metas ? PyList_New(k) : NULL;
            if (!li || (out_docs && !ld) || (out_metas && !lm)) {
                Py_XDECREF(li); Py_XDECREF(ld); Py_XDECREF(lm);
                goto fail;
            }
            PyList_SET_ITEM(out_ids, b, li);
            if (ld) PyList_SET_ITEM(out_docs, b, ld);
            if (lm) PyList_SET_ITEM(out_metas, b, lm);
            for (Py_ssize_t j = 0; j < k; ++j) {
                const int64_t r = rows[b * k + j];
                if (r < 0 || r >= n_rows) {
                    PyErr_Format(PyExc_IndexError, "row %lld outside the table of %zd rows", (long long)r, n_rows);
                    goto fail;
                }
                PyObject *o = PyList_GET_ITEM(ids_t, r);
                Py_INCREF(o);
                PyList_SET_ITEM(li, j, o);
                if (ld) {
                    o = PyList_GET_ITEM(docs_t, r);
                    Py_INCREF(o);
                    PyList_SET_ITEM(ld, j, o);
                }
                if (lm) {
                    PyObject *m = PyList_GET_ITEM(metas_t, r);
                    PyObject *c = PyDict_Check(m) ? PyDict_Copy(m) : PyObject_CallFunctionObjArgs((PyObject *)&PyDict_Type, m, NULL);
                    if (!c) goto fail;
                    PyList_SET_ITEM(lm, j, c);
                }
            }
        }
        res = PyTuple_Pack(3, out_ids, out_docs ? out_docs : Py_None, out_metas ? out_metas : Py_None);
    }
fail:
    /* (lists that were only partly filled hold NULL slots: list dealloc copes with them) */
    Py_XDECREF(out_ids);
    Py_XDECREF(out_docs);
    Py_XDECREF(out_metas);
done:
    PyBuffer_Release(&view);
    return res;
}

/* split(keys, columns) -> [ {keys[0]: columns[0][b], ...} for b in range(len(columns[0])) ]   (columns: lists of equal length) */
static PyObject *split(PyObject *self, PyObject *args) {
    PyObject *keys, *cols;
    if (!PyArg_ParseTuple(args, "O!O!", &PyTuple_Type, &keys, &PyTuple_Type, &cols)) return NULL;
    const Py_ssize_t nk = PyTuple_GET_SIZE(keys);
    if (nk == 0 || PyTuple_GET_SIZE(cols) != nk) {
        PyErr_SetString(PyExc_ValueError, "split(keys, columns): one column per key");
        return NULL;
    }
    Py_ssize_t n = -1;
    for (Py_ssize_t c = 0; c < nk; ++c) {
        PyObject *col = PyTuple_GET_ITEM(cols, c);
        if (!PyList_Check(col) || (n >= 0 && PyList_GET_SIZE(col) != n)) {
            PyErr_SetString(PyExc_ValueError, "split: columns must be lists of one length");
            return NULL;
        }
        n = PyList_GET_SIZE(col);
    }
    PyObject *out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t b = 0; b < n; ++b) {
        PyObject *d = _PyDict_NewPresized(nk);
        if (!d) { Py_DECREF(out); return NULL; }
        PyList_SET_ITEM(out, b, d);
        for (Py_ssize_t c = 0; c < nk; ++c)
            if (PyDict_SetItem(d, PyTuple_GET_ITEM(keys, c), PyList_GET_ITEM(PyTuple_GET_ITEM(cols, c), b)) != 0) {
                Py_DECREF(out);
                return NULL;
            }
    }
    return out;
}

static PyMethodDef methods[] = {
    {"gather", gather, METH_VARARGS, "rows of a batch's hits -> (ids, documents, metadata copies) as lists of lists"},
    {"split", split, METH_VARARGS, "columns -> one dict per query"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_hostrows", "result building for batched searches", -1, methods};
PyMODINIT_FUNC PyInit__hostrows(void) { return PyModule_Create(&module); }
