/* pytexts.c -- the one piece of the Python shell that is C: a result's texts as a Python list of str, built straight
 * from the blocks the library holds (ctcdec_result_texts_joined), one PyUnicode_DecodeUTF8 per utterance instead of
 * string_at + decode + split over the whole blob (three passes and three copies of ~4 MB at 4096 utterances).
 * Loaded through ctypes.PyDLL (the GIL is held); never linked into libctcdec.so, which stays free of Python. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* blob: n texts separated by `sep`; returns a new list (NULL + exception on failure) */
PyObject* ctcdec_py_split_texts(const char* blob, int64_t nbytes, int64_t n, char sep) {
  PyObject* list = PyList_New((Py_ssize_t)n);
  if (!list) return NULL;
  const char* p = blob;
  const char* end = blob + nbytes;
  for (int64_t i = 0; i < n; ++i) {
    const char* q = (i + 1 < n) ? (const char*)memchr(p, sep, (size_t)(end - p)) : end;
    if (!q) {
      Py_DECREF(list);
      PyErr_SetString(PyExc_ValueError, "fewer texts in the blob than announced");
      return NULL;
    }
    PyObject* s = PyUnicode_DecodeUTF8(p, (Py_ssize_t)(q - p), "strict");
    if (!s) {
      Py_DECREF(list);
      return NULL;
    }
    PyList_SET_ITEM(list, (Py_ssize_t)i, s);
    p = q + 1;
  }
  return list;
}

/* the same from n blocks (off[i], len[i]) of one pool (ctcdec_result_text_blocks) */
PyObject* ctcdec_py_texts_from_blocks(const char* pool, const int64_t* off, const int64_t* len, int64_t n) {
  PyObject* list = PyList_New((Py_ssize_t)n);
  if (!list) return NULL;
  for (int64_t i = 0; i < n; ++i) {
    PyObject* s = PyUnicode_DecodeUTF8(pool + off[i], (Py_ssize_t)len[i], "strict");
    if (!s) {
      Py_DECREF(list);
      return NULL;
    }
    PyList_SET_ITEM(list, (Py_ssize_t)i, s);
  }
  return list;
}
