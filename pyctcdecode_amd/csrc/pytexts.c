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

/* ---- decode_beams / decode_beams_batch: the OutputBeam lists of a packed result (ctcdec_result_pack) ----------------------
 * What BeamSearchDecoderCTC._unpack does in Python, in one C loop: per utterance a list of
 *   cls(text, last_lm_state, text_frames = [(word, (start, end)), ...], logit_score, lm_score)
 * `cls` is the (frozen) dataclass OutputBeam: instances are allocated and their five attributes set directly -- what the
 * generated __init__ does through object.__setattr__. `states`: None, or a list with the last_lm_state of every beam in
 * packed order. A text with fewer words than word frames raises IndexError, like words[j] would. */
static PyObject *k_text, *k_state, *k_frames, *k_logit, *k_lm;

static int beam_attr_names(void) {
  if (k_text) return 0;
  k_text = PyUnicode_InternFromString("text");
  k_state = PyUnicode_InternFromString("last_lm_state");
  k_frames = PyUnicode_InternFromString("text_frames");
  k_logit = PyUnicode_InternFromString("logit_score");
  k_lm = PyUnicode_InternFromString("lm_score");
  return (k_text && k_state && k_frames && k_logit && k_lm) ? 0 : -1;
}

/* [(word, (start, end)), ...] for the first n_words words of the UTF-8 text [p, p + len) */
static PyObject* word_frames(const char* p, int64_t len, int64_t n_words, const int32_t* ws, const int32_t* we) {
  PyObject* frames = PyList_New((Py_ssize_t)n_words);
  if (!frames) return NULL;
  const char* end = p + len;
  for (int64_t j = 0; j < n_words; ++j) {
    if (len == 0 || p > end) {  /* "".split(" ") is [] for an empty text; past the last word otherwise */
      Py_DECREF(frames);
      PyErr_SetString(PyExc_IndexError, "list index out of range");
      return NULL;
    }
    const char* q = (const char*)memchr(p, ' ', (size_t)(end - p));
    if (!q) q = end;
    PyObject* w = PyUnicode_DecodeUTF8(p, (Py_ssize_t)(q - p), "strict");
    PyObject* a = w ? PyLong_FromLong((long)ws[j]) : NULL;
    PyObject* b = a ? PyLong_FromLong((long)we[j]) : NULL;
    PyObject* span = b ? PyTuple_Pack(2, a, b) : NULL;
    PyObject* item = span ? PyTuple_Pack(2, w, span) : NULL;
    Py_XDECREF(w);
    Py_XDECREF(a);
    Py_XDECREF(b);
    Py_XDECREF(span);
    if (!item) {
      Py_DECREF(frames);
      return NULL;
    }
    PyList_SET_ITEM(frames, (Py_ssize_t)j, item);
    p = q + 1;
  }
  return frames;
}

/* frames_of (a callable, or None): when given, a beam's text_frames is frames_of(text, w0, w1) -- the caller's lazy view of the
 * words w0 .. w1-1 of the result, paired with the words of `text` when somebody looks -- instead of the list built here: a
 * decode_beams_batch of thousands of utterances returns tens of millions of (word, (start, end)) tuples (round 5). */
PyObject* ctcdec_py_output_beams(PyObject* cls, int64_t n_utts, const int64_t* beam_off, const int64_t* text_off,
                                 const char* text_blob, const double* logit, const double* lm, const int64_t* word_cnt_off,
                                 const int32_t* word_start, const int32_t* word_end, PyObject* states, PyObject* frames_of) {
  if (beam_attr_names() < 0) return NULL;
  if (!PyType_Check(cls)) {
    PyErr_SetString(PyExc_TypeError, "OutputBeam class expected");
    return NULL;
  }
  PyTypeObject* tp = (PyTypeObject*)cls;
  const int with_states = states != NULL && states != Py_None;
  PyObject* out = PyList_New((Py_ssize_t)n_utts);
  if (!out) return NULL;
  for (int64_t u = 0; u < n_utts; ++u) {
    const int64_t k0 = beam_off[u], k1 = beam_off[u + 1];
    PyObject* beams = PyList_New((Py_ssize_t)(k1 - k0));
    if (!beams) goto fail;
    PyList_SET_ITEM(out, (Py_ssize_t)u, beams);
    for (int64_t k = k0; k < k1; ++k) {
      const char* tp0 = text_blob + text_off[k];
      const int64_t tlen = text_off[k + 1] - text_off[k];
      const int64_t w0 = word_cnt_off[k], w1 = word_cnt_off[k + 1];
      PyObject* obj = tp->tp_alloc(tp, 0);
      if (!obj) goto fail;
      PyList_SET_ITEM(beams, (Py_ssize_t)(k - k0), obj);
      PyObject* text = PyUnicode_DecodeUTF8(tp0, (Py_ssize_t)tlen, "strict");
      const int lazy = frames_of != NULL && frames_of != Py_None;
      PyObject* frames = !text ? NULL
                         : lazy ? PyObject_CallFunction(frames_of, "OLL", text, (long long)w0, (long long)w1)
                                : word_frames(tp0, tlen, w1 - w0, word_start ? word_start + w0 : NULL, word_end ? word_end + w0 : NULL);
      PyObject* lg = frames ? PyFloat_FromDouble(logit[k]) : NULL;
      PyObject* ls = lg ? PyFloat_FromDouble(lm[k]) : NULL;
      PyObject* st = Py_None;
      if (ls && with_states) {
        st = PyList_GetItem(states, (Py_ssize_t)k); /* borrowed */
        if (!st) {
          Py_DECREF(text);
          Py_DECREF(frames);
          Py_DECREF(lg);
          Py_DECREF(ls);
          goto fail;
        }
      }
      int rc = ls ? 0 : -1;
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_text, text);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_state, st);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_frames, frames);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_logit, lg);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_lm, ls);
      Py_XDECREF(text);
      Py_XDECREF(frames);
      Py_XDECREF(lg);
      Py_XDECREF(ls);
      if (rc < 0) goto fail;
    }
  }
  return out;
fail:
  Py_DECREF(out);
  return NULL;
}

/* ---- partial_decode_beams(_batch): the LMBeam lists of a packed streaming result -----------------------------------------
 * What _DeviceStreams.unpack does per beam in Python:
 *   cls(text, next_word = "", partial_word, last_char, text_frames = [(start, end), ...], partial_frames = (ps, pe),
 *       logit_score, lm_score)
 * `cls` is the frozen dataclass LMBeam; `labels`: list of str, last_char = labels[k] for k >= 0, None below. */
static PyObject *k_next, *k_partial, *k_last, *k_pframes;

/* frames_of (a callable, or None): when given, a beam's text_frames is frames_of(w0, w1) -- the caller's lazy view of words
 * w0 .. w1-1 of the result -- instead of a list of (start, end) tuples built here: a stream that has run for a thousand frames
 * carries ~250 words per beam, and their tuples were most of the cost of reading the beams (round 5). */
PyObject* ctcdec_py_lm_beams(PyObject* cls, int64_t n_streams, const int64_t* beam_off, const int64_t* text_off,
                             const char* text_blob, const int64_t* partial_off, const char* partial_blob, const int32_t* last_char,
                             PyObject* labels, const int64_t* word_cnt_off, const int32_t* word_start, const int32_t* word_end,
                             const int32_t* pstart, const int32_t* pend, const double* logit, const double* lm, PyObject* frames_of) {
  if (beam_attr_names() < 0) return NULL;
  if (!k_next) {
    k_next = PyUnicode_InternFromString("next_word");
    k_partial = PyUnicode_InternFromString("partial_word");
    k_last = PyUnicode_InternFromString("last_char");
    k_pframes = PyUnicode_InternFromString("partial_frames");
    if (!k_next || !k_partial || !k_last || !k_pframes) return NULL;
  }
  if (!PyType_Check(cls) || !PyList_Check(labels)) {
    PyErr_SetString(PyExc_TypeError, "LMBeam class and a list of labels expected");
    return NULL;
  }
  PyTypeObject* tp = (PyTypeObject*)cls;
  PyObject* empty = PyUnicode_FromStringAndSize("", 0);
  if (!empty) return NULL;
  PyObject* out = PyList_New((Py_ssize_t)n_streams);
  if (!out) {
    Py_DECREF(empty);
    return NULL;
  }
  for (int64_t u = 0; u < n_streams; ++u) {
    const int64_t k0 = beam_off[u], k1 = beam_off[u + 1];
    PyObject* beams = PyList_New((Py_ssize_t)(k1 - k0));
    if (!beams) goto fail;
    PyList_SET_ITEM(out, (Py_ssize_t)u, beams);
    for (int64_t k = k0; k < k1; ++k) {
      PyObject* obj = tp->tp_alloc(tp, 0);
      if (!obj) goto fail;
      PyList_SET_ITEM(beams, (Py_ssize_t)(k - k0), obj);
      const int64_t w0 = word_cnt_off[k], w1 = word_cnt_off[k + 1];
      PyObject* text = PyUnicode_DecodeUTF8(text_blob + text_off[k], (Py_ssize_t)(text_off[k + 1] - text_off[k]), "strict");
      PyObject* part = text ? PyUnicode_DecodeUTF8(partial_blob + partial_off[k], (Py_ssize_t)(partial_off[k + 1] - partial_off[k]), "strict") : NULL;
      const int lazy = frames_of != NULL && frames_of != Py_None;
      PyObject* frames = !part ? NULL
                         : lazy ? PyObject_CallFunction(frames_of, "LL", (long long)w0, (long long)w1)
                                : PyList_New((Py_ssize_t)(w1 - w0));
      int ok = frames != NULL;
      for (int64_t w = w0; ok && !lazy && w < w1; ++w) {
        PyObject* a = PyLong_FromLong((long)word_start[w]);
        PyObject* b = a ? PyLong_FromLong((long)word_end[w]) : NULL;
        PyObject* span = b ? PyTuple_Pack(2, a, b) : NULL;
        Py_XDECREF(a);
        Py_XDECREF(b);
        if (!span) ok = 0;
        else PyList_SET_ITEM(frames, (Py_ssize_t)(w - w0), span);
      }
      PyObject* pa = ok ? PyLong_FromLong((long)pstart[k]) : NULL;
      PyObject* pb = pa ? PyLong_FromLong((long)pend[k]) : NULL;
      PyObject* pf = pb ? PyTuple_Pack(2, pa, pb) : NULL;
      PyObject* lg = pf ? PyFloat_FromDouble(logit[k]) : NULL;
      PyObject* ls = lg ? PyFloat_FromDouble(lm[k]) : NULL;
      PyObject* last = Py_None; /* borrowed either way */
      if (ls && last_char[k] >= 0) {
        last = PyList_GetItem(labels, (Py_ssize_t)last_char[k]);
        if (!last) ls = (Py_DECREF(ls), (PyObject*)NULL);
      }
      int rc = ls ? 0 : -1;
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_text, text);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_next, empty);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_partial, part);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_last, last);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_frames, frames);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_pframes, pf);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_logit, lg);
      if (rc == 0) rc = PyObject_GenericSetAttr(obj, k_lm, ls);
      Py_XDECREF(text);
      Py_XDECREF(part);
      Py_XDECREF(frames);
      Py_XDECREF(pa);
      Py_XDECREF(pb);
      Py_XDECREF(pf);
      Py_XDECREF(lg);
      Py_XDECREF(ls);
      if (rc < 0) goto fail;
    }
  }
  Py_DECREF(empty);
  return out;
fail:
  Py_DECREF(empty);
  Py_DECREF(out);
  return NULL;
}
