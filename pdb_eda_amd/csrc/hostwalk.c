/* _hostwalk: host-side passes of an entry in C.  (1) The walk over a structure's object tree (Bio.PDB's or pdb_eda_amd.structure's)
 * that structure.Columns makes once per entry; (2) cloud_inputs, the index work of densityAnalysis._cloudInputsFixed (further down).  Host-side plumbing of the analysis (what densityAnalysis.py reads per atom at 596-603, 617-621,
 * 653-656, 966-971: residue.parent / .id / .resname / .child_list, atom.name / .occupancy / .bfactor / .coord) -- no arithmetic
 * of the path happens here.  structure.Columns falls back to its Python loops when this module is not built or meets an
 * object it does not understand (any exception raised here); tests/test_structure.py holds the two against each other.
 *
 * Built by __graft_entry__.build():  gcc -O2 -shared -fPIC -I<python include> hostwalk.c -o pdb_eda_amd/_hostwalk.so
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static PyObject *s_parent, *s_id, *s_resname, *s_child_list, *s_name, *s_occupancy, *s_bfactor, *s_coord, *s_space;

static int number_of(PyObject *o, double *out) {   /* float(o), None -> nan (as np.asarray(..., dtype=float64) reads it) */
    if (o == Py_None) { *out = NAN; return 0; }
    const double v = PyFloat_AsDouble(o);
    if (v == -1.0 && PyErr_Occurred()) return -1;
    *out = v;
    return 0;
}

/* residue_columns(residues) -> (res_model, res_chain, res_number, res_name, res_het: bytearray of 0/1, children) */
static PyObject *residue_columns(PyObject *self, PyObject *arg) {
    (void)self;
    if (!PyList_Check(arg)) { PyErr_SetString(PyExc_TypeError, "a list of residues"); return NULL; }
    const Py_ssize_t n = PyList_GET_SIZE(arg);
    PyObject *model = PyList_New(n), *chain = PyList_New(n), *number = PyList_New(n), *name = PyList_New(n), *children = PyList_New(n);
    PyObject *het = PyByteArray_FromStringAndSize(NULL, n);
    if (!model || !chain || !number || !name || !children || !het) goto fail;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *res = PyList_GET_ITEM(arg, i);
        PyObject *ch = PyObject_GetAttr(res, s_parent);
        if (!ch) goto fail;
        PyObject *ch_id = PyObject_GetAttr(ch, s_id), *mo = PyObject_GetAttr(ch, s_parent);
        Py_DECREF(ch);
        if (!ch_id || !mo) { Py_XDECREF(ch_id); Py_XDECREF(mo); goto fail; }
        PyList_SET_ITEM(chain, i, ch_id);
        PyObject *mo_id = PyObject_GetAttr(mo, s_id);
        Py_DECREF(mo);
        if (!mo_id) goto fail;
        PyList_SET_ITEM(model, i, mo_id);
        PyObject *rid = PyObject_GetAttr(res, s_id);
        if (!rid) goto fail;
        PyObject *flag = PySequence_GetItem(rid, 0), *num = PySequence_GetItem(rid, 1);
        Py_DECREF(rid);
        if (!flag || !num) { Py_XDECREF(flag); Py_XDECREF(num); goto fail; }
        PyList_SET_ITEM(number, i, num);
        const int ne = PyObject_RichCompareBool(flag, s_space, Py_NE);
        Py_DECREF(flag);
        if (ne < 0) goto fail;
        PyByteArray_AS_STRING(het)[i] = (char)ne;
        PyObject *rn = PyObject_GetAttr(res, s_resname);
        if (!rn) goto fail;
        PyList_SET_ITEM(name, i, rn);
        PyObject *cl = PyObject_GetAttr(res, s_child_list);
        if (!cl) goto fail;
        PyList_SET_ITEM(children, i, cl);
    }
    {
        PyObject *out = PyTuple_Pack(6, model, chain, number, name, het, children);
        Py_DECREF(model); Py_DECREF(chain); Py_DECREF(number); Py_DECREF(name); Py_DECREF(het); Py_DECREF(children);
        return out;
    }
fail:
    Py_XDECREF(model); Py_XDECREF(chain); Py_XDECREF(number); Py_XDECREF(name); Py_XDECREF(children); Py_XDECREF(het);
    return NULL;
}

/* atom_columns(children: list of lists of atoms) ->
 *   (atoms, name, occupancy_raw, counts: bytearray int64[n_res], occupancy: bytearray float64[n], bfactor: bytearray float64[n],
 *    coord32: bytearray float32[3 n], name_of_atom: bytearray int64[n], atom_names: list of the distinct names by first appearance) */
static PyObject *atom_columns(PyObject *self, PyObject *arg) {
    (void)self;
    if (!PyList_Check(arg)) { PyErr_SetString(PyExc_TypeError, "a list of child lists"); return NULL; }
    const Py_ssize_t n_res = PyList_GET_SIZE(arg);
    Py_ssize_t n = 0;
    for (Py_ssize_t r = 0; r < n_res; ++r) {
        PyObject *cl = PyList_GET_ITEM(arg, r);
        if (!PyList_Check(cl)) { PyErr_SetString(PyExc_TypeError, "child_list is not a list"); return NULL; }
        n += PyList_GET_SIZE(cl);
    }
    PyObject *atoms = PyList_New(n), *name = PyList_New(n), *occ_raw = PyList_New(n), *distinct = PyList_New(0), *ids = PyDict_New();
    PyObject *counts = PyByteArray_FromStringAndSize(NULL, 8 * n_res), *occ = PyByteArray_FromStringAndSize(NULL, 8 * n),
             *bfac = PyByteArray_FromStringAndSize(NULL, 8 * n), *coord = PyByteArray_FromStringAndSize(NULL, 12 * n),
             *name_id = PyByteArray_FromStringAndSize(NULL, 8 * n);
    if (!atoms || !name || !occ_raw || !distinct || !ids || !counts || !occ || !bfac || !coord || !name_id) goto fail;
    {
        int64_t *p_counts = (int64_t *)PyByteArray_AS_STRING(counts), *p_id = (int64_t *)PyByteArray_AS_STRING(name_id);
        double *p_occ = (double *)PyByteArray_AS_STRING(occ), *p_b = (double *)PyByteArray_AS_STRING(bfac);
        float *p_xyz = (float *)PyByteArray_AS_STRING(coord);
        Py_ssize_t k = 0;
        for (Py_ssize_t r = 0; r < n_res; ++r) {
            PyObject *cl = PyList_GET_ITEM(arg, r);
            const Py_ssize_t m = PyList_GET_SIZE(cl);
            p_counts[r] = (int64_t)m;
            for (Py_ssize_t j = 0; j < m; ++j, ++k) {
                if (PyList_GET_SIZE(cl) != m || k >= n) { PyErr_SetString(PyExc_RuntimeError, "the structure changed during the walk"); goto fail; }
                PyObject *atom = PyList_GET_ITEM(cl, j);
                Py_INCREF(atom);
                PyList_SET_ITEM(atoms, k, atom);
                PyObject *nm = PyObject_GetAttr(atom, s_name);
                if (!nm) goto fail;
                PyList_SET_ITEM(name, k, nm);
                PyObject *known = PyDict_GetItemWithError(ids, nm);      /* borrowed */
                if (!known) {
                    if (PyErr_Occurred()) goto fail;
                    PyObject *fresh = PyLong_FromSsize_t(PyList_GET_SIZE(distinct));
                    if (!fresh || PyDict_SetItem(ids, nm, fresh) < 0 || PyList_Append(distinct, nm) < 0) { Py_XDECREF(fresh); goto fail; }
                    p_id[k] = (int64_t)(PyList_GET_SIZE(distinct) - 1);
                    Py_DECREF(fresh);
                } else {
                    p_id[k] = (int64_t)PyLong_AsSsize_t(known);
                }
                PyObject *o = PyObject_GetAttr(atom, s_occupancy);
                if (!o) goto fail;
                PyList_SET_ITEM(occ_raw, k, o);
                if (number_of(o, p_occ + k) < 0) goto fail;
                PyObject *b = PyObject_GetAttr(atom, s_bfactor);
                if (!b) goto fail;
                const int bad = number_of(b, p_b + k);
                Py_DECREF(b);
                if (bad < 0) goto fail;
                PyObject *xyz = PyObject_GetAttr(atom, s_coord);
                if (!xyz) goto fail;
                Py_buffer view;
                if (PyObject_GetBuffer(xyz, &view, PyBUF_STRIDES | PyBUF_FORMAT) < 0) { Py_DECREF(xyz); goto fail; }
                int ok = view.ndim == 1 && view.shape[0] == 3 && view.format != NULL;
                if (ok) {
                    const char *base = (const char *)view.buf;
                    const Py_ssize_t step = view.strides ? view.strides[0] : view.itemsize;
                    const char *f = view.format;
                    if (*f == '<' || *f == '=' || *f == '@') ++f;
                    if (f[0] == 'f' && f[1] == 0 && view.itemsize == 4) {
                        for (int q = 0; q < 3; ++q) memcpy(p_xyz + 3 * k + q, base + q * step, 4);
                    } else if (f[0] == 'd' && f[1] == 0 && view.itemsize == 8) {
                        for (int q = 0; q < 3; ++q) { double v; memcpy(&v, base + q * step, 8); p_xyz[3 * k + q] = (float)v; }
                    } else {
                        ok = 0;
                    }
                }
                PyBuffer_Release(&view);
                Py_DECREF(xyz);
                if (!ok) { PyErr_SetString(PyExc_TypeError, "atom.coord is not three float32 / float64 values"); goto fail; }
            }
        }
        if (k != n) { PyErr_SetString(PyExc_RuntimeError, "the structure changed during the walk"); goto fail; }
    }
    {
        PyObject *out = PyTuple_Pack(9, atoms, name, occ_raw, counts, occ, bfac, coord, name_id, distinct);
        Py_DECREF(atoms); Py_DECREF(name); Py_DECREF(occ_raw); Py_DECREF(counts); Py_DECREF(occ); Py_DECREF(bfac); Py_DECREF(coord);
        Py_DECREF(name_id); Py_DECREF(distinct); Py_DECREF(ids);
        return out;
    }
fail:
    /* (lists that are only partly filled hold NULLs: the list deallocator copes with them) */
    Py_XDECREF(atoms); Py_XDECREF(name); Py_XDECREF(occ_raw); Py_XDECREF(counts); Py_XDECREF(occ); Py_XDECREF(bfac); Py_XDECREF(coord);
    Py_XDECREF(name_id); Py_XDECREF(distinct); Py_XDECREF(ids);
    return NULL;
}

/* ---- cloud_inputs: the index work of densityAnalysis._cloudInputsFixed (ref densityAnalysis.py:596-604, 617-621, 653-656) ----
 * In: res_of_atom int64[n], pair_of_atom int64[n], res_plain uint8[n_res] (residue.id[0] == ' '), known uint8[n_pairs] (the name has
 * an atom type), occupancy float64[n], coord32 float32[3 n], nb_off int64[n_pairs + 1] / nb int64[] (bonded names of a name, as
 * pair ids).  Out (bytearrays): rows int64 (the eligible atoms, in order), residue int32 (running number of the plain residue),
 * pair int64, key int32 ((residue, name) numbered by first appearance), alias int32 (last eligible atom with the same float32
 * coordinate), bonded_off int64[n_keys + 1], bonded int32, owner_key int32 / owner_pair int64 (every child atom of a plain
 * residue whose (residue, name) has a key), plain_residues int64.  tests/test_cloud_inputs.py holds it against the plain walk. */
typedef struct { int64_t *key; int32_t *val; size_t cap; } Map64;
static int map_init(Map64 *m, size_t n) {
    size_t cap = 16;
    while (cap < 2 * n + 8) cap <<= 1;
    m->cap = cap;
    m->key = (int64_t *)malloc(cap * sizeof(int64_t));
    m->val = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!m->key || !m->val) { free(m->key); free(m->val); m->key = NULL; m->val = NULL; return -1; }
    for (size_t i = 0; i < cap; ++i) m->key[i] = INT64_MIN;
    return 0;
}
static void map_free(Map64 *m) { free(m->key); free(m->val); }
static inline size_t map_slot(const Map64 *m, int64_t k) {
    size_t h = (size_t)((uint64_t)k * 0x9E3779B97F4A7C15ull) & (m->cap - 1);
    while (m->key[h] != INT64_MIN && m->key[h] != k) h = (h + 1) & (m->cap - 1);
    return h;
}
typedef struct { float c[3]; } Tri;
static inline uint64_t tri_hash(const float *c) {
    uint32_t b[3];
    for (int q = 0; q < 3; ++q) { float v = c[q] + 0.0f; memcpy(&b[q], &v, 4); }      /* (+0.0: -0.0 and 0.0 are one key, as for a tuple of floats) */
    return (((uint64_t)b[0] << 32) | b[1]) * 0x9E3779B97F4A7C15ull ^ ((uint64_t)b[2] * 0xC2B2AE3D27D4EB4Full);
}
static inline int tri_equal(const float *a, const float *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

static int view_of(PyObject *o, Py_buffer *v, Py_ssize_t itemsize, const char *what) {
    if (PyObject_GetBuffer(o, v, PyBUF_C_CONTIGUOUS) < 0) return -1;
    if (v->itemsize != itemsize && v->len != 0) { PyBuffer_Release(v); PyErr_Format(PyExc_TypeError, "%s: item size %zd expected", what, itemsize); return -1; }
    return 0;
}
static PyObject *bytes_of(const void *p, size_t n) { return PyByteArray_FromStringAndSize((const char *)p, (Py_ssize_t)n); }

static PyObject *cloud_inputs(PyObject *self, PyObject *args) {
    (void)self;
    PyObject *o[8];
    if (!PyArg_ParseTuple(args, "OOOOOOOO", &o[0], &o[1], &o[2], &o[3], &o[4], &o[5], &o[6], &o[7])) return NULL;
    Py_buffer v[8];
    const Py_ssize_t sizes[8] = {8, 8, 1, 1, 8, 4, 8, 8};
    const char *names[8] = {"res_of_atom", "pair_of_atom", "res_plain", "known", "occupancy", "coord32", "nb_off", "nb"};
    int got = 0;
    for (; got < 8; ++got)
        if (view_of(o[got], &v[got], sizes[got], names[got]) < 0) { for (int k = 0; k < got; ++k) PyBuffer_Release(&v[k]); return NULL; }
    PyObject *out = NULL;
    const int64_t n = v[0].len / 8, n_res = v[2].len, n_pairs = v[3].len;
    const int64_t *res_of = (const int64_t *)v[0].buf, *pair_of_atom = (const int64_t *)v[1].buf, *nb_off = (const int64_t *)v[6].buf, *nb = (const int64_t *)v[7].buf;
    const uint8_t *plain = (const uint8_t *)v[2].buf, *known = (const uint8_t *)v[3].buf;
    const double *occ = (const double *)v[4].buf;
    const float *xyz = (const float *)v[5].buf;
    const int64_t np_ = n_pairs > 0 ? n_pairs : 1;
    int64_t *ordinal = NULL, *rows = NULL, *pair = NULL, *key_code = NULL, *bonded_off = NULL, *owner_pair = NULL, *child_code = NULL, *child_pair = NULL, *plain_res = NULL;
    int32_t *residue = NULL, *key = NULL, *alias = NULL, *bonded = NULL, *owner_key = NULL;
    Map64 keys = {NULL, NULL, 0}, coords = {NULL, NULL, 0};
    if (v[1].len / 8 != n || v[4].len / 8 != n || v[5].len / 12 != n || v[6].len / 8 != n_pairs + 1) { PyErr_SetString(PyExc_ValueError, "cloud_inputs: array lengths do not agree"); goto done; }
    for (int64_t a = 0; a < n; ++a)
        if (res_of[a] < 0 || res_of[a] >= n_res || pair_of_atom[a] < 0 || pair_of_atom[a] >= n_pairs) { PyErr_SetString(PyExc_ValueError, "cloud_inputs: index out of range"); goto done; }
    ordinal = (int64_t *)malloc((size_t)(n_res + 1) * 8); plain_res = (int64_t *)malloc((size_t)(n_res + 1) * 8);
    rows = (int64_t *)malloc((size_t)(n + 1) * 8); pair = (int64_t *)malloc((size_t)(n + 1) * 8); key_code = (int64_t *)malloc((size_t)(n + 1) * 8);
    child_code = (int64_t *)malloc((size_t)(n + 1) * 8); child_pair = (int64_t *)malloc((size_t)(n + 1) * 8); owner_pair = (int64_t *)malloc((size_t)(n + 1) * 8);
    residue = (int32_t *)malloc((size_t)(n + 1) * 4); key = (int32_t *)malloc((size_t)(n + 1) * 4); alias = (int32_t *)malloc((size_t)(n + 1) * 4);
    owner_key = (int32_t *)malloc((size_t)(n + 1) * 4);
    if (!ordinal || !plain_res || !rows || !pair || !key_code || !child_code || !child_pair || !owner_pair || !residue || !key || !alias || !owner_key ||
        map_init(&keys, (size_t)n) < 0 || map_init(&coords, (size_t)n) < 0) { PyErr_NoMemory(); goto done; }
    {
        int64_t n_plain = 0, m = 0, n_child = 0, n_keys = 0, n_owner = 0, n_bonded = 0;
        for (int64_t r = 0; r < n_res; ++r) { ordinal[r] = plain[r] ? n_plain : -1; if (plain[r]) plain_res[n_plain++] = r; }
        for (int64_t a = 0; a < n; ++a) {
            const int64_t ri = ordinal[res_of[a]], p = pair_of_atom[a];
            if (ri < 0) continue;
            const int64_t code = ri * np_ + p;
            child_code[n_child] = code; child_pair[n_child] = p; ++n_child;
            if (!known[p] || occ[a] == 0.0) continue;
            rows[m] = a; residue[m] = (int32_t)ri; pair[m] = p;
            const size_t h = map_slot(&keys, code);
            if (keys.key[h] == INT64_MIN) { keys.key[h] = code; keys.val[h] = (int32_t)n_keys; key_code[n_keys++] = code; }
            key[m] = keys.val[h];
            ++m;
        }
        /* the last eligible atom of every coordinate: a second table keyed by a hash of the triple, collisions told apart by the values */
        for (int64_t i = 0; i < m; ++i) {
            const float *c = xyz + 3 * rows[i];
            int64_t hk = (int64_t)(tri_hash(c) >> 1);            /* (never INT64_MIN) */
            for (;; ++hk) {                                        /* linear re-keying on a collision of different triples */
                const size_t h = map_slot(&coords, hk);
                if (coords.key[h] == INT64_MIN) { coords.key[h] = hk; coords.val[h] = (int32_t)i; break; }
                if (tri_equal(xyz + 3 * rows[coords.val[h]], c)) { coords.val[h] = (int32_t)i; break; }
            }
        }
        for (int64_t i = 0; i < m; ++i) {
            const float *c = xyz + 3 * rows[i];
            int64_t hk = (int64_t)(tri_hash(c) >> 1);
            for (;; ++hk) {
                const size_t h = map_slot(&coords, hk);
                if (coords.key[h] == INT64_MIN) { alias[i] = (int32_t)i; break; }      /* (cannot happen: the triple was entered above) */
                if (tri_equal(xyz + 3 * rows[coords.val[h]], c)) { alias[i] = coords.val[h]; break; }
            }
        }
        /* bonded keys of every key, in key order then table order */
        bonded_off = (int64_t *)malloc((size_t)(n_keys + 1) * 8);
        if (!bonded_off) { PyErr_NoMemory(); goto done; }
        int64_t cap_b = 0;
        for (int64_t k = 0; k < n_keys; ++k) { const int64_t p = key_code[k] % np_; cap_b += nb_off[p + 1] - nb_off[p]; }
        bonded = (int32_t *)malloc((size_t)(cap_b + 1) * 4);
        if (!bonded) { PyErr_NoMemory(); goto done; }
        bonded_off[0] = 0;
        for (int64_t k = 0; k < n_keys; ++k) {
            const int64_t ri = key_code[k] / np_, p = key_code[k] % np_;
            for (int64_t q = nb_off[p]; q < nb_off[p + 1]; ++q) {
                if (nb[q] < 0 || nb[q] >= n_pairs) { PyErr_SetString(PyExc_ValueError, "cloud_inputs: bonded name out of range"); goto done; }
                const size_t h = map_slot(&keys, ri * np_ + nb[q]);
                if (keys.key[h] != INT64_MIN) bonded[n_bonded++] = keys.val[h];
            }
            bonded_off[k + 1] = n_bonded;
        }
        for (int64_t c = 0; c < n_child; ++c) {
            const size_t h = map_slot(&keys, child_code[c]);
            if (keys.key[h] != INT64_MIN) { owner_key[n_owner] = keys.val[h]; owner_pair[n_owner] = child_pair[c]; ++n_owner; }
        }
        out = Py_BuildValue("(NNNNNNNNNN)", bytes_of(rows, (size_t)m * 8), bytes_of(residue, (size_t)m * 4), bytes_of(pair, (size_t)m * 8), bytes_of(key, (size_t)m * 4),
                            bytes_of(alias, (size_t)m * 4), bytes_of(bonded_off, (size_t)(n_keys + 1) * 8), bytes_of(bonded, (size_t)n_bonded * 4),
                            bytes_of(owner_key, (size_t)n_owner * 4), bytes_of(owner_pair, (size_t)n_owner * 8), bytes_of(plain_res, (size_t)n_plain * 8));
    }
done:
    free(ordinal); free(plain_res); free(rows); free(pair); free(key_code); free(child_code); free(child_pair); free(owner_pair);
    free(residue); free(key); free(alias); free(owner_key); free(bonded_off); free(bonded);
    map_free(&keys); map_free(&coords);
    for (int k = 0; k < 8; ++k) PyBuffer_Release(&v[k]);
    return out;
}

static PyMethodDef methods[] = {
    {"residue_columns", residue_columns, METH_O, "residue_columns(residues) -> (model ids, chain ids, numbers, names, hetero flags, child lists)"},
    {"atom_columns", atom_columns, METH_O, "atom_columns(child lists) -> the per-atom columns of structure.Columns"},
    {"cloud_inputs", cloud_inputs, METH_VARARGS, "cloud_inputs(res_of_atom, pair_of_atom, res_plain, known, occupancy, coord32, nb_off, nb) -> the index arrays of pdbeda_cloud_atoms"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_hostwalk", "one-pass walk of a structure's object tree", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__hostwalk(void) {
    s_parent = PyUnicode_InternFromString("parent");
    s_id = PyUnicode_InternFromString("id");
    s_resname = PyUnicode_InternFromString("resname");
    s_child_list = PyUnicode_InternFromString("child_list");
    s_name = PyUnicode_InternFromString("name");
    s_occupancy = PyUnicode_InternFromString("occupancy");
    s_bfactor = PyUnicode_InternFromString("bfactor");
    s_coord = PyUnicode_InternFromString("coord");
    s_space = PyUnicode_InternFromString(" ");
    if (!s_parent || !s_id || !s_resname || !s_child_list || !s_name || !s_occupancy || !s_bfactor || !s_coord || !s_space) return NULL;
    return PyModule_Create(&module);
}
