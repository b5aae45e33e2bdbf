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
    PyObject *coord_objs = PyList_New(n);      /* the atoms' own coordinate objects (what the symmetry-atom tables list), kept while they are in hand */
    PyObject *counts = PyByteArray_FromStringAndSize(NULL, 8 * n_res), *occ = PyByteArray_FromStringAndSize(NULL, 8 * n),
             *bfac = PyByteArray_FromStringAndSize(NULL, 8 * n), *coord = PyByteArray_FromStringAndSize(NULL, 12 * n),
             *name_id = PyByteArray_FromStringAndSize(NULL, 8 * n);
    if (!atoms || !name || !occ_raw || !distinct || !ids || !counts || !occ || !bfac || !coord || !name_id || !coord_objs) goto fail;
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
                PyList_SET_ITEM(coord_objs, k, xyz);      /* (the list owns the reference from here) */
                Py_buffer view;
                if (PyObject_GetBuffer(xyz, &view, PyBUF_STRIDES | PyBUF_FORMAT) < 0) goto fail;
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
                if (!ok) { PyErr_SetString(PyExc_TypeError, "atom.coord is not three float32 / float64 values"); goto fail; }
            }
        }
        if (k != n) { PyErr_SetString(PyExc_RuntimeError, "the structure changed during the walk"); goto fail; }
    }
    {
        PyObject *out = PyTuple_Pack(10, atoms, name, occ_raw, counts, occ, bfac, coord, name_id, distinct, coord_objs);
        Py_DECREF(atoms); Py_DECREF(name); Py_DECREF(occ_raw); Py_DECREF(counts); Py_DECREF(occ); Py_DECREF(bfac); Py_DECREF(coord);
        Py_DECREF(name_id); Py_DECREF(distinct); Py_DECREF(ids); Py_DECREF(coord_objs);
        return out;
    }
fail:
    /* (lists that are only partly filled hold NULLs: the list deallocator copes with them) */
    Py_XDECREF(atoms); Py_XDECREF(name); Py_XDECREF(occ_raw); Py_XDECREF(counts); Py_XDECREF(occ); Py_XDECREF(bfac); Py_XDECREF(coord);
    Py_XDECREF(name_id); Py_XDECREF(distinct); Py_XDECREF(ids); Py_XDECREF(coord_objs);
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

/* ---- the statistics tail of aggregateCloud (densityAnalysis.py:734-767 of the reference; densityAnalysis._cloudStatistics here) ----
 * Per atom type: medians of nine columns of the atom table (np.nanmedian: NaNs last, the middle value or the mean of the middle
 * two), the b-factor regression (scipy.stats.linregress: slope and two-sided p-value) and the corrected fractions.  The numpy form
 * is ~60 small array calls (0.45 ms of a 2 000-atom entry's 1.3 ms); this is one pass per type over plain arrays.  Sums are taken
 * in index order, as np.bincount takes them. */

/* the regularised incomplete beta function I_x(a, b) by its continued fraction (Lentz), as in every numerical text */
static double beta_cf(double a, double b, double x) {
    const double tiny = 1e-300, eps = 1e-16;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 500; ++m) {
        const int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < eps) break;
    }
    return h;
}
static double beta_inc(double a, double b, double x) {
    if (!(x > 0.0)) return 0.0;
    if (!(x < 1.0)) return 1.0;
    const double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log(1.0 - x));
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * beta_cf(a, b, x) / a;
    return 1.0 - bt * beta_cf(b, a, 1.0 - x) / b;
}
/* Student's t distribution function at t <= 0 with df degrees of freedom (scipy.special.stdtr) */
static double stdtr_neg(double df, double t) {
    if (t != t) return t;
    if (t == 0.0) return 0.5;
    if (isinf(t)) return 0.0;
    return 0.5 * beta_inc(0.5 * df, 0.5, df / (df + t * t));
}

/* np.nanmedian of vals[0 .. m) (scratch: sorted in place; NaNs must have been left out): f applied to the one or two middle values */
typedef double (*mono_fn)(double, const double *);
static double f_id(double x, const double *p) { (void)p; return x; }
static double f_times(double x, const double *p) { return x * p[0]; }
static double f_fraction(double x, const double *p) { return (x - p[0]) / p[0]; }
static double f_ratio(double x, const double *p) { return x * p[0] + p[0]; }
/* the k-th smallest of a[0 .. m) (0-based), by selection: a is permuted so that a[k] holds it, smaller values before it, larger after */
static void select_kth(double *a, int64_t m, int64_t k) {
    int64_t lo = 0, hi = m - 1;
    while (lo < hi) {
        const double pivot = a[lo + (hi - lo) / 2];
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (a[i] < pivot) ++i;
            while (a[j] > pivot) --j;
            if (i <= j) { const double t = a[i]; a[i] = a[j]; a[j] = t; ++i; --j; }
        }
        if (k <= j) hi = j; else if (k >= i) lo = i; else return;
    }
}
/* np.nanmedian's two middle order statistics of vals[0 .. m) (no NaNs among them; permuted in place) */
static void middle_two(double *a, int64_t m, double *lo_v, double *hi_v) {
    const int64_t k = (m - 1) / 2;
    select_kth(a, m, k);
    *lo_v = a[k];
    if (m / 2 == k) { *hi_v = a[k]; return; }
    double mn = a[k + 1];                       /* the next order statistic: the smallest of what lies behind position k */
    for (int64_t i = k + 2; i < m; ++i) if (a[i] < mn) mn = a[i];
    *hi_v = mn;
}


/* nan_cutoff(values float64[n], k) -> nanmedian(values) + nanstd(values) * k as numpy computes them, to the bit (the row filter of the atom table,
 * densityAnalysis.py:731: two numpy nan-functions were 0.1 ms of an entry for their Python).  np.nanmedian: the mean of the one or two middle order
 * statistics of what is not NaN.  np.nanstd (_nanvar): NaNs become 0, avg = np.sum(arr) / count, arr -= avg, the NaN places are set back to 0,
 * var = np.sum(arr * arr) / count, sqrt -- np.sum over the whole array being numpy's pairwise sum (blocks of <= 128 with eight accumulators).
 * All NaN (or empty): NaN, and numpy's warnings are not reproduced. */
static double np_pairwise(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.0;      /* (numpy starts from -0.0 where the output is uninitialised; the sums here never meet that case: n >= 1 values or 0.0) */
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
static double np_sum(const double *a, int64_t n) {      /* add.reduce of a contiguous array: the inner loop sees at most 8192 elements a call */
    double out = 0.0;
    for (int64_t off = 0; off < n; off += 8192) out += np_pairwise(a + off, n - off < 8192 ? n - off : 8192);
    return out;
}
static PyObject *nan_cutoff(PyObject *self, PyObject *args) {
    (void)self;
    PyObject *o;
    double k = 0.0;
    if (!PyArg_ParseTuple(args, "Od", &o, &k)) return NULL;
    Py_buffer v;
    if (view_of(o, &v, 8, "values") < 0) return NULL;
    const int64_t n = v.len / 8;
    const double *x = (const double *)v.buf;
    double *w = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    if (!w) { PyBuffer_Release(&v); return PyErr_NoMemory(); }
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i) if (!isnan(x[i])) w[cnt++] = x[i];
    double result = NAN;
    if (cnt > 0) {
        double lo_v, hi_v;
        middle_two(w, cnt, &lo_v, &hi_v);
        const double median = (cnt & 1) ? lo_v : (lo_v + hi_v) / 2.0;      /* (odd: the two are the same element) */
        for (int64_t i = 0; i < n; ++i) w[i] = isnan(x[i]) ? 0.0 : x[i];
        const double avg = np_sum(w, n) / (double)cnt;
        for (int64_t i = 0; i < n; ++i) { const double d = isnan(x[i]) ? 0.0 : w[i] - avg; w[i] = d * d; }
        result = median + sqrt(np_sum(w, n) / (double)cnt) * k;
    }
    free(w);
    PyBuffer_Release(&v);
    return PyFloat_FromDouble(result);
}

static PyObject *cloud_stats(PyObject *self, PyObject *args) {
    (void)self;
    PyObject *o[6];
    int n_types = 0;
    double ratio = 0.0, unit_volume = 0.0;
    if (!PyArg_ParseTuple(args, "OiOOOOOdd", &o[0], &n_types, &o[1], &o[2], &o[3], &o[4], &o[5], &ratio, &unit_volume)) return NULL;
    Py_buffer v[6];
    const Py_ssize_t sizes[6] = {8, 8, 8, 8, 8, 8};
    const char *names[6] = {"group", "density_electron_ratio", "num_voxels", "bfactor", "centroid_distance", "table_slopes"};
    int got = 0;
    for (; got < 6; ++got)
        if (view_of(o[got], &v[got], sizes[got], names[got]) < 0) { for (int k = 0; k < got; ++k) PyBuffer_Release(&v[k]); return NULL; }
    PyObject *out = NULL;
    const int64_t n = v[0].len / 8;
    const int64_t *group = (const int64_t *)v[0].buf, *nvox = (const int64_t *)v[2].buf;
    const double *der = (const double *)v[1].buf, *bfac_in = (const double *)v[3].buf, *dist = (const double *)v[4].buf, *table_slopes = (const double *)v[5].buf;
    const int nt = n_types > 0 ? n_types : 0;
    double *row = NULL, *typ = NULL, *scratch = NULL, *logb = NULL;
    int64_t *start = NULL, *order = NULL;
    if (v[1].len / 8 != n || v[2].len / 8 != n || v[3].len / 8 != n || v[4].len / 8 != n || v[5].len / 8 != nt) { PyErr_SetString(PyExc_ValueError, "cloud_stats: array lengths do not agree"); goto done; }
    for (int64_t i = 0; i < n; ++i) if (group[i] < 0 || group[i] >= nt) { PyErr_SetString(PyExc_ValueError, "cloud_stats: type out of range"); goto done; }
    /* rows: adj, bfactor (filled), domain_fraction, corrected_fraction, corrected_ratio, volume; types: ten medians / slopes */
    row = (double *)malloc((size_t)(6 * n + 1) * 8); typ = (double *)malloc((size_t)(10 * nt + 1) * 8); scratch = (double *)malloc((size_t)(n + 1) * 8); logb = (double *)malloc((size_t)(n + 1) * 8);
    start = (int64_t *)malloc((size_t)(nt + 2) * 8); order = (int64_t *)malloc((size_t)(n + 1) * 8);
    if (!row || !typ || !scratch || !logb || !start || !order) { PyErr_NoMemory(); goto done; }
    {
        double *adj = row, *bfac = row + n, *fraction = row + 2 * n, *corrected = row + 3 * n, *corrected_ratio = row + 4 * n, *volume = row + 5 * n;
        double *m_vox = typ, *m_volume = typ + nt, *m_der = typ + 2 * nt, *m_dist = typ + 3 * nt, *m_adj = typ + 4 * nt, *m_b = typ + 5 * nt,
               *m_fraction = typ + 6 * nt, *slopes = typ + 7 * nt, *m_corrected = typ + 8 * nt, *m_corrected_ratio = typ + 9 * nt;
        /* rows in type order (stable) */
        for (int t = 0; t <= nt; ++t) start[t] = 0;
        for (int64_t i = 0; i < n; ++i) start[group[i] + 1]++;
        for (int t = 0; t < nt; ++t) start[t + 1] += start[t];
        {
            int64_t *fill = (int64_t *)scratch;      /* (n + 1 doubles hold nt <= n + 1 cursors only if nt <= n + 1: checked) */
            if (nt > n + 1) { PyErr_SetString(PyExc_ValueError, "cloud_stats: more types than rows"); goto done; }
            for (int t = 0; t < nt; ++t) fill[t] = start[t];
            for (int64_t i = 0; i < n; ++i) order[fill[group[i]]++] = i;
        }
        /* the median of a column per type, and -- dst2 -- of a weakly monotone function f2 of it (the same order statistics) */
#define MEDIAN_OF(expr, keep, dst, f2, par2, dst2) \
        for (int t = 0; t < nt; ++t) { int64_t m = 0; for (int64_t k = start[t]; k < start[t + 1]; ++k) { const int64_t i = order[k]; const double x = (expr); if ((keep) && x == x) scratch[m++] = x; } \
                                       double lo_v = NAN, hi_v = NAN; if (m > 0) middle_two(scratch, m, &lo_v, &hi_v); \
                                       (dst)[t] = m > 0 ? (lo_v + hi_v) / 2.0 : NAN; \
                                       if (dst2) ((double *)(dst2))[t] = m > 0 ? (f2(lo_v, par2) + f2(hi_v, par2)) / 2.0 : NAN; }
        MEDIAN_OF((double)nvox[i], 1, m_vox, f_times, &unit_volume, m_volume)
        for (int64_t i = 0; i < n; ++i) {
            adj[i] = der[i] / (double)nvox[i] * m_vox[group[i]];
            volume[i] = (double)nvox[i] * unit_volume;
            bfac[i] = bfac_in[i];
        }
        MEDIAN_OF(der[i], 1, m_der, f_id, NULL, (double *)NULL)
        MEDIAN_OF(dist[i], 1, m_dist, f_id, NULL, (double *)NULL)
        MEDIAN_OF(adj[i], 1, m_adj, f_fraction, &ratio, m_fraction)
        MEDIAN_OF(bfac[i], bfac[i] > 0.0, m_b, f_id, NULL, (double *)NULL)
        for (int64_t i = 0; i < n; ++i) if (bfac[i] <= 0.0) bfac[i] = m_b[group[i]];
        for (int64_t i = 0; i < n; ++i) { fraction[i] = (adj[i] - ratio) / ratio; logb[i] = log(bfac[i]); }
        /* slope of the b-factor dependence per type: scipy.stats.linregress(log b, fraction) where there is something to fit */
        for (int t = 0; t < nt; ++t) {
            const int64_t lo = start[t], hi = start[t + 1];
            const double cnt = (double)(hi - lo), safe_n = cnt > 1.0 ? cnt : 1.0;
            double sx = 0.0, sy = 0.0, b_lo = NAN, b_hi = NAN;
            int64_t n_nan = 0;
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t i = order[k];
                sx += logb[i]; sy += fraction[i];
                if (bfac[i] != bfac[i]) ++n_nan;
                else { if (!(b_lo <= bfac[i])) b_lo = bfac[i]; if (!(b_hi >= bfac[i])) b_hi = bfac[i]; }
            }
            const double xm = sx / safe_n, ym = sy / safe_n;
            double ssxm = 0.0, ssym = 0.0, ssxym = 0.0;
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t i = order[k];
                const double dx = logb[i] - xm, dy = fraction[i] - ym;
                ssxm += dx * dx; ssym += dy * dy; ssxym += dx * dy;
            }
            ssxm /= safe_n; ssym /= safe_n; ssxym /= safe_n;
            const int one_value = (n_nan == hi - lo) || (n_nan == 0 && b_lo == b_hi);
            const int fitted = (hi - lo > 2) && !one_value;
            double r = (ssxm == 0.0 || ssym == 0.0) ? 0.0 : ssxym / sqrt(ssxm * ssym);
            if (r > 1.0) r = 1.0; else if (r < -1.0) r = -1.0;   /* (a NaN stays a NaN, as np.clip leaves it) */
            const double slope = ssxym / ssxm, df = cnt - 2.0;
            const double tt = r * sqrt(df / ((1.0 - r + 1.0e-20) * (1.0 + r + 1.0e-20)));
            const double pv = 2.0 * stdtr_neg(df > 1.0 ? df : 1.0, -fabs(tt));
            slopes[t] = (fitted && !(pv > 0.05)) ? slope : table_slopes[t];
        }
        for (int64_t i = 0; i < n; ++i) {
            const int64_t g = group[i];
            corrected[i] = fraction[i] - (logb[i] - log(m_b[g])) * slopes[g];
            corrected_ratio[i] = corrected[i] * ratio + ratio;
        }
        MEDIAN_OF(corrected[i], 1, m_corrected, f_ratio, &ratio, m_corrected_ratio)
#undef MEDIAN_OF
    }
    out = Py_BuildValue("(NN)", bytes_of(row, (size_t)(6 * n) * 8), bytes_of(typ, (size_t)(10 * nt) * 8));
done:
    free(row); free(typ); free(scratch); free(logb); free(start); free(order);
    for (int k = 0; k < 6; ++k) PyBuffer_Release(&v[k]);
    return out;
}


/* table_rows(columns) -> list of row lists: what list(map(list, zip(*columns))) makes, in one pass and without the column lists of numbers.
 * A column is a list / tuple (items taken as they are), a C-contiguous numpy array -- 1-D float64 / int64 / int32 / bool: Python floats, ints, bools
 * (what .tolist() would hold); 2-D float64 (n x k): a list of k floats per row; 2-D int64 (n x k): a TUPLE of k ints per row -- or a pair (list / tuple, index array of int64): the picked items.
 * The result tables of densityAnalysis (region discrepancies, blob statistics: densityAnalysis.py:914-1035) are 2 000 rows x 16 columns; zip + list took
 * 0.25-0.5 ms a table, the columns' .tolist() another 0.1-0.2. */
#define ROWS_MAX_COLS 32
typedef struct { int kind; PyObject *fast; PyObject **items; Py_buffer view; int has_view; Py_buffer idx; int has_idx; Py_ssize_t n, k; } RowCol;   /* kind 0 objects, 1 f64, 2 i64, 3 i32, 4 bool, 5 f64 n x k */

static void rowcols_release(RowCol *c, int n) {
    for (int j = 0; j < n; ++j) {
        Py_XDECREF(c[j].fast);
        if (c[j].has_view) PyBuffer_Release(&c[j].view);
        if (c[j].has_idx) PyBuffer_Release(&c[j].idx);
    }
}

static PyObject *table_rows(PyObject *self, PyObject *arg) {
    PyObject *cols = PySequence_Fast(arg, "table_rows: a sequence of columns");
    if (!cols) return NULL;
    const Py_ssize_t nc = PySequence_Fast_GET_SIZE(cols);
    if (nc < 1 || nc > ROWS_MAX_COLS) { Py_DECREF(cols); PyErr_SetString(PyExc_ValueError, "table_rows: 1 .. 32 columns"); return NULL; }
    RowCol c[ROWS_MAX_COLS];
    memset(c, 0, sizeof c);
    int made = 0;
    PyObject *out = NULL;
    Py_ssize_t n = -1;
    for (Py_ssize_t j = 0; j < nc; ++j, ++made) {
        PyObject *o = PySequence_Fast_GET_ITEM(cols, j);
        RowCol *rc = &c[j];
        PyObject *src = o;
        if (PyTuple_Check(o) && PyTuple_GET_SIZE(o) == 2 && PyObject_CheckBuffer(PyTuple_GET_ITEM(o, 1)) && (PyList_Check(PyTuple_GET_ITEM(o, 0)) || PyTuple_Check(PyTuple_GET_ITEM(o, 0)))) {
            if (PyObject_GetBuffer(PyTuple_GET_ITEM(o, 1), &rc->idx, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) goto fail;
            rc->has_idx = 1;
            if (rc->idx.ndim != 1 || rc->idx.itemsize != 8 || !rc->idx.format || (rc->idx.format[0] != 'l' && rc->idx.format[0] != 'q')) { PyErr_SetString(PyExc_TypeError, "table_rows: an index is a 1-D int64 array"); goto fail; }
            src = PyTuple_GET_ITEM(o, 0);
        }
        if (PyList_Check(src) || PyTuple_Check(src)) {
            rc->fast = PySequence_Fast(src, "table_rows: column");
            if (!rc->fast) goto fail;
            rc->items = PySequence_Fast_ITEMS(rc->fast);
            rc->kind = 0;
            rc->n = rc->has_idx ? rc->idx.shape[0] : PySequence_Fast_GET_SIZE(rc->fast);
            if (rc->has_idx) {
                const long long *ix = (const long long *)rc->idx.buf;
                const Py_ssize_t len = PySequence_Fast_GET_SIZE(rc->fast);
                for (Py_ssize_t i = 0; i < rc->n; ++i) if (ix[i] < 0 || ix[i] >= len) { PyErr_SetString(PyExc_IndexError, "table_rows: index out of range"); goto fail; }
            }
        } else if (PyObject_CheckBuffer(src)) {
            if (PyObject_GetBuffer(src, &rc->view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) goto fail;
            rc->has_view = 1;
            const char f = rc->view.format ? rc->view.format[0] : 'B';
            if (rc->view.ndim == 1 && f == 'd') rc->kind = 1;
            else if (rc->view.ndim == 1 && (f == 'l' || f == 'q') && rc->view.itemsize == 8) rc->kind = 2;
            else if (rc->view.ndim == 1 && f == 'i' && rc->view.itemsize == 4) rc->kind = 3;
            else if (rc->view.ndim == 1 && f == '?') rc->kind = 4;
            else if (rc->view.ndim == 2 && f == 'd') { rc->kind = 5; rc->k = rc->view.shape[1]; }
            else if (rc->view.ndim == 2 && (f == 'l' || f == 'q') && rc->view.itemsize == 8) { rc->kind = 6; rc->k = rc->view.shape[1]; }
            else { PyErr_SetString(PyExc_TypeError, "table_rows: arrays are 1-D float64 / int64 / int32 / bool or 2-D float64 / int64"); goto fail; }
            rc->n = rc->view.shape[0];
        } else {
            PyErr_SetString(PyExc_TypeError, "table_rows: a column is a list, a tuple, an array or (list, index array)");
            goto fail;
        }
        if (n < 0) n = rc->n;
        else if (rc->n != n) { PyErr_SetString(PyExc_ValueError, "table_rows: columns of different lengths"); goto fail; }
    }
    out = PyList_New(n);
    if (!out) goto fail;
    /* thousands of fresh lists would run the cycle collector several times over rows that hold scalars and strings only (2 000 x 16: 0.48 ms with,
     * 0.24 without); it is switched back to what it was before the rows are handed out */
#if PY_VERSION_HEX >= 0x030A0000
    const int gc_was_on = PyGC_Disable();
#else
    const int gc_was_on = 0;      /* (no C switch for the collector before 3.10: it runs as it likes) */
#define PyGC_Enable() 0
#endif
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *row = PyList_New(nc);
        if (!row) { if (gc_was_on) PyGC_Enable(); goto fail_out; }
        PyList_SET_ITEM(out, i, row);
        for (Py_ssize_t j = 0; j < nc; ++j) {
            const RowCol *rc = &c[j];
            PyObject *v = NULL;
            switch (rc->kind) {
            case 0: v = rc->items[rc->has_idx ? ((const long long *)rc->idx.buf)[i] : i]; Py_INCREF(v); break;
            case 1: v = PyFloat_FromDouble(((const double *)rc->view.buf)[i]); break;
            case 2: v = PyLong_FromLongLong(((const long long *)rc->view.buf)[i]); break;
            case 3: v = PyLong_FromLong(((const int *)rc->view.buf)[i]); break;
            case 4: v = PyBool_FromLong(((const unsigned char *)rc->view.buf)[i]); break;
            case 6: {      /* a TUPLE of ints per row (symmetry operators: (i, j, k, op)) */
                v = PyTuple_New(rc->k);
                if (v) {
                    const long long *p = (const long long *)rc->view.buf + i * rc->k;
                    for (Py_ssize_t q = 0; q < rc->k; ++q) {
                        PyObject *f = PyLong_FromLongLong(p[q]);
                        if (!f) { Py_CLEAR(v); break; }
                        PyTuple_SET_ITEM(v, q, f);
                    }
                }
                break;
            }
            default: {
                v = PyList_New(rc->k);
                if (v) {
                    const double *p = (const double *)rc->view.buf + i * rc->k;
                    for (Py_ssize_t q = 0; q < rc->k; ++q) {
                        PyObject *f = PyFloat_FromDouble(p[q]);
                        if (!f) { Py_CLEAR(v); break; }
                        PyList_SET_ITEM(v, q, f);
                    }
                }
            }
            }
            if (!v) { if (gc_was_on) PyGC_Enable(); goto fail_out; }
            PyList_SET_ITEM(row, j, v);
        }
    }
    if (gc_was_on) PyGC_Enable();
    rowcols_release(c, made);
    Py_DECREF(cols);
    return out;
fail_out:
    Py_DECREF(out);
    rowcols_release(c, made);
    Py_DECREF(cols);
    return NULL;
fail:
    rowcols_release(c, made + 1 <= ROWS_MAX_COLS ? made + 1 : made);
    Py_DECREF(cols);
    return NULL;
}

static PyMethodDef methods[] = {
    {"residue_columns", residue_columns, METH_O, "residue_columns(residues) -> (model ids, chain ids, numbers, names, hetero flags, child lists)"},
    {"atom_columns", atom_columns, METH_O, "atom_columns(child lists) -> the per-atom columns of structure.Columns"},
    {"cloud_inputs", cloud_inputs, METH_VARARGS, "cloud_inputs(res_of_atom, pair_of_atom, res_plain, known, occupancy, coord32, nb_off, nb) -> the index arrays of pdbeda_cloud_atoms"},
    {"cloud_stats", cloud_stats, METH_VARARGS, "cloud_stats(group, n_types, density_electron_ratio, num_voxels, bfactor, centroid_distance, table_slopes, ratio, unit_volume) -> (six row columns, ten per-type columns) of aggregateCloud's statistics tail"},
    {"nan_cutoff", nan_cutoff, METH_VARARGS, "nan_cutoff(values, k) -> np.nanmedian(values) + np.nanstd(values) * k, as numpy computes them"},
    {"table_rows", table_rows, METH_O, "table_rows(columns) -> list of row lists (columns: lists, numpy arrays, or (list, int64 index array))"},
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
