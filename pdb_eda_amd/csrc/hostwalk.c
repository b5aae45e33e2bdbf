/* _hostwalk: the walk over a structure's object tree (Bio.PDB's or pdb_eda_amd.structure's) that structure.Columns makes once
 * per entry, as one pass in C.  Host-side plumbing of the analysis (what densityAnalysis.py reads per atom at 596-603, 617-621,
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

static PyMethodDef methods[] = {
    {"residue_columns", residue_columns, METH_O, "residue_columns(residues) -> (model ids, chain ids, numbers, names, hetero flags, child lists)"},
    {"atom_columns", atom_columns, METH_O, "atom_columns(child lists) -> the per-atom columns of structure.Columns"},
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
