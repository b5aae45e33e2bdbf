/*
 * pdbeda_oracle.c -- CPU restatement of the pdb_eda voxel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (pdb_eda_amd/, the C-ABI
 * library, the HIP kernels) may include, link, import or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as
 * the checker / reported CPU baseline.
 *
 * Parity status: PINNED.  The restatement is checked against golden vectors
 * produced by the reference itself (its Cython cutils + ccp4.py + densityAnalysis.py,
 * imported in the build container by tests/golden/refload.py; fixtures in
 * tests/golden/ *.npz, generator tests/golden/make_golden.py) by
 * tests/test_oracle_golden.py; the aggregateCloud composite at the end of this file by
 * tests/test_oracle_cloud.py (five reference runs of DensityAnalysis, three of them at the
 * BASELINE sizes: tests/golden/make_golden_analysis.py, make_golden_big.py).  The reference's own tests need the network and pin
 * nothing offline.
 *
 * Every function cites the reference lines (relative to /root/reference/) it
 * restates.  The algorithmic difference to the reference is deliberate and
 * result-neutral: clustering is an O(N) union-find over a bounding-box grid instead
 * of the reference's O(N^2) scipy cdist matrix (cutils.pyx:44-70); cluster
 * membership and emission order are identical (proved by the golden vectors).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct ora_map {
    int32_t ncrs[3];         /* header.ncrs (columns, rows, sections)          ccp4.py:168 */
    int32_t crs_start[3];    /* header.crsStart                                 ccp4.py:186 */
    int32_t xyz_interval[3]; /* header.xyzInterval                              ccp4.py:227 */
    int32_t map2xyz[3];      /* header.map2xyz                                  ccp4.py:230-234 */
    int32_t map2crs[3];      /* header.map2crs                                  ccp4.py:235 */
    int32_t crs_interval[3]; /* header.crsInterval                              ccp4.py:237 */
    int32_t unique_ncrs[3];  /* header.uniqueNcrs                               ccp4.py:262-269 */
    int32_t orthogonal;      /* alpha == beta == gamma == 90                    ccp4.py:297,313 */
    double ortho[9];         /* header.orthoMat, row major                      ccp4.py:248-250 */
    double deortho[9];       /* header.deOrthoMat, row major                    ccp4.py:252-253 */
    double origin[3];        /* header.origin                                   ccp4.py:272-286 */
    double grid_len[3];      /* header.gridLength                               ccp4.py:228 */
    double unit_volume;      /* header.unitVolume                               ccp4.py:243-244 */
    const float *density;    /* [s][r][c], c fastest                            ccp4.py:338 */
} ora_map;

/* ---- floor division as Python's int(np.floor(a / b) * b) for the small ints here ---- */
static int64_t floordiv(int64_t a, int64_t b) {
    int64_t q = a / b, r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) q -= 1;
    return q;
}

/* cutils.pyx:125-145 getPointDensityFromCrs / cutils.pyx:147-167 testValidCrs.
 * Returns 1 and writes the wrapped index when the voxel is stored, 0 otherwise. */
static int wrap_crs(const ora_map *m, const int32_t crs_in[3], int32_t out[3]) {
    for (int i = 0; i < 3; ++i) {
        int64_t v = crs_in[i];
        if (v < 0 || v >= m->ncrs[i])
            v -= floordiv(v, m->crs_interval[i]) * (int64_t)m->crs_interval[i];
        if ((m->ncrs[i] <= v && v < m->crs_interval[i]) || v < 0) return 0;
        out[i] = (int32_t)v;
    }
    return 1;
}

double ora_point_density(const ora_map *m, const int32_t crs[3]) {
    int32_t w[3];
    if (!wrap_crs(m, crs, w)) return 0.0;
    return (double)m->density[((int64_t)w[2] * m->ncrs[1] + w[1]) * m->ncrs[0] + w[0]];
}

int ora_valid_crs(const ora_map *m, const int32_t crs[3]) {
    int32_t w[3];
    return wrap_crs(m, crs, w);
}

/* 3x3 * 3 product in the accumulation order numpy's np.dot (float64 matrix x vector ->
 * OpenBLAS dgemv) produces for this shape in the build container, identified from the
 * golden vectors (triclinic case, bit exact): fma(a2, v2, fma(a0, v0, a1 * v1)).  This is
 * BLAS-kernel specific; membership only depends on it at |dist - r| ~ 1e-15. */
static void matvec3(const double a[9], const double v[3], double out[3]) {
    for (int i = 0; i < 3; ++i) {
        volatile double p1 = a[3 * i + 1] * v[1];
        out[i] = fma(a[3 * i + 2], v[2], fma(a[3 * i + 0], v[0], p1));
    }
}

/* ccp4.py:304-316 crs2xyzCoord */
void ora_crs2xyz(const ora_map *m, const int32_t crs[3], double xyz[3]) {
    if (m->orthogonal) {
        for (int i = 0; i < 3; ++i) {
            volatile double p = (double)crs[m->map2xyz[i]] * m->grid_len[i]; /* unfused */
            xyz[i] = p + m->origin[i];
        }
    } else {
        double f[3];
        for (int i = 0; i < 3; ++i)
            f[i] = (double)((int64_t)crs[m->map2xyz[i]] + m->crs_start[m->map2xyz[i]]) / (double)m->xyz_interval[i];
        matvec3(m->ortho, f, xyz);
    }
}

/* Python round(): half to even -> rint under the default rounding mode. */
static int64_t pyround(double x) { return (int64_t)rint(x); }

/* ccp4.py:288-302 xyz2crsCoord */
void ora_xyz2crs(const ora_map *m, const double xyz[3], int32_t crs[3]) {
    int64_t g[3];
    if (m->orthogonal) {
        for (int i = 0; i < 3; ++i) {
            volatile double d = xyz[i] - m->origin[i];
            g[i] = pyround(d / m->grid_len[i]);
        }
    } else {
        double f[3];
        matvec3(m->deortho, xyz, f);
        for (int i = 0; i < 3; ++i) {
            volatile double p = f[i] * (double)m->xyz_interval[i];
            g[i] = pyround(p) - m->crs_start[m->map2xyz[i]];
        }
    }
    for (int i = 0; i < 3; ++i) crs[i] = (int32_t)g[m->map2crs[i]];
}

/* cutils.pyx:185-203 createFullCrsList.  `cutoff` arrives as a C float (Q1).
 * Output order: itertools.product(range(nc'), range(nr'), range(ns')) = c slowest.
 * Returns the number of voxels (also when out == NULL), -1 for cutoff == 0 (None). */
int64_t ora_full_crs_list(const ora_map *m, float cutoff, int32_t *out, int64_t cap) {
    const double cut = (double)cutoff;
    if (cut == 0.0) return -1;
    int64_t n = 0;
    for (int32_t c = 0; c < m->unique_ncrs[0]; ++c)
        for (int32_t r = 0; r < m->unique_ncrs[1]; ++r)
            for (int32_t s = 0; s < m->unique_ncrs[2]; ++s) {
                int32_t crs[3] = {c, r, s};
                double d = ora_point_density(m, crs);
                int keep = cut > 0.0 ? (d >= cut) : (d <= cut);
                if (keep) {
                    if (out && n < cap) { out[3 * n] = c; out[3 * n + 1] = r; out[3 * n + 2] = s; }
                    ++n;
                }
            }
    return n;
}

/* ---- union-find ---- */
static int64_t uf_find(int64_t *p, int64_t x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}
static void uf_union(int64_t *p, int64_t a, int64_t b) {
    a = uf_find(p, a); b = uf_find(p, b);
    if (a == b) return;
    if (a < b) p[b] = a; else p[a] = b; /* root = lowest list index */
}

/* cutils.pyx:44-70 createCrsLists.  Points closer than sqrt(3) (= 26-neighbours on
 * integer coordinates, raw / non-periodic, Q3) are clustered; clusters are emitted in
 * order of their lowest list index.  cluster_of[i] = emission index of point i's
 * cluster.  Returns the number of clusters, or -1 on allocation failure. */
int64_t ora_cluster(const int32_t *crs, int64_t n, int32_t *cluster_of) {
    if (n <= 0) return 0;
    int32_t lo[3], hi[3];
    for (int k = 0; k < 3; ++k) lo[k] = hi[k] = crs[k];
    for (int64_t i = 1; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            if (crs[3 * i + k] < lo[k]) lo[k] = crs[3 * i + k];
            if (crs[3 * i + k] > hi[k]) hi[k] = crs[3 * i + k];
        }
    int64_t d0 = (int64_t)hi[0] - lo[0] + 3, d1 = (int64_t)hi[1] - lo[1] + 3, d2 = (int64_t)hi[2] - lo[2] + 3;
    int64_t vol = d0 * d1 * d2;
    int64_t *grid = (int64_t *)malloc(sizeof(int64_t) * (size_t)vol);
    int64_t *par = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    if (!grid || !par) { free(grid); free(par); return -1; }
    for (int64_t i = 0; i < vol; ++i) grid[i] = -1;
    for (int64_t i = 0; i < n; ++i) {
        par[i] = i;
        int64_t g = (((int64_t)crs[3 * i + 2] - lo[2] + 1) * d1 + ((int64_t)crs[3 * i + 1] - lo[1] + 1)) * d0 +
                    ((int64_t)crs[3 * i] - lo[0] + 1);
        if (grid[g] >= 0) par[i] = grid[g]; /* duplicate point: distance 0 -> same cluster */
        else grid[g] = i;
    }
    for (int64_t i = 0; i < n; ++i) {
        int64_t c = (int64_t)crs[3 * i] - lo[0] + 1, r = (int64_t)crs[3 * i + 1] - lo[1] + 1, s = (int64_t)crs[3 * i + 2] - lo[2] + 1;
        for (int ds = -1; ds <= 1; ++ds)
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) {
                    if (!ds && !dr && !dc) continue;
                    int64_t j = grid[((s + ds) * d1 + (r + dr)) * d0 + (c + dc)];
                    if (j >= 0) uf_union(par, i, j);
                }
    }
    /* emission order = ascending lowest list index = ascending root (root is the min index) */
    int64_t ncl = 0;
    int64_t *rank = grid; /* reuse: rank[root] for roots, indexed by list index (< n <= vol) */
    for (int64_t i = 0; i < n; ++i)
        if (uf_find(par, i) == i) rank[i] = ncl++;
    for (int64_t i = 0; i < n; ++i) cluster_of[i] = (int32_t)rank[uf_find(par, i)];
    free(grid); free(par);
    return ncl;
}

/* ccp4.py:522-545 DensityBlob.fromCrsList: sequential fp64 sums over the list.
 * stats = {totalDensity, centroid[3], coordCenter[3], volume}. */
void ora_blob_stats(const ora_map *m, const int32_t *crs, int64_t n, double stats[8]) {
    double w[3] = {0, 0, 0}, cc[3] = {0, 0, 0}, total = 0;
    for (int64_t i = 0; i < n; ++i) {
        double d = ora_point_density(m, crs + 3 * i), xyz[3];
        ora_crs2xyz(m, crs + 3 * i, xyz);
        for (int k = 0; k < 3; ++k) { w[k] = w[k] + d * xyz[k]; cc[k] += xyz[k]; }
        total += d;
    }
    stats[0] = total;
    for (int k = 0; k < 3; ++k) { stats[1 + k] = w[k] / total; stats[4 + k] = cc[k] / (double)n; }
    stats[7] = m->unit_volume * (double)n;
}

/* cutils.pyx:205-218 + 220-248 getSphereCrsFromXyz.  radius and density_cutoff arrive
 * as C floats (Q1); the box is the asymmetric [C-R-1, C+R] (Q4); the density filter is
 * strict (Q2); the distance is fp64 sqrt of squares <= radius.  Output in the
 * reference's iteration order (c slowest).  Returns the count (also when out == NULL). */
int64_t ora_sphere_crs(const ora_map *m, const double xyz[3], float radius, float density_cutoff,
                       int32_t *out, int64_t cap) {
    const double rad = (double)radius, cut = (double)density_cutoff;
    int32_t C[3], R[3];
    ora_xyz2crs(m, xyz, C);
    double o[3];
    for (int i = 0; i < 3; ++i) { volatile double t = m->origin[i] + rad; o[i] = t; } /* origin + [r,r,r] */
    ora_xyz2crs(m, o, R);
    int64_t n = 0;
    for (int32_t c = C[0] - R[0] - 1; c < C[0] + R[0] + 1; ++c)
        for (int32_t r = C[1] - R[1] - 1; r < C[1] + R[1] + 1; ++r)
            for (int32_t s = C[2] - R[2] - 1; s < C[2] + R[2] + 1; ++s) {
                int32_t crs[3] = {c, r, s};
                double d = ora_point_density(m, crs);
                if (!((0.0 < cut && cut < d) || (d < cut && cut < 0.0) || cut == 0.0)) continue;
                double p[3];
                ora_crs2xyz(m, crs, p);
                volatile double dx = p[0] - xyz[0], dy = p[1] - xyz[1], dz = p[2] - xyz[2];
                volatile double xx = dx * dx, yy = dy * dy, zz = dz * dz;
                volatile double ss = xx + yy;
                double dist = sqrt(ss + zz);
                if (dist <= rad) {
                    if (out && n < cap) { out[3 * n] = c; out[3 * n + 1] = r; out[3 * n + 2] = s; }
                    ++n;
                }
            }
    return n;
}

static int cmp_crs(const void *a, const void *b) {
    const int32_t *x = (const int32_t *)a, *y = (const int32_t *)b;
    for (int k = 0; k < 3; ++k) { if (x[k] < y[k]) return -1; if (x[k] > y[k]) return 1; }
    return 0;
}

/* cutils.pyx:250-271 getSphereCrsFromXyzList: set union over atoms on RAW crs tuples.
 * radii: per-atom (C float each after the reference's per-call cast).  Output sorted
 * lexicographically (the reference returns an unordered set).  Returns the count, or
 * -(needed) if cap is too small. */
int64_t ora_sphere_crs_list(const ora_map *m, const double *xyz, const float *radii, int64_t n_atoms,
                            float density_cutoff, int32_t *out, int64_t cap) {
    int64_t n = 0;
    for (int64_t a = 0; a < n_atoms; ++a) {
        int64_t k = ora_sphere_crs(m, xyz + 3 * a, radii[a], density_cutoff, out ? out + 3 * n : NULL, out ? cap - n : 0);
        if (out && n + k > cap) return -(n + k);
        n += k;
    }
    if (!out) return n;
    qsort(out, (size_t)n, 3 * sizeof(int32_t), cmp_crs);
    int64_t u = 0;
    for (int64_t i = 0; i < n; ++i)
        if (i == 0 || cmp_crs(out + 3 * i, out + 3 * (u - 1)) != 0) {
            memmove(out + 3 * u, out + 3 * i, 3 * sizeof(int32_t));
            ++u;
        }
    return u;
}

/* cutils.pyx:273-292 testValidXyz: every voxel of the (cutoff-free) sphere is stored. */
int ora_valid_xyz(const ora_map *m, const double xyz[3], float radius) {
    int64_t n = ora_sphere_crs(m, xyz, radius, 0.0f, NULL, 0);
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(n > 0 ? n : 1));
    ora_sphere_crs(m, xyz, radius, 0.0f, buf, n);
    int ok = 1;
    for (int64_t i = 0; i < n && ok; ++i) ok = ora_valid_crs(m, buf + 3 * i);
    free(buf);
    return ok;
}

/* cutils.pyx:28-39 sumOfAbs over ALL stored voxels (ccp4.py:365-376), strict, sequential. */
double ora_sum_of_abs(const float *a, int64_t n, float cutoff) {
    const double cut = (double)cutoff;
    double s = 0;
    for (int64_t i = 0; i < n; ++i) {
        double v = fabs((double)a[i]);
        if (v > cut) s += v;
    }
    return s;
}

/* ccp4.py:343-363 meanDensity / stdDensity = np.mean / np.std of the tuple of Python floats, i.e. of a contiguous float64
 * array.  The summation tree belongs to a third-party dependency that is not under /root/reference: numpy (2.2.6 in the
 * environment the golden vectors were made in).  Its published algorithm, restated here:
 *   - add.reduce hands the inner loop at most `bufsize` = 8192 elements at a time and accumulates the calls in order:
 *     out = 0; for each block of 8192: out += pairwise_sum(block)           (numpy/_core/src/umath/reduction.c, ufunc bufsize)
 *   - pairwise_sum(a, n) (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum):
 *       n < 8:    plain left-to-right sum
 *       n <= 128: 8 interleaved accumulators r[j] += a[8 i + j], combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
 *                 then the n % 8 tail added one by one
 *       else:     n2 = n / 2 rounded down to a multiple of 8;  pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2)
 *   - np.mean = sum / n;  np.std (numpy/_core/_methods.py _var): m = sum / n; x = a - m; x = x * x; sqrt(sum(x) / n).
 * mode 0: a[i]   mode 1: (a[i] - shift)^2.   Pinned by the `mean` / `std` of every golden case (== , not approx). */
static double np_elem(const float *a, int64_t i, int mode, double shift) {
    double v = (double)a[i];
    if (mode == 1) { v = v - shift; v = v * v; }
    return v;
}
static double np_pairwise(const float *a, int64_t off, int64_t n, int mode, double shift) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += np_elem(a, off + i, mode, shift);
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = np_elem(a, off + j, mode, shift);
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += np_elem(a, off + i + j, mode, shift);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += np_elem(a, off + i, mode, shift);
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, off, n2, mode, shift) + np_pairwise(a, off + n2, n - n2, mode, shift);
}
static double np_add_reduce(const float *a, int64_t n, int mode, double shift) {
    double out = 0.0;
    for (int64_t off = 0; off < n; off += 8192) out += np_pairwise(a, off, n - off < 8192 ? n - off : 8192, mode, shift);
    return out;
}
void ora_mean_std(const float *a, int64_t n, double *mean, double *std) {
    const double m = np_add_reduce(a, n, 0, 0.0) / (double)n;
    *mean = m;
    *std = sqrt(np_add_reduce(a, n, 1, m) / (double)n);
}

/* cutils.pyx:8-25 testOverlap: any pair with Chebyshev distance <= 1. */
int ora_test_overlap(const int32_t *a, int64_t na, const int32_t *b, int64_t nb) {
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < nb; ++j) {
            int ok = 1;
            for (int k = 0; k < 3 && ok; ++k) {
                int64_t d = (int64_t)a[3 * i + k] - b[3 * j + k];
                ok = (d >= -1 && d <= 1);
            }
            if (ok) return 1;
        }
    return 0;
}

/* cutils.pyx:73-103 createSymmetryAtoms (+ densityAnalysis.py:885-912 for the box).
 * coords: f32 atom coordinates promoted to double by the caller; rot: n_ops x 3 x 4.
 * Emits, in the reference's itertools.product order (i, j, k, op), atom index, symmetry
 * 4-tuple and coordinate.  Identity (0,0,0,0) keeps every atom with its original coord.
 * Returns count (also when outputs are NULL). */
int64_t ora_symmetry_atoms(const double *coords, int64_t n_atoms, const double *rot, int32_t n_ops,
                           const double ortho[9], const double bbox_lo[3], const double bbox_hi[3],
                           int32_t *out_atom, int32_t *out_sym, double *out_xyz, int64_t cap) {
    int64_t n = 0;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
            for (int k = -1; k <= 1; ++k)
                for (int32_t op = 0; op < n_ops; ++op) {
                    const double *rm = rot + 12 * op;
                    double ijk[3] = {(double)i, (double)j, (double)k}, ot[3];
                    matvec3(ortho, ijk, ot);
                    int identity = (i == 0 && j == 0 && k == 0 && op == 0);
                    for (int64_t a = 0; a < n_atoms; ++a) {
                        double x[3];
                        if (identity) {
                            for (int q = 0; q < 3; ++q) x[q] = coords[3 * a + q];
                        } else {
                            for (int q = 0; q < 3; ++q) {
                                volatile double p0 = rm[4 * q + 0] * coords[3 * a + 0];
                                volatile double p1 = rm[4 * q + 1] * coords[3 * a + 1];
                                volatile double p2 = rm[4 * q + 2] * coords[3 * a + 2];
                                volatile double s = p0 + p1;
                                volatile double d = s + p2;
                                volatile double t = d + rm[4 * q + 3];
                                x[q] = t + ot[q];
                            }
                            int in = 1;
                            for (int q = 0; q < 3; ++q)
                                in = in && (bbox_lo[q] - 5 <= x[q]) && (x[q] <= bbox_hi[q] + 5);
                            if (!in) continue;
                        }
                        if (out_atom && n < cap) {
                            out_atom[n] = (int32_t)a;
                            out_sym[4 * n] = i; out_sym[4 * n + 1] = j; out_sym[4 * n + 2] = k; out_sym[4 * n + 3] = op;
                            for (int q = 0; q < 3; ++q) out_xyz[3 * n + q] = x[q];
                        }
                        ++n;
                    }
                }
    return n;
}

/* ---------------------------------------------------------------------------------
 * Composite: the full-map blob path the bench's cpu_baseline leg times
 * (createFullCrsList -> createCrsLists -> fromCrsList, ccp4.py:463-485), restated with
 * an O(N) two-pass union-find directly on the thresholded sub-volume so that it can run
 * at 256^3 / 1.5 sigma, which the reference's O(N^2) clustering cannot.
 * Outputs per blob (in the reference's emission order = ascending c-major first key):
 * n, stats[8] as ora_blob_stats, first_key.  labels (optional): int32 [us][ur][uc]
 * blob index or -1.  Returns the number of blobs, -1 for cutoff == 0, -2 on OOM,
 * -(needed) if cap is too small.
 * --------------------------------------------------------------------------------- */
typedef struct { int64_t key; int64_t root; } keyroot;
static int cmp_keyroot(const void *a, const void *b) {
    int64_t x = ((const keyroot *)a)->key, y = ((const keyroot *)b)->key;
    return x < y ? -1 : (x > y ? 1 : 0);
}

int64_t ora_full_blobs(const ora_map *m, float cutoff, int64_t *out_n, double *out_stats, int64_t *out_key,
                       int64_t cap, int32_t *labels) {
    const double cut = (double)cutoff;
    if (cut == 0.0) return -1;
    const int64_t uc = m->unique_ncrs[0], ur = m->unique_ncrs[1], us = m->unique_ncrs[2];
    const int64_t nvox = uc * ur * us;
    int64_t *par = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nvox > 0 ? nvox : 1));
    if (!par) return -2;
    /* threshold (inclusive, Q2); inside [0,unique) no wrap is needed (unique <= ncrs) */
    for (int64_t s = 0; s < us; ++s)
        for (int64_t r = 0; r < ur; ++r)
            for (int64_t c = 0; c < uc; ++c) {
                double d = (double)m->density[(s * m->ncrs[1] + r) * m->ncrs[0] + c];
                int keep = cut > 0.0 ? (d >= cut) : (d <= cut);
                int64_t v = (s * ur + r) * uc + c;
                par[v] = keep ? v : -1;
            }
    /* union with the 13 already-visited neighbours */
    for (int64_t s = 0; s < us; ++s)
        for (int64_t r = 0; r < ur; ++r)
            for (int64_t c = 0; c < uc; ++c) {
                int64_t v = (s * ur + r) * uc + c;
                if (par[v] < 0) continue;
                for (int ds = -1; ds <= 0; ++ds)
                    for (int dr = -1; dr <= 1; ++dr)
                        for (int dc = -1; dc <= 1; ++dc) {
                            if (ds == 0 && (dr > 0 || (dr == 0 && dc >= 0))) continue;
                            int64_t s2 = s + ds, r2 = r + dr, c2 = c + dc;
                            if (s2 < 0 || r2 < 0 || r2 >= ur || c2 < 0 || c2 >= uc) continue;
                            int64_t u = (s2 * ur + r2) * uc + c2;
                            if (par[u] >= 0) uf_union(par, v, u);
                        }
            }
    /* per-root accumulation */
    int64_t nroot = 0;
    for (int64_t v = 0; v < nvox; ++v)
        if (par[v] == v) ++nroot;
    if (nroot > cap) { free(par); return -nroot; }
    int64_t *slot = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nvox > 0 ? nvox : 1));
    keyroot *kr = (keyroot *)malloc(sizeof(keyroot) * (size_t)(nroot > 0 ? nroot : 1));
    double *acc = (double *)calloc((size_t)(nroot > 0 ? nroot : 1) * 8, sizeof(double));
    int64_t *cnt = (int64_t *)calloc((size_t)(nroot > 0 ? nroot : 1), sizeof(int64_t));
    if (!slot || !kr || !acc || !cnt) { free(par); free(slot); free(kr); free(acc); free(cnt); return -2; }
    int64_t k = 0;
    for (int64_t v = 0; v < nvox; ++v)
        if (par[v] == v) { slot[v] = k; kr[k].root = v; kr[k].key = INT64_MAX; ++k; }
    for (int64_t s = 0; s < us; ++s)
        for (int64_t r = 0; r < ur; ++r)
            for (int64_t c = 0; c < uc; ++c) {
                int64_t v = (s * ur + r) * uc + c;
                if (par[v] < 0) continue;
                int64_t b = slot[uf_find(par, v)];
                int32_t crs[3] = {(int32_t)c, (int32_t)r, (int32_t)s};
                double xyz[3], d = (double)m->density[(s * m->ncrs[1] + r) * m->ncrs[0] + c];
                ora_crs2xyz(m, crs, xyz);
                acc[8 * b] += d;
                for (int q = 0; q < 3; ++q) { acc[8 * b + 1 + q] += d * xyz[q]; acc[8 * b + 4 + q] += xyz[q]; }
                cnt[b] += 1;
                int64_t key = (c * ur + r) * us + s; /* c-major position in createFullCrsList */
                if (key < kr[b].key) kr[b].key = key;
            }
    for (int64_t b = 0; b < nroot; ++b) kr[b].root = b;
    qsort(kr, (size_t)nroot, sizeof(keyroot), cmp_keyroot);
    int64_t *rank = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nroot > 0 ? nroot : 1));
    for (int64_t i = 0; i < nroot; ++i) {
        int64_t b = kr[i].root;
        rank[b] = i;
        out_n[i] = cnt[b];
        out_key[i] = kr[i].key;
        out_stats[8 * i] = acc[8 * b];
        for (int q = 0; q < 3; ++q) {
            out_stats[8 * i + 1 + q] = acc[8 * b + 1 + q] / acc[8 * b];
            out_stats[8 * i + 4 + q] = acc[8 * b + 4 + q] / (double)cnt[b];
        }
        out_stats[8 * i + 7] = m->unit_volume * (double)cnt[b];
    }
    if (labels)
        for (int64_t v = 0; v < nvox; ++v) labels[v] = par[v] < 0 ? -1 : (int32_t)rank[slot[uf_find(par, v)]];
    free(par); free(slot); free(kr); free(acc); free(cnt); free(rank);
    return nroot;
}

/* ---------------------------------------------------------------------------------
 * Composite: DensityAnalysis.aggregateCloud up to its statistics tail (densityAnalysis.py:571-731), on the flattened
 * structure the product hands to pdbeda_aggregate_cloud (include/pdbeda.h: pdbeda_cloud_atoms; the flattening itself is
 * checked against a statement-by-statement walk of the reference's loops in tests/test_cloud_inputs.py).  Checker for
 * entries of thousands of atoms, where the reference itself takes minutes; pinned on the reference's analysis goldens by
 * tests/test_oracle_cloud.py.  Two phases, because the cut-off between them is numpy's (np.nanmedian + 2.5 np.nanstd, 607):
 * the caller computes it with numpy itself from the distances of phase 1.
 * Atoms that share a coordinate (round 4; pinned on tests/golden/analysis_alias.npz): allAtomClouds is keyed by
 * tuple(atom.coord) (605), so phase 2 hands every atom of a coordinate the clouds of the LAST eligible one (alias[i]; found
 * with that atom's radius) -- the same DensityBlob objects, whose .atoms the atom loop overwrites (639): a residue's pool may
 * hold one object twice, naming only the later of its two atoms.  Phase 1's distances (606) are each atom's own.
 * --------------------------------------------------------------------------------- */
typedef struct ora_cloud1 {      /* one cloud: sorted distinct voxels + fromCrsList statistics */
    int32_t *crs;
    int64_t n;
    double total, centroid[3];
    int32_t lo[3], hi[3];
    int32_t *atoms;              /* indices of the eligible atoms behind it (cloud.atoms), distinct */
    int64_t n_atoms;
} ora_cloud1;

typedef struct ora_cloud_state {
    const ora_map *m;
    int64_t n;
    const double *xyz, *weight;
    const int32_t *residue, *key, *alias;
    int64_t n_keys;
    const int64_t *bonded_off;
    const int32_t *bonded;
    int64_t n_owners;
    const int32_t *owner_key;
    ora_cloud1 **clouds;         /* per atom: its clouds, in createBlobList order */
    int64_t *n_clouds;
} ora_cloud_state;

static void cloud_finish(const ora_map *m, ora_cloud1 *c) {
    double st[8];
    ora_blob_stats(m, c->crs, c->n, st);
    c->total = st[0];
    for (int k = 0; k < 3; ++k) { c->centroid[k] = st[1 + k]; c->lo[k] = c->hi[k] = c->crs[k]; }
    for (int64_t i = 1; i < c->n; ++i)
        for (int k = 0; k < 3; ++k) {
            if (c->crs[3 * i + k] < c->lo[k]) c->lo[k] = c->crs[3 * i + k];
            if (c->crs[3 * i + k] > c->hi[k]) c->hi[k] = c->crs[3 * i + k];
        }
}

static void cloud_free1(ora_cloud1 *c) { free(c->crs); free(c->atoms); c->crs = NULL; c->atoms = NULL; }

/* utils.testOverlap (cutils.pyx:8-25) behind a bounding-box rejection */
static int cloud_overlap(const ora_cloud1 *a, const ora_cloud1 *b) {
    for (int k = 0; k < 3; ++k)
        if (a->lo[k] > b->hi[k] + 1 || b->lo[k] > a->hi[k] + 1) return 0;
    return ora_test_overlap(a->crs, a->n, b->crs, b->n);
}

/* DensityBlob.clone + merge (ccp4.py:575-594) of the clouds pool[idx[0..k)]: set union of the voxels, union of the atoms,
 * statistics recomputed over the union (fromCrsList). */
static int cloud_union(const ora_map *m, ora_cloud1 *const *pool, const int64_t *idx, int64_t k, ora_cloud1 *out) {
    int64_t nv = 0, na = 0;
    for (int64_t i = 0; i < k; ++i) { nv += pool[idx[i]]->n; na += pool[idx[i]]->n_atoms; }
    out->crs = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(nv > 0 ? nv : 1));
    out->atoms = (int32_t *)malloc(sizeof(int32_t) * (size_t)(na > 0 ? na : 1));
    if (!out->crs || !out->atoms) return -1;
    nv = 0; na = 0;
    for (int64_t i = 0; i < k; ++i) {
        memcpy(out->crs + 3 * nv, pool[idx[i]]->crs, sizeof(int32_t) * 3 * (size_t)pool[idx[i]]->n);
        nv += pool[idx[i]]->n;
        for (int64_t j = 0; j < pool[idx[i]]->n_atoms; ++j) {
            int seen = 0;
            for (int64_t q = 0; q < na; ++q) seen |= out->atoms[q] == pool[idx[i]]->atoms[j];
            if (!seen) out->atoms[na++] = pool[idx[i]]->atoms[j];
        }
    }
    qsort(out->crs, (size_t)nv, sizeof(int32_t) * 3, cmp_crs);
    int64_t u = 0;
    for (int64_t i = 0; i < nv; ++i)
        if (i == 0 || cmp_crs(out->crs + 3 * i, out->crs + 3 * (i - 1)) != 0) { if (u != i) memcpy(out->crs + 3 * u, out->crs + 3 * i, sizeof(int32_t) * 3); ++u; }
    out->n = u;
    out->n_atoms = na;
    cloud_finish(m, out);
    return 0;
}

void ora_cloud_end(ora_cloud_state *s) {
    if (!s) return;
    if (s->clouds)
        for (int64_t i = 0; i < s->n; ++i) {
            for (int64_t j = 0; s->clouds[i] && j < s->n_clouds[i]; ++j) cloud_free1(&s->clouds[i][j]);
            free(s->clouds[i]);
        }
    free(s->clouds); free(s->n_clouds); free(s);
}

/* Phase 1 (596-606): the clouds of every eligible atom (findAberrantBlobs with the atom's own radius); min_distance[i] = the
 * distance from the atom to the nearest of its clouds' centroids (np.linalg.norm), NaN for an atom without clouds. */
ora_cloud_state *ora_cloud_begin(const ora_map *m, int64_t n, const double *xyz, const float *radius, const double *weight, const int32_t *residue,
                                 const int32_t *alias, const int32_t *key, int64_t n_keys, const int64_t *bonded_off, const int32_t *bonded,
                                 int64_t n_owners, const int32_t *owner_key, float density_cutoff, double *min_distance) {
    for (int64_t i = 0; i < n; ++i) if (alias[i] < 0 || alias[i] >= n) return NULL;
    ora_cloud_state *s = (ora_cloud_state *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->m = m; s->n = n; s->xyz = xyz; s->weight = weight; s->residue = residue; s->key = key; s->alias = alias; s->n_keys = n_keys;
    s->bonded_off = bonded_off; s->bonded = bonded; s->n_owners = n_owners; s->owner_key = owner_key;
    s->clouds = (ora_cloud1 **)calloc((size_t)(n > 0 ? n : 1), sizeof *s->clouds);
    s->n_clouds = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof *s->n_clouds);
    if (!s->clouds || !s->n_clouds) { ora_cloud_end(s); return NULL; }
    for (int64_t i = 0; i < n; ++i) {
        min_distance[i] = NAN;
        const int64_t nv = ora_sphere_crs(m, xyz + 3 * i, radius[i], density_cutoff, NULL, 0);
        if (nv == 0) continue;
        int32_t *crs = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nv), *lab = (int32_t *)malloc(sizeof(int32_t) * (size_t)nv);
        if (!crs || !lab) { free(crs); free(lab); ora_cloud_end(s); return NULL; }
        ora_sphere_crs(m, xyz + 3 * i, radius[i], density_cutoff, crs, nv);
        const int64_t nc = ora_cluster(crs, nv, lab);
        s->clouds[i] = (ora_cloud1 *)calloc((size_t)nc, sizeof(ora_cloud1));
        s->n_clouds[i] = nc;
        for (int64_t c = 0; c < nc; ++c) {
            ora_cloud1 *cl = &s->clouds[i][c];
            int64_t cnt = 0;
            for (int64_t v = 0; v < nv; ++v) cnt += lab[v] == c;
            cl->crs = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)cnt);
            cl->atoms = (int32_t *)malloc(sizeof(int32_t));
            cl->atoms[0] = (int32_t)i; cl->n_atoms = 1;           /* aCloud.atoms = [atom] (639) */
            cl->n = 0;
            for (int64_t v = 0; v < nv; ++v)
                if (lab[v] == c) { memcpy(cl->crs + 3 * cl->n, crs + 3 * v, sizeof(int32_t) * 3); ++cl->n; }
            cloud_finish(m, cl);                                   /* (list order: the sphere's iteration order, as the reference sums) */
            double d2 = 0;
            for (int k = 0; k < 3; ++k) { const double d = xyz[3 * i + k] - cl->centroid[k]; d2 += d * d; }
            const double d = sqrt(d2);
            if (!(min_distance[i] <= d)) min_distance[i] = d;
        }
        free(crs); free(lab);
    }
    return s;
}

/* connected components of the overlap graph over pool[0..n): comp[i] = component number in order of lowest member */
static int64_t overlap_components(ora_cloud1 *const *pool, int64_t n, int64_t *comp) {
    int64_t *par = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) par[i] = i;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j)
            if (uf_find(par, i) != uf_find(par, j) && cloud_overlap(pool[i], pool[j])) uf_union(par, i, j);
    int64_t nc = 0;
    for (int64_t i = 0; i < n; ++i) comp[i] = -1;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = uf_find(par, i);
        if (comp[r] < 0) comp[r] = nc++;
        comp[i] = comp[r];
    }
    free(par);
    return nc;
}

/* Phase 2 (609-726).  Row buffers are caller-allocated: atom rows <= n, residue / domain rows <= total clouds (cap_rows each).
 * counts[3] = atom rows, residue rows, domain rows; totals[3] = numVoxels, totalElectrons, totalDensity (718-721).
 * Residue rows in the reference's emission order up to the set order inside a cluster (lowest pooled index first). */
int ora_cloud_finish(ora_cloud_state *s, double centroid_cutoff, double min_cloud_electrons, int64_t cap_rows,
                     int32_t *atom_idx, double *atom_total, int64_t *atom_n, double *atom_centroid, double *atom_distance,
                     int32_t *res_residue, double *res_total, int64_t *res_n, double *res_electrons, double *res_centroid,
                     int32_t *dom_residue, double *dom_total, int64_t *dom_n, double *dom_electrons, double *dom_centroid,
                     uint8_t *owner_state, int64_t counts[3], double totals[3]) {
    const ora_map *m = s->m;
    int64_t total_clouds = 0;
    for (int64_t i = 0; i < s->n; ++i) total_clouds += s->n_clouds[i];
    ora_cloud1 *dom_pool = (ora_cloud1 *)calloc((size_t)(total_clouds > 0 ? total_clouds : 1), sizeof(ora_cloud1));
    int32_t *dom_res = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total_clouds > 0 ? total_clouds : 1));
    ora_cloud1 **pool = (ora_cloud1 **)malloc(sizeof(ora_cloud1 *) * (size_t)(total_clouds > 0 ? total_clouds : 1));
    int64_t *comp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total_clouds > 0 ? total_clouds : 1));
    int64_t *members = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total_clouds > 0 ? total_clouds : 1));
    int64_t *first_of_key = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s->n_keys > 0 ? s->n_keys : 1));
    int64_t *count_of_key = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s->n_keys > 0 ? s->n_keys : 1));
    if (!dom_pool || !dom_res || !pool || !comp || !members || !first_of_key || !count_of_key) return -1;
    for (int64_t k = 0; k < s->n_keys; ++k) { first_of_key[k] = -1; count_of_key[k] = 0; }
    for (int64_t o = 0; o < s->n_owners; ++o) owner_state[o] = 0;
    int64_t n_atom_rows = 0, n_res_rows = 0, n_dom_pool = 0, owner_cursor = 0;
    int64_t a0 = 0;
    while (a0 < s->n) {                                  /* one residue: atoms [a0, a1) */
        int64_t a1 = a0;
        while (a1 < s->n && s->residue[a1] == s->residue[a0]) ++a1;
        int64_t n_pool = 0;
        for (int64_t i = a0; i < a1; ++i) {
            const int64_t ia = s->alias[i];                         /* allAtomClouds[tuple(atom.coord)] (622): the last atom of this coordinate */
            ora_cloud1 *const mine = s->clouds[ia];
            const int64_t nc = s->n_clouds[ia];
            if (nc == 0) continue;
            int64_t best = 0;
            double best_d = 0;
            {
                double dmin = INFINITY;
                for (int64_t c = 0; c < nc; ++c) {
                    double d2 = 0;
                    for (int k = 0; k < 3; ++k) { const double d = s->xyz[3 * i + k] - mine[c].centroid[k]; d2 += d * d; }
                    const double d = sqrt(d2);
                    if (d < dmin) { dmin = d; best = c; }          /* distances.index(min): first */
                }
                best_d = dmin;
                if (nc > 1 && dmin > centroid_cutoff) continue;     /* (627-629; a single cloud is never tested, 622-623) */
            }
            first_of_key[s->key[i]] = n_pool;                       /* atomCloudIndeces[resAtom]: the LAST atom of that name wins (640) */
            count_of_key[s->key[i]] = nc;
            for (int64_t c = 0; c < nc; ++c) {
                mine[c].atoms[0] = (int32_t)i;                      /* aCloud.atoms = [atom] (639): the shared object now names THIS atom */
                pool[n_pool++] = &mine[c];
            }
            if (n_atom_rows < cap_rows) {
                const ora_cloud1 *b = &mine[best];
                atom_idx[n_atom_rows] = (int32_t)i; atom_total[n_atom_rows] = b->total; atom_n[n_atom_rows] = b->n;
                for (int k = 0; k < 3; ++k) atom_centroid[3 * n_atom_rows + k] = b->centroid[k];
                atom_distance[n_atom_rows] = best_d;
            }
            ++n_atom_rows;
        }
        /* bonded-atom overlap completeness (652-659): the owners of this residue are the next ones whose key belongs to it */
        const int32_t res_id = s->residue[a0];
        (void)res_id;
        /* keys of this residue: those of its atoms; owners are listed residue by residue, so consume while the key is one of them */
        while (owner_cursor < s->n_owners) {
            const int32_t k = s->owner_key[owner_cursor];
            int mine = 0;
            for (int64_t i = a0; i < a1 && !mine; ++i) mine = s->key[i] == k;
            if (!mine) break;
            if (first_of_key[k] >= 0) {
                int all_ok = 1;
                for (int64_t b = s->bonded_off[k]; b < s->bonded_off[k + 1] && all_ok; ++b) {
                    const int32_t k2 = s->bonded[b];
                    if (first_of_key[k2] < 0) continue;              /* "if resAtom2 in atomCloudIndeces" */
                    int any = 0;
                    for (int64_t i1 = first_of_key[k]; i1 < first_of_key[k] + count_of_key[k] && !any; ++i1)
                        for (int64_t i2 = first_of_key[k2]; i2 < first_of_key[k2] + count_of_key[k2] && !any; ++i2)
                            any = i1 != i2 && cloud_overlap(pool[i1], pool[i2]);   /* (overlap[i][i] stays 0, 645-648) */
                    all_ok = any;
                }
                owner_state[owner_cursor] = all_ok ? 1 : 2;
            }
            ++owner_cursor;
        }
        /* residue clouds = clusters of the pool under testOverlap, merged (661-683) */
        const int64_t nc = overlap_components(pool, n_pool, comp);
        for (int64_t c = 0; c < nc; ++c) {
            int64_t k = 0;
            for (int64_t i = 0; i < n_pool; ++i) if (comp[i] == c) members[k++] = i;
            ora_cloud1 *rc = &dom_pool[n_dom_pool];
            if (cloud_union(m, pool, members, k, rc)) return -1;
            dom_res[n_dom_pool++] = s->residue[a0];
            double el = 0;
            for (int64_t j = 0; j < rc->n_atoms; ++j) el += s->weight[rc->atoms[j]];
            if (el >= min_cloud_electrons) {
                if (n_res_rows < cap_rows) {
                    res_residue[n_res_rows] = s->residue[a0]; res_total[n_res_rows] = rc->total; res_n[n_res_rows] = rc->n; res_electrons[n_res_rows] = el;
                    for (int q = 0; q < 3; ++q) res_centroid[3 * n_res_rows + q] = rc->centroid[q];
                }
                ++n_res_rows;
            }
        }
        for (int64_t i = a0; i < a1; ++i) { first_of_key[s->key[i]] = -1; count_of_key[s->key[i]] = 0; }
        a0 = a1;
    }
    /* domain clouds (692-712) and the totals (714-726) */
    ora_cloud1 **dpool = (ora_cloud1 **)calloc((size_t)(n_dom_pool > 0 ? n_dom_pool : 1), sizeof(ora_cloud1 *));
    int64_t *dcomp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_dom_pool > 0 ? n_dom_pool : 1));
    if (!dpool || !dcomp) return -1;
    for (int64_t i = 0; i < n_dom_pool; ++i) dpool[i] = &dom_pool[i];
    const int64_t nd = overlap_components(dpool, n_dom_pool, dcomp);
    int64_t n_dom_rows = 0;
    double num_voxels = 0, total_electrons = 0, total_density = 0;
    for (int64_t c = 0; c < nd; ++c) {
        int64_t k = 0;
        for (int64_t i = 0; i < n_dom_pool; ++i) if (dcomp[i] == c) members[k++] = i;
        ora_cloud1 dc;
        memset(&dc, 0, sizeof dc);
        if (cloud_union(m, dpool, members, k, &dc)) return -1;
        double el = 0;
        for (int64_t j = 0; j < dc.n_atoms; ++j) el += s->weight[dc.atoms[j]];
        total_electrons += el; num_voxels += (double)dc.n; total_density += dc.total;
        if (el >= min_cloud_electrons) {
            if (n_dom_rows < cap_rows) {
                dom_residue[n_dom_rows] = dom_res[members[0]]; dom_total[n_dom_rows] = dc.total; dom_n[n_dom_rows] = dc.n; dom_electrons[n_dom_rows] = el;
                for (int q = 0; q < 3; ++q) dom_centroid[3 * n_dom_rows + q] = dc.centroid[q];
            }
            ++n_dom_rows;
        }
        cloud_free1(&dc);
    }
    counts[0] = n_atom_rows; counts[1] = n_res_rows; counts[2] = n_dom_rows;
    totals[0] = num_voxels; totals[1] = total_electrons; totals[2] = total_density;
    for (int64_t i = 0; i < n_dom_pool; ++i) cloud_free1(&dom_pool[i]);
    free(dom_pool); free(dom_res); free(pool); free(comp); free(members); free(first_of_key); free(count_of_key); free(dpool); free(dcomp);
    return (n_atom_rows > cap_rows || n_res_rows > cap_rows || n_dom_rows > cap_rows) ? -2 : 0;
}
