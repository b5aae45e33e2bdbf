// pdbeda_device.h -- device-side geometry, wrap rule and bit helpers shared by all kernels.
// gfx950 only.  The whole library is compiled with -ffp-contract=off: the reference's
// Python arithmetic is unfused (mul then add, two roundings) and voxel membership depends
// on it bit for bit (SURVEY.md 8a "hard parts").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pdbeda.h"

namespace pdbeda {

// Device copy of pdbeda_geometry + derived integer tables (ccp4.py:237, 262-269).
struct Geom {
    int32_t ncrs[3];
    int32_t crs_start[3];
    int32_t xyz_interval[3];
    int32_t map2xyz[3];
    int32_t map2crs[3];
    int32_t crs_interval[3];
    int32_t unique_ncrs[3];
    int32_t orthogonal;
    double ortho[9];
    double deortho[9];
    double origin[3];
    double grid_len[3];
    double unit_volume;
};

__host__ __device__ inline int32_t sel3(const int32_t v[3], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }

// Python's floor division for the wrap rule: crs -= floor(crs / interval) * interval
// (cutils.pyx:139-140).
__host__ __device__ inline int32_t floor_mod(int32_t v, int32_t m) {
    int32_t r = v % m;
    return (r != 0 && ((r < 0) != (m < 0))) ? r + m : r;
}

// cutils.pyx:125-145 / 147-167: wrap one axis; returns -1 when the voxel is not stored.
__host__ __device__ inline int32_t wrap_axis(int32_t v, int32_t n, int32_t interval) {
    if (v < 0 || v >= n) v = floor_mod(v, interval);
    if ((n <= v && v < interval) || v < 0) return -1;
    return v;
}

// getPointDensityFromCrs (cutils.pyx:125-145); *valid = testValidCrs (cutils.pyx:147-167).
__device__ inline float fetch_wrapped(const Geom &g, const float *__restrict__ density, int32_t c, int32_t r, int32_t s,
                                      bool *valid = nullptr) {
    int32_t wc = wrap_axis(c, g.ncrs[0], g.crs_interval[0]);
    int32_t wr = wrap_axis(r, g.ncrs[1], g.crs_interval[1]);
    int32_t ws = wrap_axis(s, g.ncrs[2], g.crs_interval[2]);
    bool ok = (wc >= 0) && (wr >= 0) && (ws >= 0);
    if (valid) *valid = ok;
    if (!ok) return 0.0f;
    return density[((int64_t)ws * g.ncrs[1] + wr) * g.ncrs[0] + wc];
}

// np.dot(3x3 float64, 3) as the reference's numpy/OpenBLAS evaluates it (identified bit-exactly
// from the golden vectors): fma(a2, v2, fma(a0, v0, a1 * v1)).  Explicit fma() calls survive
// -ffp-contract=off.  BLAS-kernel specific.
__host__ __device__ inline void matvec3(const double a[9], const double v[3], double out[3]) {
    for (int i = 0; i < 3; ++i) {
        double p1 = a[3 * i + 1] * v[1];
        out[i] = fma(a[3 * i + 2], v[2], fma(a[3 * i + 0], v[0], p1));
    }
}

// DensityHeader.crs2xyzCoord (ccp4.py:304-316) for integer crs.
__host__ __device__ inline void crs2xyz(const Geom &g, int32_t c, int32_t r, int32_t s, double xyz[3]) {
    const int32_t crs[3] = {c, r, s};
    if (g.orthogonal) {
        for (int i = 0; i < 3; ++i) {
            double p = (double)sel3(crs, g.map2xyz[i]) * g.grid_len[i];
            xyz[i] = p + g.origin[i];
        }
    } else {
        double f[3];
        for (int i = 0; i < 3; ++i) {
            int a = g.map2xyz[i];
            f[i] = (double)((int64_t)sel3(crs, a) + sel3(g.crs_start, a)) / (double)g.xyz_interval[i];
        }
        matvec3(g.ortho, f, xyz);
    }
}

// The same affine map on fractional crs (blob centroids: sum(rho*xyz)/sum(rho) is
// evaluated as xyz(sum(rho*crs)/sum(rho)); exact in real arithmetic, ~1e-15 relative in fp64).
__host__ __device__ inline void crs2xyz_frac(const Geom &g, const double crs[3], double xyz[3]) {
    if (g.orthogonal) {
        for (int i = 0; i < 3; ++i) {
            int a = g.map2xyz[i];
            double v = a == 0 ? crs[0] : (a == 1 ? crs[1] : crs[2]);
            xyz[i] = v * g.grid_len[i] + g.origin[i];
        }
    } else {
        double f[3];
        for (int i = 0; i < 3; ++i) {
            int a = g.map2xyz[i];
            double v = a == 0 ? crs[0] : (a == 1 ? crs[1] : crs[2]);
            f[i] = (v + (double)sel3(g.crs_start, a)) / (double)g.xyz_interval[i];
        }
        matvec3(g.ortho, f, xyz);
    }
}

// DensityHeader.xyz2crsCoord (ccp4.py:288-302); Python round() = half-to-even = rint.
__host__ __device__ inline void xyz2crs(const Geom &g, const double xyz[3], int32_t crs[3]) {
    int64_t grid[3];
    if (g.orthogonal) {
        for (int i = 0; i < 3; ++i) {
            double d = xyz[i] - g.origin[i];
            grid[i] = (int64_t)rint(d / g.grid_len[i]);
        }
    } else {
        double f[3];
        matvec3(g.deortho, xyz, f);
        for (int i = 0; i < 3; ++i) {
            double p = f[i] * (double)g.xyz_interval[i];
            grid[i] = (int64_t)rint(p) - sel3(g.crs_start, g.map2xyz[i]);
        }
    }
    for (int i = 0; i < 3; ++i) {
        int a = g.map2crs[i];
        crs[i] = (int32_t)(a == 0 ? grid[0] : (a == 1 ? grid[1] : grid[2]));
    }
}

// ---- 64-bit word helpers (a mask word = 64 consecutive voxels along c) -------------
__host__ __device__ inline int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
__host__ __device__ inline int ctz64(uint64_t x) { // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
__host__ __device__ inline int clz64(uint64_t x) { // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}
__host__ __device__ inline uint64_t bits_below(int p) { return p >= 64 ? ~0ull : ((1ull << p) - 1ull); } // bits [0,p)
__host__ __device__ inline uint64_t run_starts(uint64_t m) { return m & ~(m << 1); }
// first bit of the run (within this word) that contains set bit p
__host__ __device__ inline int run_start_of(uint64_t m, int p) {
    uint64_t z = ~m & bits_below(p);
    return z ? 64 - clz64(z) : 0;
}
// ordinal (0-based, inside the word) of the run that contains set bit p: the run's own start and the
// starts of all earlier runs are exactly the starts at positions <= p.  32-bit halves only.
__host__ __device__ inline uint32_t run_ordinal(uint64_t starts, int p) {
    const uint32_t lo = (uint32_t)starts, hi = (uint32_t)(starts >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t ml = p >= 31 ? 0xffffffffu : ((2u << p) - 1u);
    const uint32_t mh = p < 32 ? 0u : (p >= 63 ? 0xffffffffu : ((2u << (p - 32)) - 1u));
    return (uint32_t)__popc(lo & ml) + (uint32_t)__popc(hi & mh) - 1u;
#else
    const uint32_t ml = p >= 31 ? 0xffffffffu : ((2u << p) - 1u);
    const uint32_t mh = p < 32 ? 0u : (p >= 63 ? 0xffffffffu : ((2u << (p - 32)) - 1u));
    return (uint32_t)__builtin_popcount(lo & ml) + (uint32_t)__builtin_popcount(hi & mh) - 1u;
#endif
}
// last bit of the run (within this word) that starts at bit a
__host__ __device__ inline int run_end_of(uint64_t m, int a) {
    uint64_t inv = ~(m >> a);
    int len = inv ? ctz64(inv) : 64;
    if (len > 64 - a) len = 64 - a;
    return a + len - 1;
}

}  // namespace pdbeda
