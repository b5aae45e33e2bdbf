/*
 * pdbeda.h -- C-ABI of libpdbeda_hip.so: the MI355X (gfx950) electron-density voxel core.
 *
 * This is the drop-in boundary for the one accelerated path of pdb_eda: everything the
 * reference routes through its `utils` seam (`from . import cutils as utils`,
 * /root/reference/pdb_eda/ccp4.py:16-19 and densityAnalysis.py:26-29).  A per-voxel
 * Python-tuple API cannot cross a device boundary, so the seam sits one level up:
 * ccp4.py hands over the raw [s][r][c] grid and the unit-cell basis once
 * (pdbeda_map_upload) and each entry point below replaces one reference call chain,
 * cited per function (paths relative to /root/reference/pdb_eda/).
 *
 * Conventions
 *   - plain C types only; all functions return 0 on success or a negative pdbeda_status;
 *     pdbeda_last_error(ctx) gives the message.  No exception crosses the ABI.
 *   - the caller owns every host buffer; the library owns device memory behind the
 *     opaque handles.  A context is bound to ONE device and ONE HIP stream; N contexts
 *     = N streams.  Calls on one context are not re-entrant; different contexts may be
 *     used from different threads concurrently.
 *   - crs triples are (column, row, section) = the reference's crsCoord order.
 *   - "cutoff"/"radius" parameters are C floats on purpose: the reference's Cython
 *     declares them `float` (cutils.pyx:28,185,205,220,250,273), i.e. the Python double
 *     is rounded to float32 before use.
 *   - there is no CPU fallback: without a usable gfx950 device every entry point fails
 *     with PDBEDA_ERR_DEVICE.
 */
#ifndef PDBEDA_H
#define PDBEDA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pdbeda_status {
    PDBEDA_OK = 0,
    PDBEDA_ERR_DEVICE = -1,   /* no device / HIP runtime error */
    PDBEDA_ERR_ARGUMENT = -2, /* bad argument */
    PDBEDA_ERR_MEMORY = -3,   /* device or host allocation failed */
    PDBEDA_ERR_CAPACITY = -4, /* caller buffer too small */
    PDBEDA_ERR_STATE = -5,    /* handle used in the wrong state */
    PDBEDA_ERR_TIMEOUT = -6   /* the per-entry watchdog expired; the context is abandoned (see pdbeda_ctx_set_timeout) */
} pdbeda_status;

typedef struct pdbeda_ctx pdbeda_ctx;           /* device + stream + reusable workspace */
typedef struct pdbeda_map pdbeda_map;           /* one density grid resident in HBM */
typedef struct pdbeda_bloblist pdbeda_bloblist; /* result of a labelling call */

/* The "unit-cell basis" ccp4.py computes on the host (DensityHeader, ccp4.py:225-286). */
typedef struct pdbeda_geometry {
    int32_t ncrs[3];         /* header.ncrs                                   ccp4.py:168   */
    int32_t crs_start[3];    /* header.crsStart                               ccp4.py:186   */
    int32_t xyz_interval[3]; /* header.xyzInterval                            ccp4.py:227   */
    int32_t map2xyz[3];      /* header.map2xyz                                ccp4.py:230-234 */
    int32_t map2crs[3];      /* header.map2crs                                ccp4.py:235   */
    int32_t orthogonal;      /* alpha == beta == gamma == 90                  ccp4.py:297   */
    double ortho[9];         /* header.orthoMat (row major)                   ccp4.py:248-250 */
    double deortho[9];       /* header.deOrthoMat (row major)                 ccp4.py:252-253 */
    double origin[3];        /* header.origin                                 ccp4.py:272-286 */
    double grid_len[3];      /* header.gridLength                             ccp4.py:228   */
    double unit_volume;      /* header.unitVolume                             ccp4.py:243-244 */
} pdbeda_geometry;

/* ---- library / context ------------------------------------------------------------ */
const char *pdbeda_version(void);
int pdbeda_device_count(void);
/* PCI address of a device ("0000:dc:00.0"): /sys/bus/pci/devices/<address>/local_cpulist names the host cores of the GPU's
 * NUMA node.  The reference's multiprocessing pool (multipleStructures.py:167-168) leaves placement to the OS; a worker that
 * feeds a GPU from the other socket reads and uploads its maps across the socket link (measured: 4x slower per entry). */
int pdbeda_device_pci_address(int device_id, char *out, int out_len);
int pdbeda_ctx_create(int device_id, pdbeda_ctx **out);
/* Bind to an existing hipStream_t (e.g. torch's current stream); stream == NULL -> new stream. */
int pdbeda_ctx_create_on_stream(int device_id, void *hip_stream, pdbeda_ctx **out);
int pdbeda_ctx_destroy(pdbeda_ctx *ctx);
int pdbeda_ctx_synchronize(pdbeda_ctx *ctx);
void *pdbeda_ctx_stream(pdbeda_ctx *ctx); /* the hipStream_t the kernels are launched on */
/* Per-entry watchdog: the reference wraps every entry of `pdb_eda multiple` in a SIGALRM time-out
 * (multipleStructures.py:297-304, 359-377), which threads cannot use.  With seconds > 0 every wait of this context on its
 * stream is a timed hipStreamQuery loop against ONE deadline, now + seconds, shared by all waits until the next call of this
 * function: the caller re-arms it when an entry starts (an entry makes dozens of waits; a clock per wait would let it run for
 * many multiples of the time-out).  When the deadline passes the call returns PDBEDA_ERR_TIMEOUT, the context is marked
 * abandoned (every later call on it or its handles fails at once with the same status) and pdbeda_ctx_destroy does not wait
 * for the stream: the context is parked and its device memory -- pool, buffers, and the arenas of the maps / jobs the entry
 * left behind -- is released by pdbeda_reap_abandoned once the stream has drained.  seconds == 0 disarms the watchdog (plain
 * hipStreamSynchronize).  Host-side phases of an entry are outside the library: the caller checks its own clock. */
int pdbeda_ctx_set_timeout(pdbeda_ctx *ctx, double seconds);
/* Destroy the abandoned contexts of this process whose streams have drained (hipStreamQuery, no waiting); returns how many are
 * still parked.  Called by the library itself at every context creation and before an allocation is reported as failed (then
 * the parked arenas of the sibling contexts on the device go back to the driver too); exposed for callers and tests. */
int64_t pdbeda_reap_abandoned(void);
const char *pdbeda_last_error(pdbeda_ctx *ctx);
/* Per-kernel timing with HIP events recorded on the context's stream (measurement aid for
 * bench.py; no reference counterpart).  profile_end synchronises and writes one
 * "kernel_name calls total_ms" line per kernel into buf. */
int pdbeda_ctx_profile_begin(pdbeda_ctx *ctx);
int pdbeda_ctx_profile_end(pdbeda_ctx *ctx, char *buf, int64_t cap);

/* ---- map residency: replaces DensityMatrix.__init__ (ccp4.py:322-341) ------------- */
/* density: host float32 [ns][nr][nc] (c fastest); copied to HBM. */
int pdbeda_map_upload(pdbeda_ctx *ctx, const float *density, const pdbeda_geometry *geom, pdbeda_map **out);
/* ... with the map's mean / std from the same wait (see pdbeda_map_upload_file_stats); mean / std may be NULL. */
int pdbeda_map_upload_stats(pdbeda_ctx *ctx, const float *density, const pdbeda_geometry *geom, pdbeda_map **out, double *mean, double *std);
/* density_dev: a device pointer the caller keeps alive (zero-copy, e.g. a torch tensor). */
int pdbeda_map_from_device(pdbeda_ctx *ctx, const float *density_dev, const pdbeda_geometry *geom, pdbeda_map **out);
/* The caller has rewritten a borrowed buffer in place: drop what the library cached about the map's contents (the quantum of
 * the order-independent blob sums, derived once per map from sum |rho| and max |rho|).  The reference has no counterpart: its
 * DensityMatrix owns its array (ccp4.py:322-341).  A map that holds a NaN or an infinity is refused by the labelling calls. */
int pdbeda_map_invalidate(pdbeda_map *map);
/* The grid of a CCP4 FILE straight into HBM (ccp4.read -> parse, ccp4.py:58-127): n = ncrs[0]*ncrs[1]*ncrs[2] float32 values
 * starting at byte `offset` (1024 + the symmetry records) of `path`, read through the process's upload engine (three pread() readers with
 * pinned chunks of their own, their PCIe copies queued on three streams; the context's stream is ordered behind them); byteswap != 0 when the
 * file has the other endianness (swapped on the device).  No host copy
 * of the map exists afterwards (pdbeda_map_download fetches one on demand). */
int pdbeda_map_upload_file(pdbeda_ctx *ctx, const char *path, int64_t offset, int byteswap, const pdbeda_geometry *geom, pdbeda_map **out);
/* The same with the map's mean and standard deviation (pdbeda_map_stats: DensityMatrix.meanDensity / stdDensity, ccp4.py:343-361)
 * computed behind the copies and returned from the SAME wait: every cutoff of the analysis is mean + k std, so the statistics are
 * what a caller asks for next.  mean / std may be NULL. */
int pdbeda_map_upload_file_stats(pdbeda_ctx *ctx, const char *path, int64_t offset, int byteswap, const pdbeda_geometry *geom, pdbeda_map **out,
                                 double *mean, double *std);
int pdbeda_map_free(pdbeda_map *map);
/* A new map on the geometry of `a` with density  float32( double(a) + alpha * double(b) )  per voxel -- the Fc map of
 * DensityAnalysis.fc is (2Fo-Fc) - 2 (Fo-Fc), densityAnalysis.py:426-435 (alpha = -2).  Same grid shape required. */
int pdbeda_map_combine(pdbeda_map *a, pdbeda_map *b, double alpha, pdbeda_map **out);
/* One digit of a radix select over the voxels of the unique box with |a| < cut_a and |a + alpha b| < cut_b (b may be NULL:
 * then only the first test applies and which must be 0): the 65536-bin histogram of bits [shift, shift+16) of the key of the
 * voxels whose key agrees with `prefix` on `prefix_mask`.  key = bits of float |a| (which = 0, 32 bits) or of double
 * |a + alpha b| (which = 1, 64 bits) -- both orders are the numeric orders.  Host code walks the digits to an exact order
 * statistic: DensityAnalysis.medianAbsFoFc (densityAnalysis.py:783-801) without moving the maps.  hist: 65536 uint32. */
int pdbeda_abs_select_hist(pdbeda_map *a, pdbeda_map *b, double alpha, double cut_a, double cut_b, int which, int shift,
                           unsigned long long prefix, unsigned long long prefix_mask, uint32_t *hist);
/* Copy the float32 grid [ns][nr][nc] back to the host (inspection; DensityMatrix.density of a derived map). */
int pdbeda_map_download(pdbeda_map *map, float *density_out);

/* ---- whole-map reductions --------------------------------------------------------- */
/* DensityMatrix.meanDensity / stdDensity: np.mean / np.std (population) over ALL stored
 * voxels in fp64 (ccp4.py:343-363). */
int pdbeda_map_stats(pdbeda_map *map, double *mean, double *std);
/* utils.sumOfAbs via getTotalAbsDensity: sum |v| for |v| > cutoff, strict (cutils.pyx:28-39,
 * ccp4.py:365-376). */
int pdbeda_sum_of_abs(pdbeda_map *map, float cutoff, double *out);

/* ---- point / geometry helpers (batched) ------------------------------------------- */
/* utils.getPointDensityFromCrs (cutils.pyx:125-145): periodic wrap, 0 outside the data. */
int pdbeda_point_density(pdbeda_map *map, const int32_t *crs, int64_t n, double *out);
/* utils.testValidCrs (cutils.pyx:147-167). */
int pdbeda_valid_crs(pdbeda_map *map, const int32_t *crs, int64_t n, uint8_t *out);
/* DensityHeader.crs2xyzCoord / xyz2crsCoord evaluated by the DEVICE code (ccp4.py:288-316);
 * exposed so the parity tests can pin the kernels' geometry arithmetic. */
int pdbeda_crs2xyz(pdbeda_map *map, const int32_t *crs, int64_t n, double *xyz);
int pdbeda_xyz2crs(pdbeda_map *map, const double *xyz, int64_t n, int32_t *crs);

/* ---- blob labelling --------------------------------------------------------------- */
#define PDBEDA_FLAG_LABELS 1u /* also materialise the dense int32 label volume in HBM */

/* DensityMatrix.createFullBlobList(cutoff) = utils.createFullCrsList + utils.createCrsLists
 * + DensityBlob.fromCrsList (ccp4.py:463-485, 522-545; cutils.pyx:185-203, 44-70):
 * inclusive threshold over the non-repeating box header.uniqueNcrs, 26-connected
 * components (non-periodic), per-blob fp64 statistics, blobs ordered by the c-major
 * position of their first voxel (the reference's emission order).  Asynchronous: returns
 * after enqueueing; the first accessor synchronises.  cutoff == 0 -> PDBEDA_ERR_ARGUMENT
 * (the reference returns None). */
int pdbeda_full_blobs(pdbeda_map *map, float cutoff, uint32_t flags, pdbeda_bloblist **out);
/* Fused green/red: ONE pass over the grid labels density >= cutoff_pos and
 * density <= cutoff_neg (densityAnalysis.py:392-412 calls the above twice). */
int pdbeda_full_blobs_pm(pdbeda_map *map, float cutoff_pos, float cutoff_neg, uint32_t flags,
                         pdbeda_bloblist **green, pdbeda_bloblist **red);

/* DensityMatrix.findAberrantBlobs (ccp4.py:437-461) = utils.getSphereCrsFromXyz[List]
 * (cutils.pyx:220-271) + createBlobList, batched: atoms [group_offsets[g], group_offsets[g+1])
 * form group g whose sphere voxels are unioned on RAW crs (a one-atom group is the
 * single-coordinate call).  xyz: n_atoms x 3 doubles (float32 atom coordinates promoted
 * exactly); radii: per atom.  Blobs come back sorted by (group, c-major first voxel of the
 * group's bounding box). */
int pdbeda_sphere_blobs(pdbeda_map *map, const double *xyz, const float *radii, int64_t n_atoms,
                        const int64_t *group_offsets, int64_t n_groups, float density_cutoff,
                        pdbeda_bloblist **out);

/* DensityMatrix.createBlobList(crsList) (ccp4.py:475-485) on explicit raw voxel sets:
 * voxels [group_offsets[g], group_offsets[g+1]) of crs (n x 3) form group g (duplicates
 * collapse, as in DensityBlob's set).  Used for DensityBlob.merge / fromCrsList and for
 * the residue / domain cloud unions of aggregateCloud (densityAnalysis.py:646-708). */
int pdbeda_list_blobs(pdbeda_map *map, const int32_t *crs, int64_t n, const int64_t *group_offsets,
                      int64_t n_groups, pdbeda_bloblist **out);

/* ---- blob list accessors ---------------------------------------------------------- */
int64_t pdbeda_bloblist_count(pdbeda_bloblist *bl);      /* number of blobs (synchronises) */
int64_t pdbeda_bloblist_num_voxels(pdbeda_bloblist *bl); /* total voxels in all blobs */
/* Per blob, any pointer may be NULL: n voxels, totalDensity, centroid[3], coordCenter[3],
 * volume (= unitVolume * n), first_key (c-major position of the first voxel), group. */
int pdbeda_bloblist_stats(pdbeda_bloblist *bl, int64_t *n, double *total_density, double *centroid,
                          double *coord_center, double *volume, int64_t *first_key, int32_t *group);
/* Voxel membership: crs (N x 3, raw coordinates) grouped by blob in blob order;
 * blob_offsets has count+1 entries. */
int pdbeda_bloblist_voxels(pdbeda_bloblist *bl, int32_t *crs, int64_t *blob_offsets);
/* Dense labels of a full-map list: int32 [us][ur][uc] over header.uniqueNcrs, blob index or
 * -1.  Computed on first use unless PDBEDA_FLAG_LABELS was given. */
int pdbeda_bloblist_labels(pdbeda_bloblist *bl, int32_t *labels_host);
int pdbeda_bloblist_free(pdbeda_bloblist *bl);
/* Diagnostic (no reference counterpart): counters of the labelling job behind a list,
 * out[8] = run ids, component ids, runs of the job beyond the first (1 = the typical-size arena was too small for this map and
 * the job ran again in a worst-case one), blobs, tiles off the fast path by kind (unit tiles: run slots full; wide tiles: more
 * components than the LDS tables hold, united in LDS all the same; unit tiles: no ids left for a wide tile), bytes of device memory
 * the job holds. */
int pdbeda_bloblist_counters(pdbeda_bloblist *bl, int64_t *out);

/* ---- regional sums ---------------------------------------------------------------- */
/* The voxel part of calculateRegionDiscrepancy / calculateRegionDensity
 * (densityAnalysis.py:1037-1068, 1160-1211): per group (atom or residue sphere-union,
 * deduplicated on raw crs): pos = sum of density > cutoff, neg = sum of density < -cutoff
 * (strict, cutils.pyx:245), n_region = |sphere union| with no density filter
 * (densityAnalysis.py:1198), valid = utils.testValidXyzList (cutils.pyx:273-313).
 * Any output pointer may be NULL. */
int pdbeda_region_sums(pdbeda_map *map, const double *xyz, const float *radii, int64_t n_atoms,
                       const int64_t *group_offsets, int64_t n_groups, float cutoff,
                       double *pos, double *neg, int64_t *n_region, uint8_t *valid);

/* ---- aggregateCloud ------------------------------------------------------------------ */
/* DensityAnalysis.aggregateCloud up to its statistics tail (densityAnalysis.py:571-731) as ONE call: the clouds of every
 * eligible atom (findAberrantBlobs, 603), the centroid-distance cut-off over all atoms (607), the best cloud and the pooled
 * clouds per atom (622-642), the bonded-atom overlap completeness (652-659), the residue clouds = clusters of a residue's
 * pooled clouds under testOverlap, merged (644-650, 661-690), the domain clouds = the same over everything pooled
 * (692-712) and the totals behind densityElectronRatio (714-731).  Voxel lists never leave the device.
 *
 * The caller flattens what the reference reads from the structure, in the reference's iteration order
 * (residues with id[0] == ' ', their child atoms whose residue_atom name has an atom type and whose occupancy != 0): */
typedef struct pdbeda_cloud_atoms {
    int64_t n;                 /* eligible atoms */
    const double *xyz;         /* n x 3: atom.coord (float32 promoted exactly) */
    const float *radius;       /* n: radii[atom type]                                         densityAnalysis.py:603 */
    const double *weight;      /* n: electrons[residue_atom] * occupancy                      densityAnalysis.py:690, 718 */
    const int32_t *residue;    /* n: ordinal of the atom's residue, non-decreasing */
    const int32_t *alias;      /* n: LAST eligible atom with the same coordinate (allAtomClouds is keyed by coordinate, 604) */
    const int32_t *key;        /* n: id of (residue, residue_atom name); atomCloudIndeces is keyed by the name (640) */
    int64_t n_keys;
    const int64_t *bonded_off; /* n_keys + 1: CSR over keys ... */
    const int32_t *bonded;     /* ... of the keys of bondedAtoms[name] that exist in the same residue      (656) */
    int64_t n_owners;          /* child atoms (ANY occupancy) of ATOM residues whose name has a key, in iteration order (653-655) */
    const int32_t *owner_key;
} pdbeda_cloud_atoms;

typedef struct pdbeda_cloud pdbeda_cloud; /* the result tables (host memory owned by the library) */

int pdbeda_aggregate_cloud(pdbeda_map *map, const pdbeda_cloud_atoms *atoms, float density_cutoff, double min_cloud_electrons,
                           pdbeda_cloud **out);
/* counts[4] = atom rows, residue-cloud rows, domain-cloud rows, owners;
 * totals[4] = numVoxels, totalElectrons, totalDensity (718-721, over ALL domain clouds), centroidDistanceCutoff (607). */
int pdbeda_cloud_counts(pdbeda_cloud *c, int64_t counts[4], double totals[4]);
/* One row per atom that has a best cloud, in iteration order (atomList, 642): index into the eligible atoms, the best
 * cloud's totalDensity, voxel count, centroid (x3) and |atom - centroid|.  Any pointer may be NULL. */
int pdbeda_cloud_atom_rows(pdbeda_cloud *c, int32_t *atom, double *total_density, int64_t *n_voxels, double *centroid, double *distance);
/* Residue clouds with >= min_cloud_electrons, in the reference's emission order (residue by residue; inside a residue by
 * lowest pooled-cloud index -- the reference's own order there follows CPython's set iteration): residue ordinal,
 * totalDensity, voxels, electrons, centroid (x3). */
int pdbeda_cloud_residue_rows(pdbeda_cloud *c, int32_t *residue, double *total_density, int64_t *n_voxels, double *electrons, double *centroid);
/* Domain clouds with >= min_cloud_electrons (unsorted: the caller sorts by ratio, 730): a representative residue ordinal
 * (the reference's is set-order dependent), totalDensity, voxels, electrons, centroid (x3). */
int pdbeda_cloud_domain_rows(pdbeda_cloud *c, int32_t *residue, double *total_density, int64_t *n_voxels, double *electrons, double *centroid);
/* Per owner: 0 = its name has no pooled cloud, 1 = every bonded partner's clouds touch its own, 2 = some do not (656-659). */
int pdbeda_cloud_owner_states(pdbeda_cloud *c, uint8_t *state);
int pdbeda_cloud_free(pdbeda_cloud *c);

/* ---- voxel-set adjacency ---------------------------------------------------------- */
/* utils.testOverlap (cutils.pyx:8-25) batched: pair p tests set a_idx[p] against set
 * b_idx[p]; sets are slices [set_offsets[i], set_offsets[i+1]) of crs. */
int pdbeda_test_overlap(pdbeda_ctx *ctx, const int32_t *crs, const int64_t *set_offsets, int64_t n_sets,
                        const int32_t *a_idx, const int32_t *b_idx, int64_t n_pairs, uint8_t *out);

/* ---- symmetry atoms --------------------------------------------------------------- */
/* utils.createSymmetryAtoms (cutils.pyx:73-103): x' = R x + t + orthoMat (i,j,k) for
 * (i,j,k) in {-1,0,1}^3 x ops, kept when inside bbox +- 5 A; the identity keeps all.
 * Outputs in the reference's order; returns the count through n_out (capacity cap). */
int pdbeda_symmetry_atoms(pdbeda_ctx *ctx, const double *xyz, int64_t n_atoms, const double *rot /* n_ops x 12 */,
                          int32_t n_ops, const double ortho[9], const double bbox_lo[3], const double bbox_hi[3],
                          int32_t *atom_index, int32_t *symmetry /* x4 */, double *out_xyz, int64_t cap, int64_t *n_out);
/* calculateAtomSpecificBlobStatistics inner step (densityAnalysis.py:932-935): for every
 * centroid the nearest atom (first index on ties, fp64 Euclidean as scipy cdist). */
int pdbeda_nearest_atom(pdbeda_ctx *ctx, const double *centroids, int64_t n_centroids, const double *atom_xyz,
                        int64_t n_atoms, int64_t *index, double *distance);

#ifdef __cplusplus
}
#endif
#endif /* PDBEDA_H */
