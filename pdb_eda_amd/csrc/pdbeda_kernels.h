// pdbeda_kernels.h -- the HIP kernels of the voxel core (gfx950 / CDNA4, wave64).
//
// Data model ("volume batch"): every labelling job works on a batch of bit-mask volumes.
// A volume is a box of voxels in RAW crs coordinates (org + local); one mask bit per voxel,
// 64 consecutive voxels along c per 64-bit word (one wave64 ballot), rows padded to whole
// words.  The whole-map job has one volume per sign (the non-repeating box
// header.uniqueNcrs); a findAberrantBlobs batch has one small volume per atom / residue.
//
// Labelling is run based: the union-find element is a *run* (maximal set of consecutive
// set bits inside one word).  Runs of a word are found with bit tricks, indexed compactly
// (runBase[word] + popcount), carry their own fp64 partial sums, and are united with the
// runs of the 4 "earlier" neighbour rows + the previous word (26-connectivity,
// non-periodic, cutils.pyx:44-70).
#pragma once
#include "pdbeda_device.h"

namespace pdbeda {

struct VolDesc {
    int32_t dim[3];     // voxels along c, r, s
    int32_t org[3];     // raw crs of local voxel (0,0,0)
    int32_t row_words;  // ceil(dim[0] / 64)
    int32_t group;      // caller-visible group id (plane for whole-map jobs)
    int64_t word_base;  // first mask word of this volume
    int64_t key_base;   // first key bit of this volume (key = (c*dim[1] + r)*dim[2] + s)
};

// ---- Order-independent sums --------------------------------------------------------------------------------------
// sum(rho) and the first moments sum(rho * c / r / s) of a blob are folded from thousands of partial sums by atomics whose
// order the hardware picks; in floating point that makes the last bits differ from run to run, and aggregateCloud DECIDES
// things with these sums (best cloud, centroid-distance cut-off, pooling).  So they are kept as integers: a partial sum is
// rounded ONCE to a multiple of 2^-S (fix_of; S is chosen per map from sum |rho| and max |rho|, so that no sum can overflow
// and the quantum is ~1e-12 of a significant voxel), and integer addition commutes.  A moment is two limbs,
// value = hi * 2^32 + lo: sums of limbs over a whole map stay far inside 64 bits.
struct FixSums {
    long long rho, c_lo, c_hi, r_lo, r_hi, s_lo, s_hi;
};
__device__ inline long long fix_of(double v, double mul) { return __double2ll_rn(v * mul); }
__device__ inline void fix_limbs(__int128 v, long long &lo, long long &hi) {
    lo = (long long)(unsigned long long)((unsigned __int128)v & 0xffffffffu);
    hi = (long long)(v >> 32);
}
__device__ inline double fix_moment(long long lo, long long hi) { return (double)hi * 4294967296.0 + (double)lo; }
// rho = F and moments F * (c, r, s) + (m_c, m_r, m_s): coordinates of the frame's origin, moments relative to it
__device__ inline FixSums fix_sums(long long F, long long m_c, long long m_r, long long m_s, long long c0, long long r0, long long s0) {
    FixSums o;
    o.rho = F;
    fix_limbs((__int128)F * c0 + m_c, o.c_lo, o.c_hi);
    fix_limbs((__int128)F * r0 + m_r, o.r_lo, o.r_hi);
    fix_limbs((__int128)F * s0 + m_s, o.s_lo, o.s_hi);
    return o;
}
__device__ inline void fix_add(FixSums &a, const FixSums &b) {
    a.rho += b.rho; a.c_lo += b.c_lo; a.c_hi += b.c_hi; a.r_lo += b.r_lo; a.r_hi += b.r_hi; a.s_lo += b.s_lo; a.s_hi += b.s_hi;
}
__device__ inline void fix_atomic_add(FixSums *dst, const FixSums &v) {   // (global or LDS)
    atomicAdd((unsigned long long *)&dst->rho, (unsigned long long)v.rho);
    atomicAdd((unsigned long long *)&dst->c_lo, (unsigned long long)v.c_lo);
    atomicAdd((unsigned long long *)&dst->c_hi, (unsigned long long)v.c_hi);
    atomicAdd((unsigned long long *)&dst->r_lo, (unsigned long long)v.r_lo);
    atomicAdd((unsigned long long *)&dst->r_hi, (unsigned long long)v.r_hi);
    atomicAdd((unsigned long long *)&dst->s_lo, (unsigned long long)v.s_lo);
    atomicAdd((unsigned long long *)&dst->s_hi, (unsigned long long)v.s_hi);
}
template <typename JobRef> __device__ inline FixSums fix_load(const JobRef &job, uint32_t i) {
    const long long *p = job.r_sum + i;
    const int64_t s = job.r_sum_stride;
    FixSums o;
    o.rho = p[0]; o.c_lo = p[s]; o.c_hi = p[2 * s]; o.r_lo = p[3 * s]; o.r_hi = p[4 * s]; o.s_lo = p[5 * s]; o.s_hi = p[6 * s];
    return o;
}
template <typename JobRef> __device__ inline void fix_store(const JobRef &job, uint32_t i, const FixSums &v) {
    long long *p = job.r_sum + i;
    const int64_t s = job.r_sum_stride;
    p[0] = v.rho; p[s] = v.c_lo; p[2 * s] = v.c_hi; p[3 * s] = v.r_lo; p[4 * s] = v.r_hi; p[5 * s] = v.s_lo; p[6 * s] = v.s_hi;
}
template <typename JobRef> __device__ inline void fix_fold(const JobRef &job, uint32_t i, const FixSums &v) {   // atomically, into record i
    unsigned long long *p = (unsigned long long *)(job.r_sum + i);
    const int64_t s = job.r_sum_stride;
    atomicAdd(p, (unsigned long long)v.rho);
    atomicAdd(p + s, (unsigned long long)v.c_lo); atomicAdd(p + 2 * s, (unsigned long long)v.c_hi);
    atomicAdd(p + 3 * s, (unsigned long long)v.r_lo); atomicAdd(p + 4 * s, (unsigned long long)v.r_hi);
    atomicAdd(p + 5 * s, (unsigned long long)v.s_lo); atomicAdd(p + 6 * s, (unsigned long long)v.s_hi);
}
__device__ inline FixSums fix_zero() { FixSums z; z.rho = z.c_lo = z.c_hi = z.r_lo = z.r_hi = z.s_lo = z.s_hi = 0; return z; }

struct Counters {
    unsigned int n_runs;
    unsigned int n_comps;
    unsigned int n_blobs;
    unsigned int n_blobs_vol0;   // blobs whose first key lies in volume 0 (the split of a fused green / red job's table)
    unsigned int unit_wait_failed;   // sphere batches: the device's volumes outgrew what the host sized the job for (k_make_vols)
    unsigned int unit_tiles[3];  // whole-map tiles that fell back to unit mode: run slots / edge buffer / component table full
    // whole-map jobs are carved for what maps typically need, not for the worst case (round 4): bit 0 = the unit tiles asked for
    // more run / component ids than the job has, bit 1 = more blobs than table rows, bit 2 = the job has unit tiles and was enqueued
    // without their two launches (Job::unit_form).  Every kernel stays inside the arena (the
    // unit work is skipped, rows beyond the table are dropped); the host sees the flag at its first read of the counters and
    // runs the job again in a worst-case arena.
    unsigned int overflow;
    unsigned long long n_voxels;
    long long total_words;
    long long total_keys;
};

struct Job {
    VolDesc *vols;
    int32_t n_vols;
    int64_t total_words;
    int64_t key_words;
    uint64_t *mask;
    uint32_t *run_base;
    uint64_t *key_bits;
    // rank of a key = set bits of key_bits below it.  k_paint_keys keeps one 16-bit counter per KEY_FINE key words beside the
    // bitmap (two per 32-bit word; spread over many cache lines: same-line atomics serialise); every block of k_emit turns them
    // into a prefix table of <= KEY_GROUPS entries in LDS (fine_per_group counters per entry, a multiple of 8) -- no scan kernels
    uint32_t *fine_count;
    int32_t n_fine, n_fine_alloc, fine_per_group;   // counters in use / allocated (whole groups) / per table entry
    Counters *ctr;
    // union-find elements ("components"): tile-local components on the whole-map fast path,
    // single runs (comp_of_run == nullptr, identity) on the generic path
    int32_t comps_are_runs;
    uint32_t *comp_of_run;
    int32_t *label_of_comp;   // final signed label of every component (whole-map jobs)
    // per tile, sign and mask word: the tile-local components of the word's first 7 word-runs and of the run at its last bit,
    // a byte each -- k_face_merge unites across tile faces from the mask words and these records alone (one round trip)
    uint8_t *word_comps;               // [tile][2][256][8]
    uint32_t *unit_done;               // (unused since round 5: was a per-tile flag the k_face_merge workgroups polled)
    uint32_t *unit_flag;      // [0] == epoch iff some tile of THIS job is a wide or a unit tile (stale values of a recycled arena never match); [1] == epoch once k_tile_label's workgroup 0 has initialised the counters; [2] == epoch iff some tile is a UNIT tile (modes 1 / 3: the two extra launches, Job::unit_form)
    uint32_t epoch;           // job number of the context (never 0)
    uint8_t *tile_mode;       // per tile: 0 = united in LDS; 1 / 3 = unit tile (too many runs / no ids for its components); 2 = wide tile (united in LDS, components above the tiles' id ranges)
    uint32_t *tile_runs;      // per tile: number of word-runs (ids tile * runs_per_tile ...)
    // Scattered global atomics are the scarce resource of the merge (~20 G/s chip-wide: folding 9 fields per (tile, root) pair
    // took 12 of k_resolve_tiles' 22 us).  So a tile's members POST their summed record to the inbox of the tile that owns
    // their root -- one returning atomic for the slot + plain stores -- and whoever writes the blob table rows (the fused
    // label writer, or k_emit_tiles) sums the tile's inbox in LDS first.  A full inbox falls back to the atomics.
    struct InboxEntry *inbox;     // [tile][INBOX_CAP]
    uint32_t *inbox_count;        // [tile * INBOX_STRIDE]: one counter per 128-B line (atomics on one line serialise); cleared by the
                                  // tile's own k_face_merge workgroup
    int32_t vol_sign[2];      // whole-map jobs: +1 / -1 list of volume p
    // per-component records
    int32_t *parent;
    uint32_t *r_n;
    // sum(rho) and its moments, as order-independent integers: field k of component i at r_sum[k * r_sum_stride + i] -- seven
    // arrays, not an array of records: the atomics that fold thousands of tile components into ONE root (a blob that spans the
    // map) then fall on seven cache lines instead of one, where they would serialise
    long long *r_sum;
    int64_t r_sum_stride;
    double fix_mul;           // 2^S of this job's map (see FixSums)
    long long *r_c, *r_r, *r_s;
    unsigned long long *r_key;
    uint32_t *r_rank;
    // blob table
    int64_t *b_n, *b_key;
    double *b_total, *b_centroid, *b_center, *b_volume;
    int32_t *b_group;
    // (last: the tile kernel's scalar-register allocation is sensitive to the offsets of the fields above -- inserting these
    //  in the middle cost it 10 more SGPR spills and 1.6 us)
    uint64_t *root_mask;      // per tile: which of its TILE_COMPS component slots are blob roots (4 ballots, written by k_resolve_tiles):
    int32_t n_tiles;          // k_emit_tiles visits the ~36 k roots of a 256^3 job, not its 262 k component ids
    // Whole-map jobs unite by FIRST KEY, not by id (round 4): kpar[x] = key32(parent) << 32 | parent, key32 = plane << 31 | the
    // c-major key of the component's first voxel inside its plane (unique: a position).  The root of a blob is then the
    // component that holds the blob's first voxel, its own key IS the blob's first key as soon as the unions are done, and the
    // kernel that used to fold min keys into the roots before the keys could be painted (k_paint_tiles) is gone.  The value read
    // from kpar[x] carries the parent's key AND its id: a find costs the same trips as on ids.  ~0 = unused id.
    unsigned long long *kpar;
    // a third level between the 16-bit counters (2048 keys) and the bitmap: one BYTE per 256 keys (4 bitmap words; two first
    // voxels are never neighbours along s, so a byte holds at most 128) -- rank_of_key then reads 32 + 8 + 32 bytes where it read
    // 32 + 256, few enough registers for the label writer to rank its own components (round 4)
    uint32_t *mid_count;
    // fine_per_group is a power of two (>= 16): a key's group is a shift, and the number of groups comes with the job -- the
    // integer divisions they replace ran in the prologue of every wave of the label writer
    int32_t fine_shift, n_groups;
    uint32_t run_cap, comp_cap, blob_cap;   // ids / table rows the arena holds (see Counters::overflow)
    // words of the first-key bitmap / 16-bit counter words / byte counter words every tile's k_face_merge workgroup clears (the
    // ceilings of the table sizes over the tile count: three 64-bit divisions per wave where the kernel computed them)
    int32_t clear_bits, clear_fine, clear_mid;
    // whole-map jobs of more than 2^25 keys (groups of more than 16 counters): the groups' totals, summed ONCE by k_group_counts
    // behind k_resolve_tiles.  Every workgroup that ranks builds its prefix table from these n_groups words; summing the 16-bit
    // counters itself it read the whole counter array -- 32 KiB at 256^3, but 256 KiB at 512^3, by each of 8 192 workgroups
    // (round 4: the fused label writer took 557 us at 512^3 for that reason, 3.3 x what eight 256^3 maps take).  nullptr otherwise.
    uint32_t *group_count;
    // Tiles beyond every LDS capacity ("unit tiles": more than 4 096 word-runs -- checkerboards) are labelled and united by TWO
    // launches of their own behind k_face_merge (k_unit_label, k_unit_pairs: the launch boundary is the only synchronisation).
    // A job is first enqueued WITHOUT them (unit_form 0: four launches, no cost for the maps that have no such tile); if a tile
    // turns out to be one, k_face_merge raises Counters::overflow bit 2, every later kernel leaves the unit work alone, and the
    // host runs the job again in this form.  (Rounds 3-4: the tiles' k_face_merge workgroups did both phases themselves and
    // polled each other's flags in between -- bounded, but a late dispatch on a shared GPU failed the job.)
    int32_t unit_form;
};

struct InboxEntry {           // 88 bytes: what a (tile, root) pair folds into the root's record (no key: the root holds the first voxel)
    uint32_t local, n;        // component index of the root inside its tile; voxels
    FixSums sum;
    unsigned long long c, r, s;
};
constexpr int INBOX_STRIDE = 32;   // uint32 per inbox counter: a cache line each
constexpr int INBOX_CAP = 192;   // entries per tile (a root tile of a map-spanning blob overflows: those pairs fold with atomics)

constexpr int TILE_COMPS = 256;   // component ids a whole-map tile owns (== CCAP of pdbeda_tile.h)
constexpr int KEY_FINE = 32;      // key words per fine counter (2048 keys)
constexpr int KEY_GROUPS = 1024;  // entries of the prefix table a k_emit block builds in LDS
constexpr int WAVE = 64;

__device__ inline int lane_id() { return threadIdx.x & 63; }

__device__ inline int find_vol(const VolDesc *vols, int n, int64_t word) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (vols[mid].word_base <= word) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ inline int find_vol_by_key(const VolDesc *vols, int n, int64_t key) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (vols[mid].key_base <= key) lo = mid; else hi = mid - 1;
    }
    return lo;
}


__device__ inline double wave_incl_scan(double x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    return x;
}

// One wave, one non-empty mask word `mw` (lane = voxel): wrapped density fetch
// (getPointDensityFromCrs), two wave prefix sums (rho, rho*lane), run sums by difference at the
// run's first lane; writes one record per run at index rec0 + (run number inside the word).
// (cl0, rl, sl) = volume-local coordinates of lane 0, (rawc0, rawr, raws) = raw crs of lane 0.
template <typename JobRef>
__device__ inline void word_run_records_of(const JobRef &job, const VolDesc &vd, uint64_t mw, int lane, double rho, int cl0, int rl, int sl,
                                           int rawc0, int rawr, int raws, uint32_t rec0);
template <typename JobRef>
__device__ inline void word_run_records(const JobRef &job, const Geom &g, const float *__restrict__ dens, const VolDesc &vd, uint64_t mw,
                                        int lane, int cl0, int rl, int sl, int rawc0, int rawr, int raws, uint32_t rec0) {
    const bool bit = (mw >> lane) & 1ull;
    const double rho = bit ? (double)fetch_wrapped(g, dens, rawc0 + lane, rawr, raws) : 0.0;
    word_run_records_of(job, vd, mw, lane, rho, cl0, rl, sl, rawc0, rawr, raws, rec0);
}
// ... the same with the lane's density already in hand (0 where the lane's bit is clear)
template <typename JobRef>
__device__ inline void word_run_records_of(const JobRef &job, const VolDesc &vd, uint64_t mw, int lane, double rho, int cl0, int rl, int sl,
                                           int rawc0, int rawr, int raws, uint32_t rec0) {
    const double rl_ = rho * (double)lane;
    const double p1 = wave_incl_scan(rho, lane);
    const double p2 = wave_incl_scan(rl_, lane);
    const uint64_t starts = run_starts(mw);
    const bool is_start = (starts >> lane) & 1ull;
    const int e = is_start ? run_end_of(mw, lane) : lane;
    const double p1e = __shfl(p1, e);
    const double p2e = __shfl(p2, e);
    if (is_start) {
        const int len = e - lane + 1;
        const double s_rho = p1e - (p1 - rho);
        const double s_rl = p2e - (p2 - rl_);
        const uint32_t idx = rec0 + (uint32_t)popc64(starts & bits_below(lane));
        job.parent[idx] = (int32_t)idx;
        job.r_n[idx] = (uint32_t)len;
        // (the run's two sums are rounded to the job's quantum here, once; everything downstream is integer arithmetic)
        fix_store(job, idx, fix_sums(fix_of(s_rho, job.fix_mul), fix_of(s_rl, job.fix_mul), 0, 0, rawc0, rawr, raws));
        const long long a = (long long)rawc0 + lane;
        job.r_c[idx] = (long long)len * a + (long long)len * (len - 1) / 2;
        job.r_r[idx] = (long long)len * rawr;
        job.r_s[idx] = (long long)len * raws;
        const int64_t key_in = ((int64_t)(cl0 + lane) * vd.dim[1] + rl) * vd.dim[2] + sl;
        job.r_key[idx] = (unsigned long long)(vd.key_base + key_in);
        // (whole-map jobs, unit tiles: the volume's group is its plane, keys inside a plane are below 2^31)
        if (job.kpar) job.kpar[idx] = ((unsigned long long)(((uint32_t)vd.group << 31) | (uint32_t)key_in) << 32) | idx;
    }
}

// The same records for a NARROW word -- a row of at most 16 voxels, which is every row of an atom's sphere box at the grid
// spacings of real maps -- by ONE thread: its (at most 16) densities are fetched together, the sums of a run are plain
// sequential fp64 sums.  k_run_index's wave-per-word loop spends a memory round trip or two per word and uses 10 of its 64
// lanes on such rows (2 x 71 us of an aggregateCloud's 0.55 ms of kernels, round 4); here 256 words of a block go at once.
constexpr int NARROW_ROW = 16;
template <typename JobRef>
__device__ inline void narrow_run_records(const JobRef &job, const Geom &g, const float *__restrict__ dens, const VolDesc &vd, uint64_t mw,
                                          int rl, int sl, uint32_t rec0) {
    const int rawc0 = vd.org[0], rawr = vd.org[1] + rl, raws = vd.org[2] + sl;
    float v[NARROW_ROW];
#pragma unroll
    for (int k = 0; k < NARROW_ROW; ++k) v[k] = ((mw >> k) & 1ull) ? fetch_wrapped(g, dens, rawc0 + k, rawr, raws) : 0.0f;
    uint64_t todo = run_starts(mw);
    uint32_t idx = rec0;
    while (todo) {
        const int a = ctz64(todo);
        todo &= todo - 1;
        const int b = run_end_of(mw, a), len = b - a + 1;
        double s_rho = 0.0, s_rl = 0.0;
#pragma unroll
        for (int k = 0; k < NARROW_ROW; ++k)
            if (k >= a && k <= b) { s_rho += (double)v[k]; s_rl += (double)v[k] * (double)k; }
        job.parent[idx] = (int32_t)idx;
        job.r_n[idx] = (uint32_t)len;
        fix_store(job, idx, fix_sums(fix_of(s_rho, job.fix_mul), fix_of(s_rl, job.fix_mul), 0, 0, rawc0, rawr, raws));
        const long long first = (long long)rawc0 + a;
        job.r_c[idx] = (long long)len * first + (long long)len * (len - 1) / 2;
        job.r_r[idx] = (long long)len * rawr;
        job.r_s[idx] = (long long)len * raws;
        const int64_t key_in = ((int64_t)a * vd.dim[1] + rl) * vd.dim[2] + sl;
        job.r_key[idx] = (unsigned long long)(vd.key_base + key_in);
        if (job.kpar) job.kpar[idx] = ((unsigned long long)(((uint32_t)vd.group << 31) | (uint32_t)key_in) << 32) | idx;
        ++idx;
    }
}

// The same records for a SPARSE word of a wide row -- at most SPARSE_BITS set voxels among its 64: the words of aggregateCloud's union
// job (a residue's or the whole domain's pooled voxels in a box that spans the structure: 32 k words of which a quarter hold a few
// short runs each) -- by ONE thread: the set voxels' densities are fetched together, the runs' sums are plain sequential fp64 sums in
// voxel order.  The wave-per-word loop of k_run_index spends two wave scans and a memory round trip per word, one word after the
// other (16 words a wave: 35 us for that job, round 6); here every word of a block goes at once.
constexpr int SPARSE_BITS = 16;
template <typename JobRef>
__device__ inline void sparse_run_records(const JobRef &job, const Geom &g, const float *__restrict__ dens, const VolDesc &vd, uint64_t mw,
                                          int wq, int rl, int sl, uint32_t rec0) {
    const int rawc0 = vd.org[0] + wq * 64, rawr = vd.org[1] + rl, raws = vd.org[2] + sl;
    float v[SPARSE_BITS];
    int pos[SPARSE_BITS];
    {
        uint64_t t = mw;
#pragma unroll
        for (int k = 0; k < SPARSE_BITS; ++k) {
            const bool have = t != 0ull;
            pos[k] = have ? ctz64(t) : 64;
            v[k] = have ? fetch_wrapped(g, dens, rawc0 + pos[k], rawr, raws) : 0.0f;
            t &= t - 1ull;      // (0 - 1 & 0 = 0)
        }
    }
    uint32_t idx = rec0;
    int first = -1, prev = -2;
    double s_rho = 0.0, s_rl = 0.0;
    auto flush = [&]() {
        const int len = prev - first + 1;
        job.parent[idx] = (int32_t)idx;
        job.r_n[idx] = (uint32_t)len;
        fix_store(job, idx, fix_sums(fix_of(s_rho, job.fix_mul), fix_of(s_rl, job.fix_mul), 0, 0, rawc0, rawr, raws));
        const long long a0 = (long long)rawc0 + first;
        job.r_c[idx] = (long long)len * a0 + (long long)len * (len - 1) / 2;
        job.r_r[idx] = (long long)len * rawr;
        job.r_s[idx] = (long long)len * raws;
        const int64_t key_in = ((int64_t)(wq * 64 + first) * vd.dim[1] + rl) * vd.dim[2] + sl;
        job.r_key[idx] = (unsigned long long)(vd.key_base + key_in);
        if (job.kpar) job.kpar[idx] = ((unsigned long long)(((uint32_t)vd.group << 31) | (uint32_t)key_in) << 32) | idx;
        ++idx;
    };
#pragma unroll
    for (int k = 0; k < SPARSE_BITS; ++k) {
        if (pos[k] < 64) {
            if (pos[k] != prev + 1) {
                if (first >= 0) flush();
                first = pos[k]; s_rho = 0.0; s_rl = 0.0;
            }
            s_rho += (double)v[k];
            s_rl += (double)v[k] * (double)pos[k];
            prev = pos[k];
        }
    }
    if (first >= 0) flush();
}

// ------------------------------------------------------------------------------------
// Run indexing + per-run statistics.  Block = 256 threads = one chunk of WPB words (256, or 64 for a job of few words:
// the union job of an aggregateCloud has 40 k wide, sparse words -- 164 blocks of 256 words left a third of the chip idle
// behind four waves a block that walked 64 words each, one or two memory round trips a word: 92 us, round 4).
// Phase 1 (thread per word): count runs, block scan, one atomicAdd per block -> run_base; the word's volume and its place
// in it go to LDS.  Narrow words (narrow_run_records) are finished by their own thread.
// Phase 2 (wave per remaining word, lane per voxel): wrapped density fetch (getPointDensityFromCrs) -- the NEXT word's
// fetch is in flight while this one's records are made --, two wave prefix sums (rho, rho*lane), run sums by difference
// at the run's first lane.  fromCrsList's sums (ccp4.py:534-545) become per-run partials.
// ------------------------------------------------------------------------------------
template <int WPB>
__global__ void __launch_bounds__(256) k_run_index(Job job, const float *__restrict__ dens, const Geom *__restrict__ gp) {
    static_assert(WPB == 256 || WPB == 64, "a word per thread of the block, or of its first wave");
    constexpr int SCAN_WAVES = WPB / 64, PER_WAVE = WPB / 4;
    __shared__ uint64_t s_mask[WPB];
    __shared__ uint32_t s_off[WPB];
    __shared__ VolDesc s_vd[WPB];
    __shared__ int s_row[WPB][3];        // wq, rl, sl of the word
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t chunk0 = (int64_t)blockIdx.x * WPB;
    const int64_t w = chunk0 + tid;
    const bool mine = tid < WPB && w < job.total_words;
    const uint64_t m = mine ? job.mask[w] : 0ull;
    const uint32_t cnt = (uint32_t)popc64(run_starts(m));
    // block exclusive scan of cnt
    uint32_t x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wsum[wv] = x;
    if (tid < WPB) s_mask[tid] = m;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < SCAN_WAVES; ++k) tot += s_wsum[k];
        s_base = tot ? atomicAdd(&job.ctr->n_runs, tot) : 0u;
    }
    __syncthreads();
    uint32_t wpre = 0;
    for (int k = 0; k < wv && k < SCAN_WAVES; ++k) wpre += s_wsum[k];
    const uint32_t off = s_base + wpre + x - cnt;
    if (mine) job.run_base[w] = off;
    // (every thread searches for its own word's volume: the per-word loop below is serial)
    VolDesc my_vd;
    int wq = 0, rl = 0, sl = 0;
    bool narrow = false, sparse = false;
    if (m != 0ull) {
        my_vd = job.vols[find_vol(job.vols, job.n_vols, w)];
        const int64_t rem = w - my_vd.word_base;
        wq = (int)(rem % my_vd.row_words);
        const int64_t row = rem / my_vd.row_words;
        rl = (int)(row % my_vd.dim[1]);
        sl = (int)(row / my_vd.dim[1]);
        narrow = my_vd.dim[0] <= NARROW_ROW;
        sparse = !narrow && popc64(m) <= SPARSE_BITS;
        if (narrow || sparse) s_mask[tid] = 0ull;      // the thread's own work; the wave-per-word loop below skips it
        else { s_off[tid] = off; s_vd[tid] = my_vd; s_row[tid][0] = wq; s_row[tid][1] = rl; s_row[tid][2] = sl; }
    }
    __syncthreads();

    const Geom &g = *gp;
    if (narrow) narrow_run_records(job, g, dens, my_vd, m, rl, sl, off);
    else if (sparse) sparse_run_records(job, g, dens, my_vd, m, wq, rl, sl, off);
    // the words of my wave that are left, as a bit per slot
    uint64_t todo = 0ull;
    {
        const bool live = lane < PER_WAVE && s_mask[wv * PER_WAVE + lane] != 0ull;
        todo = __ballot(live);
    }
    auto fetch = [&](int slot) -> double {
        const uint64_t mw = s_mask[slot];
        const VolDesc &vd = s_vd[slot];
        return ((mw >> lane) & 1ull) ? (double)fetch_wrapped(g, dens, vd.org[0] + s_row[slot][0] * 64 + lane, vd.org[1] + s_row[slot][1], vd.org[2] + s_row[slot][2]) : 0.0;
    };
    double rho_next = todo ? fetch(wv * PER_WAVE + ctz64(todo)) : 0.0;
    while (todo) {
        const int slot = wv * PER_WAVE + ctz64(todo);
        todo &= todo - 1;
        const double rho = rho_next;
        if (todo) rho_next = fetch(wv * PER_WAVE + ctz64(todo));
        const VolDesc &vd = s_vd[slot];
        const int q = s_row[slot][0], r = s_row[slot][1], t = s_row[slot][2];
        word_run_records_of(job, vd, s_mask[slot], lane, rho, q * 64, r, t, vd.org[0] + q * 64, vd.org[1] + r, vd.org[2] + t, s_off[slot]);
    }
}

// ------------------------------------------------------------------------------------
// Lock-free union-find on run indices (parent[x] <= x; roots link to the smaller root).
// atomicMin is authoritative; stale reads in find() only cost a retry: every value a load can return was the node's
// parent at some time, i.e. a smaller-or-equal id of the SAME set, so walking it, halving with it (atomic min) and
// comparing roots stay valid, and a unite only ends on the memory-side return value of its atomic min.  That is why
// find() may use cached (workgroup-scope) loads: an L2 hit instead of a trip to the memory side on this multi-XCD part
// (the cross-tile union phase: 40 -> 35 us).
// ------------------------------------------------------------------------------------
__device__ inline int uf_load(const int32_t *p, int x) {
    return __hip_atomic_load(p + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// find with path halving by fire-and-forget atomic min (monotone: never undoes a union)
__device__ inline int uf_find(int32_t *p, int x) {
    int q;
    while ((q = uf_load(p, x)) != x) {
        const int gp = uf_load(p, q);
        if (gp != q) atomicMin(p + x, gp);
        x = gp;
    }
    return x;
}
// The same with the two finds walking in step: their loads share the memory round trips (a union is a chain of dependent
// trips: two parallel finds + one atomic min instead of three in a row).
__device__ inline void uf_unite2(int32_t *p, int a, int b) {
    while (true) {
        int qa = uf_load(p, a), qb = uf_load(p, b);
        while (qa != a || qb != b) {
            const int ga = qa != a ? uf_load(p, qa) : qa, gb = qb != b ? uf_load(p, qb) : qb;
            if (qa != a && ga != qa) atomicMin(p + a, ga);
            if (qb != b && gb != qb) atomicMin(p + b, gb);
            a = ga; b = gb;
            qa = uf_load(p, a); qb = uf_load(p, b);
        }
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(p + a, b);
        if (old == a) return;
        a = old;
    }
}

// Optimistic union for the cross-tile merge, where most elements are still roots: hook the larger id under the smaller with
// ONE atomic min and no find; if the larger one already had a parent, that parent and the smaller id are what is left to
// unite (the min keeps parent[x] <= x and never undoes a link) -- hooked the same way, a trip per level instead of the
// two or three of a find-then-min.  A long chain (a blob that spans the map) falls back to root-to-root unions after four
// levels: hooking all the way down measured 4 us faster on noise but 4.6 us slower on a protein-like map (deeper trees
// for k_resolve_tiles); four levels keep the first and lose nothing on the second (r03, tools/exp/ab.sh).
__device__ inline void uf_hook(int32_t *p, int a, int b) {
#pragma unroll
    for (int level = 0; level < 4; ++level) {
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(p + a, b);
        if (old == a || old == b) return;
        a = old;
    }
    uf_unite2(p, a, b);
}

__device__ inline void uf_unite(int32_t *p, int a, int b) {
    while (true) {
        a = uf_find(p, a);
        b = uf_find(p, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(p + a, b);
        if (old == a) return;
        a = old;
    }
}

// ---- The same union-find on PACKED parents (whole-map jobs): a value is key32 << 32 | id, ordered by key (keys are unique
// positions, so the order of the values is the order of the keys), kpar[x] <= P(x) with equality exactly at roots, and the
// larger VALUE hangs under the smaller: a root is the component with the smallest first key of its set.  Everything above
// carries over (the min is authoritative, every value a load can return was the node's parent at some time); a value names
// its node (the low half), so the walk needs no second table.
typedef unsigned long long kp_t;
constexpr kp_t KP_UNUSED = ~0ull;
__device__ __forceinline__ uint32_t kp_id(kp_t v) { return (uint32_t)v; }
__device__ inline kp_t kuf_load(const kp_t *p, uint32_t id) {
    return __hip_atomic_load(p + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// root of the set of the node named by x (x = a value read from the table, or P of a node)
__device__ inline kp_t kuf_find_from(kp_t *p, kp_t x) {
    kp_t q;
    while ((q = kuf_load(p, kp_id(x))) != x) {
        const kp_t gp = kuf_load(p, kp_id(q));
        if (gp != q) atomicMin(p + kp_id(x), gp);
        x = gp;
    }
    return x;
}
__device__ inline void kuf_unite_vals(kp_t *p, kp_t a, kp_t b) {
    while (true) {
        kp_t qa = kuf_load(p, kp_id(a)), qb = kuf_load(p, kp_id(b));
        while (qa != a || qb != b) {   // the two finds walk in step: their loads share the round trips
            const kp_t ga = qa != a ? kuf_load(p, kp_id(qa)) : qa, gb = qb != b ? kuf_load(p, kp_id(qb)) : qb;
            if (qa != a && ga != qa) atomicMin(p + kp_id(a), ga);
            if (qb != b && gb != qb) atomicMin(p + kp_id(b), gb);
            a = ga; b = gb;
            qa = kuf_load(p, kp_id(a)); qb = kuf_load(p, kp_id(b));
        }
        if (a == b) return;
        if (a < b) { const kp_t t = a; a = b; b = t; }
        const kp_t old = atomicMin(p + kp_id(a), b);
        if (old == a) return;
        a = old;
    }
}
// optimistic hook (see uf_hook) of two values that name nodes of the two sets -- e.g. what kpar[] held for the two components
// a moment ago: an ancestor-or-self each, which is all a union needs
__device__ inline void kuf_hook_vals(kp_t *p, kp_t a, kp_t b) {
#pragma unroll
    for (int level = 0; level < 4; ++level) {
        if (a == b) return;
        if (a < b) { const kp_t t = a; a = b; b = t; }
        const kp_t old = atomicMin(p + kp_id(a), b);
        if (old == a || old == b) return;
        a = old;
    }
    kuf_unite_vals(p, a, b);
}
__device__ inline void kuf_unite(kp_t *p, uint32_t ia, uint32_t ib) {
    kuf_unite_vals(p, kuf_load(p, ia), kuf_load(p, ib));
}

// Index of the run of word `nm` that contains set bit p.
__device__ inline uint32_t run_of_bit(uint64_t nm, uint32_t base, int p, int *start_out) {
    int st = run_start_of(nm, p);
    *start_out = st;
    return base + (uint32_t)popc64(run_starts(nm) & bits_below(st));
}

// Four threads per word, one per neighbour ROW -- (r-1,s), (r-1,s-1), (r,s-1), (r+1,s-1) -- : each unites the runs of the
// word with the touching runs of words w-1, w, w+1 of its row; the first also with the previous word of the word's own row.
// Two runs [a,b] and [a',b'] in adjacent rows touch iff [a-1,b+1] meets [a',b']
// (Chebyshev distance <= 1 == cdist <= sqrt(3) on integer coordinates, cutils.pyx:43,62).
// The neighbour words and their run bases are loaded up front and unconditionally (a clamped address where there is no such
// word): one memory round trip before the unions instead of one or two per neighbour; a union is a chain of dependent
// trips, so the rows of a word go side by side (round 4: a thread per word with loads on demand was 2 x 65-80 us of an
// aggregateCloud, most of it in a union job of 40 k words = 164 blocks).
// (Round 6, measured and dropped: optimistic hooks -- uf_hook, one atomic min where both runs are still roots -- as in the cross-tile merge:
//  43 -> 64 us for the union job of a 2 000-atom entry; the runs of a cloud touch each other many times over, and a hook that finds its
//  target taken walks down one level per trip where two finds that read cached parents compress the path.)
__global__ void __launch_bounds__(256) k_union(Job job) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t w = t >> 2;
    const int nb = (int)(t & 3);
    if (w >= job.total_words) return;
    const uint64_t m = job.mask[w];
    if (m == 0ull) return;
    const VolDesc vd = job.vols[find_vol(job.vols, job.n_vols, w)];
    const int64_t rem = w - vd.word_base;
    const int wq = (int)(rem % vd.row_words);
    const int64_t row = rem / vd.row_words;
    const int rl = (int)(row % vd.dim[1]);
    const int sl = (int)(row / vd.dim[1]);
    uint64_t nmask[3];
    uint32_t nbase[3];
    {
        const int dr = nb == 3 ? 1 : (nb == 2 ? 0 : -1), ds = nb == 0 ? 0 : -1;
        const int r2 = rl + dr, s2 = sl + ds;
        const bool row_ok = r2 >= 0 && r2 < vd.dim[1] && s2 >= 0;
        const int64_t rowbase = vd.word_base + ((int64_t)(row_ok ? s2 : sl) * vd.dim[1] + (row_ok ? r2 : rl)) * vd.row_words;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int w2 = wq + d - 1;
            const bool ok = row_ok && w2 >= 0 && w2 < vd.row_words;
            const int64_t at = ok ? rowbase + w2 : w;
            const uint64_t v = job.mask[at];
            nbase[d] = job.run_base[at];
            nmask[d] = ok ? v : 0ull;
        }
    }
    const bool with_prev = nb == 0 && wq > 0;
    const uint64_t pm = with_prev ? job.mask[w - 1] : 0ull;
    const uint32_t pbase = job.run_base[with_prev ? w - 1 : w];
    const uint32_t base = job.run_base[w];
    if (!(nmask[0] | nmask[1] | nmask[2] | (pm >> 63))) return;
    uint64_t todo = run_starts(m);
    uint32_t k = 0;
    while (todo) {
        const int a = ctz64(todo);
        todo &= todo - 1;
        const int b = run_end_of(m, a);
        const int me = (int)(base + k);
        ++k;
        if (a == 0 && (pm >> 63)) {
            int st;
            uf_unite2(job.parent, me, (int)run_of_bit(pm, pbase, 63, &st));
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint64_t nm = nmask[d];
            int lo = a - 1 - 64 * (d - 1), hi = b + 1 - 64 * (d - 1);
            if (nm == 0ull || hi < 0 || lo > 63) continue;
            if (lo < 0) lo = 0;
            if (hi > 63) hi = 63;
            uint64_t hit = nm & (bits_below(hi + 1) & ~bits_below(lo));
            while (hit) {
                const int p = ctz64(hit);
                int st;
                const uint32_t other = run_of_bit(nm, nbase[d], p, &st);
                uf_unite2(job.parent, me, (int)other);
                hit &= ~bits_below(run_end_of(nm, st) + 1);
            }
        }
    }
}

template <typename JobRef>
__device__ inline uint32_t n_components(const JobRef &job) { return job.comps_are_runs ? job.ctr->n_runs : job.ctr->n_comps; }

// Thread per component: flatten, and fold non-root partial sums into the root record.  Runs are numbered in word order,
// so the lanes of a wave mostly belong to a handful of components: the lanes that share a root are summed in the wave
// first and ONE lane sends the nine atomics (a 200-run atom cloud used to send 200 x 9 to the same cache line, where they
// serialise).
// (one wave's share: component i of the lane, or none -- every lane of the wave must call)
// slots: the roots take table rows in the order they come (one atomic per wave): the UNORDERED form of a job whose caller does not
// read its blobs in key order -- aggregateCloud's union job, whose rows the host orders by the pooled clouds they hold -- and which
// therefore needs neither the painted keys nor the ranks (k_paint_keys, k_emit: two launches)
__device__ inline void resolve_wave(const Job &job, uint32_t i, bool valid, int lane, bool slots = false) {
    int root = -1;
    bool is_root = false;
    if (valid) {
        root = uf_find(job.parent, (int)i);
        if (root == (int)i) { root = -1; is_root = true; } else job.parent[i] = root;
    }
    if (slots) {
        const unsigned long long roots = __ballot(is_root);
        if (roots) {
            uint32_t first = 0;
            if (lane == ctz64(roots)) first = atomicAdd(&job.ctr->n_blobs, (uint32_t)popc64(roots));
            first = __shfl(first, ctz64(roots));
            if (is_root) job.r_rank[i] = first + (uint32_t)popc64(roots & bits_below(lane));
        }
    }
    uint32_t n = 0;
    FixSums sum = fix_zero();
    unsigned long long c = 0, r = 0, sv = 0, key = ~0ull;
    if (root >= 0) {
        n = job.r_n[i]; sum = fix_load(job, i);
        c = (unsigned long long)job.r_c[i]; r = (unsigned long long)job.r_r[i]; sv = (unsigned long long)job.r_s[i]; key = job.r_key[i];
    }
    unsigned long long todo = __ballot(root >= 0);
    while (todo) {
        const int first = ctz64(todo);
        const int r0 = __shfl(root, first);
        const bool mine = root == r0;
        const unsigned long long group = __ballot(mine);
        todo &= ~group;
        uint32_t gn = mine ? n : 0u;
        FixSums g_sum = mine ? sum : fix_zero();
        unsigned long long g_c = mine ? c : 0ull, g_r = mine ? r : 0ull, g_s = mine ? sv : 0ull, g_key = mine ? key : ~0ull;
        if (group & (group - 1)) {           // more than one lane: butterfly over the wave (lanes outside the group carry the neutral element)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                gn += __shfl_xor(gn, d);
                g_sum.rho += __shfl_xor(g_sum.rho, d);
                g_sum.c_lo += __shfl_xor(g_sum.c_lo, d); g_sum.c_hi += __shfl_xor(g_sum.c_hi, d);
                g_sum.r_lo += __shfl_xor(g_sum.r_lo, d); g_sum.r_hi += __shfl_xor(g_sum.r_hi, d);
                g_sum.s_lo += __shfl_xor(g_sum.s_lo, d); g_sum.s_hi += __shfl_xor(g_sum.s_hi, d);
                g_c += __shfl_xor(g_c, d); g_r += __shfl_xor(g_r, d); g_s += __shfl_xor(g_s, d);
                const unsigned long long k2 = __shfl_xor(g_key, d);
                g_key = k2 < g_key ? k2 : g_key;
            }
        }
        if (lane == first) {
            atomicAdd(&job.r_n[r0], gn);
            fix_fold(job, (uint32_t)r0, g_sum);
            atomicAdd((unsigned long long *)&job.r_c[r0], g_c);
            atomicAdd((unsigned long long *)&job.r_r[r0], g_r);
            atomicAdd((unsigned long long *)&job.r_s[r0], g_s);
            atomicMin(&job.r_key[r0], g_key);
        }
    }
}
__global__ void __launch_bounds__(256) k_resolve(Job job, int slots) {
    const uint32_t n_runs = n_components(job);
    const int lane = lane_id();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n_runs; i0 += stride)      // (wave-uniform trip count)
        resolve_wave(job, i0 + (uint32_t)lane, i0 + (uint32_t)lane < n_runs, lane, slots != 0);
}

// A root paints its first key: the bit, the byte counter of the bit's 256-key cell, the 16-bit counter of its 2048-key bucket.
template <typename JobRef>
__device__ __forceinline__ void paint_key(const JobRef &job, unsigned long long key) {
    const uint32_t f = (uint32_t)((key >> 6) / KEY_FINE);
    const uint32_t cell = (uint32_t)(key >> 8);
    atomicOr((unsigned long long *)&job.key_bits[key >> 6], 1ull << (key & 63));
    atomicAdd(&job.mid_count[cell >> 2], 1u << ((cell & 3u) * 8u));
    atomicAdd(&job.fine_count[f >> 1], 1u << ((f & 1u) * 16u));   // (a fine bucket holds 2048 keys: its count fits 16 bits)
}

// Blob order = ascending key of the blob's first voxel in the reference's c-major
// enumeration (cutils.pyx:199 + 59-69).  Keys are unique positions, so the rank of a blob
// is a prefix population count over a bitmap of first-voxel keys -- no sort needed, and no scan launch: k_paint_keys
// bumps a counter per 2048 keys beside the bit, every k_emit block sums the counters into a <= 2048-entry prefix table
// in LDS (64 KiB of L2 reads at 256^3), and rank(key) = table entry + <= 7 counters + <= 31 bitmap words.
__global__ void __launch_bounds__(256) k_paint_keys(Job job) {
    const uint32_t n_runs = n_components(job);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += gridDim.x * blockDim.x) {
        const int32_t par = job.parent[i];
        const uint32_t cnt = job.r_n[i];
        const unsigned long long key = job.r_key[i];    // (loaded beside the other two, not after them: one round trip, then the atomics)
        if (par != (int32_t)i || cnt == 0u) continue;   // not a root / unused id
        paint_key(job, key);
    }
}

// x of another lane through the DPP network (no LDS crossbar trip); lanes without a source read 0
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp0(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, true);
}
constexpr int DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
// inclusive prefix sums inside each HALF of the wave (lanes 0..31 and 32..63 separately): five DPP adds
__device__ __forceinline__ uint32_t half_scan(uint32_t x) {
    x += dpp0<DPP_ROW_SHR + 1>(x);
    x += dpp0<DPP_ROW_SHR + 2>(x);
    x += dpp0<DPP_ROW_SHR + 4>(x);
    x += dpp0<DPP_ROW_SHR + 8>(x);
    x += dpp0<DPP_ROW_BCAST15, 0xa>(x);   // rows 1 and 3 add the totals of rows 0 and 2
    return x;
}

// inclusive prefix sums over the whole wave: six DPP adds
__device__ __forceinline__ uint32_t wave_scan(uint32_t x) {
    x = half_scan(x);
    x += dpp0<DPP_ROW_BCAST31, 0xc>(x);   // rows 2 and 3 add the total of rows 0 and 1
    return x;
}
// sum of the eight 16-bit counters of a quad, and of its first `take` (0..8) ones
__device__ __forceinline__ uint32_t sum_u16x8(const uint4 q) {
    const uint32_t a = (q.x & 0xffffu) + (q.x >> 16), b = (q.y & 0xffffu) + (q.y >> 16), c = (q.z & 0xffffu) + (q.z >> 16), d = (q.w & 0xffffu) + (q.w >> 16);
    return (a + b) + (c + d);
}
__device__ __forceinline__ uint32_t sum_u16_first(const uint4 q, int take) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += (2 * k < take ? (w[k] & 0xffffu) : 0u) + (2 * k + 1 < take ? (w[k] >> 16) : 0u);
    return acc;
}

// Exclusive prefix table over groups of fine_per_group counters, built by the calling block in LDS; returns the total.
// A thread owns KEY_GROUPS / 256 = 4 consecutive groups: all its loads are in flight at once (the table is one memory
// round trip + one block scan, not a loop of dependent trips; 32 KiB per block at 256^3).  fine_count is padded to whole
// groups and cleared with the bitmap.
// The table in two steps, so that a caller can put other loads in flight between them: rank_table_issue loads the calling
// thread's counters (groups of 16: two quads a group -- the common case; larger groups are summed in place, a loop of loads),
// rank_table_finish sums them, scans the block and writes the table (two barriers).
template <int NT>
struct RankTableLoads {
    static constexpr int PER = KEY_GROUPS / NT;
    uint4 q[2 * PER];
    uint32_t v[PER];
    bool summed;
};
template <int NT = 256, typename JobRef = Job>
__device__ __forceinline__ void rank_table_issue(const JobRef &job, RankTableLoads<NT> &ld) {
    constexpr int PER = KEY_GROUPS / NT;
    const int tid = threadIdx.x;
    const int G = job.fine_per_group, n_groups = job.n_groups, Q = G / 8;   // quads (8 counters = 16 B) per group
    const uint4 *fine4 = reinterpret_cast<const uint4 *>(job.fine_count);
    ld.summed = Q != 2;
    if (Q == 2) {   // (maps up to 2^25 keys: 256^3 fused)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid * PER + k;
            ld.q[2 * k] = e < n_groups ? fine4[(size_t)e * 2] : make_uint4(0, 0, 0, 0);
            ld.q[2 * k + 1] = e < n_groups ? fine4[(size_t)e * 2 + 1] : make_uint4(0, 0, 0, 0);
        }
    } else if (job.group_count) {   // (larger whole-map jobs: the groups' totals are there, k_group_counts)
        const uint32_t *gc = job.group_count;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid * PER + k;
            ld.v[k] = e < n_groups ? gc[e] : 0u;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid * PER + k;
            uint32_t acc = 0;
            if (e < n_groups)
                for (int j = 0; j < Q; ++j) acc += sum_u16x8(fine4[(size_t)e * Q + j]);
            ld.v[k] = acc;
        }
    }
}
template <int NT = 256>
__device__ __forceinline__ uint32_t rank_table_finish(RankTableLoads<NT> &ld, uint32_t *s_pre /* [KEY_GROUPS] */, uint32_t *s_wave /* [NT / 64] */) {
    constexpr int PER = KEY_GROUPS / NT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!ld.summed) {
#pragma unroll
        for (int k = 0; k < PER; ++k) ld.v[k] = sum_u16x8(ld.q[2 * k]) + sum_u16x8(ld.q[2 * k + 1]);
    }
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) mine += ld.v[k];
    const uint32_t x = wave_scan(mine);   // (six DPP adds: the shuffle form went through the LDS crossbar six times, one after the other)
    if (lane == 63) s_wave[wv] = x;
    __syncthreads();
    uint32_t pre = x - mine, total = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) { const uint32_t w = s_wave[k]; pre += k < wv ? w : 0u; total += w; }
#pragma unroll
    for (int k = 0; k < PER; ++k) { s_pre[tid * PER + k] = pre; pre += ld.v[k]; }
    __syncthreads();
    return total;
}
// Exclusive prefix table over groups of fine_per_group counters, built by the calling block in LDS; returns the total.
// A thread owns KEY_GROUPS / NT consecutive groups: all its loads are in flight at once (the table is one memory
// round trip + one block scan, not a loop of dependent trips; 32 KiB per block at 256^3).  fine_count is padded to whole
// groups and cleared with the bitmap.
template <int NT = 256, typename JobRef = Job>
__device__ inline uint32_t rank_table_lds(const JobRef &job, uint32_t *s_pre /* [KEY_GROUPS] */, uint32_t *s_wave /* [NT / 64] */) {
    RankTableLoads<NT> ld;
    rank_table_issue<NT>(job, ld);
    return rank_table_finish<NT>(ld, s_pre, s_wave);
}

// Totals of the groups of 16-bit counters (Job::group_count): a thread per group, behind k_resolve_tiles (which paints).
__global__ void __launch_bounds__(256) k_group_counts(Job job) {
    const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= job.n_groups) return;
    const int Q = job.fine_per_group / 8;
    const uint4 *fine4 = reinterpret_cast<const uint4 *>(job.fine_count) + (size_t)e * Q;
    uint32_t acc = 0;
    for (int j = 0; j < Q; j += 4) {   // (Q is a power of two >= 4 here: groups of 32 counters or more)
        const uint4 a = fine4[j], b = fine4[j + 1], c = fine4[j + 2], d = fine4[j + 3];
        acc += (sum_u16x8(a) + sum_u16x8(b)) + (sum_u16x8(c) + sum_u16x8(d));
    }
    job.group_count[e] = acc;
}

// Number of painted keys below `key` (s_pre: this block's rank_table_lds) = table entry + the 16-bit counters of the key's group
// before its bucket + the byte counters of the bucket before its cell + the bits of the cell's four words below the key.
// Branch-free for the common group size: five loads (2 x 16 B of counters, 8 B of bytes, 2 x 16 B of bitmap), all in flight
// together, masked afterwards -- the lanes of a wave do not wait for each other's trip counts.
// (keys are below 2^32 -- two planes of fewer than 2^31 voxels, or the volumes of a batch -- so word indices fit 32 bits)
struct RankLoads {
    uint4 q0, q1;        // the 16 counters of the key's group (groups of 16)
    uint2 mid;           // the 8 byte counters of its bucket
    ulonglong2 b0, b1;   // the 4 bitmap words of its cell
};
template <typename JobRef>
__device__ __forceinline__ void rank_issue(const JobRef &job, unsigned long long key, RankLoads &ld) {   // groups of 16 only
    const uint32_t kw = (uint32_t)(key >> 6), f = kw / (uint32_t)KEY_FINE, e = f >> 4;
    const uint4 *fine4 = reinterpret_cast<const uint4 *>(job.fine_count) + (size_t)e * 2;
    ld.q0 = fine4[0]; ld.q1 = fine4[1];
    ld.mid = reinterpret_cast<const uint2 *>(job.mid_count)[f];
    const ulonglong2 *bits2 = reinterpret_cast<const ulonglong2 *>(job.key_bits + (kw & ~3u));
    ld.b0 = bits2[0]; ld.b1 = bits2[1];
}
__device__ __forceinline__ uint32_t rank_finish(const uint32_t *s_pre, unsigned long long key, const RankLoads &ld) {
    const uint32_t kw = (uint32_t)(key >> 6), f = kw / (uint32_t)KEY_FINE, e = f >> 4;
    const int nf = (int)(f & 15u), cell = (int)((kw & (uint32_t)(KEY_FINE - 1)) >> 2), wi = (int)(kw & 3u);
    uint32_t rank = s_pre[e] + sum_u16_first(ld.q0, nf) + sum_u16_first(ld.q1, nf - 8);
    const uint32_t lo = cell >= 4 ? ld.mid.x : (ld.mid.x & ((1u << (8 * cell)) - 1u));
    const uint32_t hi = cell > 4 ? (ld.mid.y & ((1u << (8 * (cell - 4))) - 1u)) : 0u;
    rank += __builtin_amdgcn_sad_u8(lo, 0u, 0u) + __builtin_amdgcn_sad_u8(hi, 0u, 0u);
    const uint64_t below = bits_below((int)(key & 63));
    const uint64_t w[4] = {ld.b0.x, ld.b0.y, ld.b1.x, ld.b1.y};
#pragma unroll
    for (int k = 0; k < 4; ++k) rank += (uint32_t)popc64(k < wi ? w[k] : (k == wi ? w[k] & below : 0ull));
    return rank;
}
template <typename JobRef>
__device__ inline uint32_t rank_of_key(const JobRef &job, const uint32_t *s_pre, unsigned long long key) {
    const int shift = job.fine_shift;
    if (shift == 4) {
        RankLoads ld;
        rank_issue(job, key, ld);
        return rank_finish(s_pre, key, ld);
    }
    const uint32_t kw = (uint32_t)(key >> 6), f = kw / (uint32_t)KEY_FINE, e = f >> shift;
    const int Q = 1 << (shift - 3);
    uint32_t rank = s_pre[e];
    const uint4 *fine4 = reinterpret_cast<const uint4 *>(job.fine_count) + (size_t)e * Q;
    const int nf = (int)(f - (e << shift));                // counters of my group before mine
    const uint2 mid = reinterpret_cast<const uint2 *>(job.mid_count)[f];   // the 8 byte counters of my bucket (the tables are padded to whole buckets)
    const ulonglong2 *bits2 = reinterpret_cast<const ulonglong2 *>(job.key_bits + (kw & ~3u));   // my cell: 4 words, 32-B aligned
    const int cell = (int)((kw & (uint32_t)(KEY_FINE - 1)) >> 2);   // cells of my bucket before mine: < 8
    const int wi = (int)(kw & 3u);                         // words of my cell before mine
    const ulonglong2 b0 = bits2[0], b1 = bits2[1];
    for (int k = 0; k * 8 < nf; ++k) rank += sum_u16_first(fine4[k], nf - k * 8);
    {   // bytes 0 .. cell - 1 of the eight
        const uint32_t lo = cell >= 4 ? mid.x : (mid.x & ((1u << (8 * cell)) - 1u));
        const uint32_t hi = cell > 4 ? (mid.y & ((1u << (8 * (cell - 4))) - 1u)) : 0u;
        rank += __builtin_amdgcn_sad_u8(lo, 0u, 0u) + __builtin_amdgcn_sad_u8(hi, 0u, 0u);
    }
    const uint64_t below = bits_below((int)(key & 63));
    const uint64_t w[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
    for (int k = 0; k < 4; ++k) rank += (uint32_t)popc64(k < wi ? w[k] : (k == wi ? w[k] & below : 0ull));
    return rank;
}

// Position of the k-th (0-based) set bit of m (k < popcount(m)).
__device__ __forceinline__ int nth_set_bit(uint64_t m, int k) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {
        const int c = popc64(m & ((1ull << w) - 1ull));
        if (k >= c) { k -= c; m >>= w; pos += w; }
    }
    return pos;
}

// One blob table row (DensityBlob.fromCrsList, ccp4.py:542-545), the root's rank and -- whole-map jobs -- its signed label.
template <typename JobRef>
__device__ __forceinline__ void emit_row_ranked(const JobRef &job, const Geom &g, bool whole_map, int64_t key_base1, uint32_t id, uint32_t n_vox,
                                                unsigned long long first_key, const FixSums &fs, long long ic, long long ir, long long is, uint32_t rank) {
    const double tot_q = (double)fs.rho, rc = fix_moment(fs.c_lo, fs.c_hi), rr = fix_moment(fs.r_lo, fs.r_hi), rs = fix_moment(fs.s_lo, fs.s_hi);
    const double tot = tot_q / job.fix_mul;      // (a power of two: exact)
    const int vi = whole_map ? ((int64_t)first_key >= key_base1 ? 1 : 0) : find_vol_by_key(job.vols, job.n_vols, (int64_t)first_key);
    // signed by its list, numbered by its rank in the WHOLE table: k_labels_tiles takes the blobs of volume 0 off the
    // labels of volume 1 (ctr->n_blobs_vol0) -- here that count would be a second dependent round trip before any root
    if (whole_map) job.label_of_comp[id] = job.vol_sign[vi] > 0 ? 1 + (int32_t)rank : -1 - (int32_t)rank;
    const VolDesc vd = job.vols[vi];
    job.r_rank[id] = rank;
    if (rank >= job.blob_cap) return;   // (more blobs than rows: the job is flagged and runs again with the worst-case table)
    const double n = (double)n_vox;
    double wc[3] = {rc / tot_q, rr / tot_q, rs / tot_q};   // (the quantum cancels)
    double cc[3] = {(double)ic / n, (double)ir / n, (double)is / n};
    double xyz[3];
    crs2xyz_frac(g, wc, xyz);
    job.b_centroid[3 * rank + 0] = xyz[0];
    job.b_centroid[3 * rank + 1] = xyz[1];
    job.b_centroid[3 * rank + 2] = xyz[2];
    crs2xyz_frac(g, cc, xyz);
    job.b_center[3 * rank + 0] = xyz[0];
    job.b_center[3 * rank + 1] = xyz[1];
    job.b_center[3 * rank + 2] = xyz[2];
    job.b_n[rank] = (int64_t)n_vox;
    job.b_total[rank] = tot;
    job.b_volume[rank] = g.unit_volume * n;
    job.b_key[rank] = (int64_t)first_key - vd.key_base;
    job.b_group[rank] = vd.group;
}
template <typename JobRef>
__device__ __forceinline__ void emit_row(const JobRef &job, const Geom &g, const uint32_t *s_pre, bool whole_map, int64_t key_base1, uint32_t id,
                                         uint32_t n_vox, unsigned long long first_key, const FixSums &fs, long long ic, long long ir, long long is) {
    emit_row_ranked(job, g, whole_map, key_base1, id, n_vox, first_key, fs, ic, ir, is, rank_of_key(job, s_pre, first_key));
}

// Grid-stride over component ids [first, n_comp): the roots among them emit their rows (generic jobs: every id; whole-map
// jobs: the unit components above the tiles' id ranges).  The next trip's first step rides along.
template <typename JobRef>
__device__ inline void emit_ids(const JobRef &job, const Geom &g, const uint32_t *s_pre, bool whole_map, int64_t key_base1, uint32_t first, uint32_t n_comp,
                                uint32_t block, uint32_t n_blocks) {   // (the caller's linear workgroup id and count: its grid may be 3-D)
    const uint32_t stride = n_blocks * blockDim.x;
    uint32_t i = first + block * blockDim.x + threadIdx.x;
    int32_t par = i < n_comp ? job.parent[i] : -1;
    uint32_t cnt = i < n_comp ? job.r_n[i] : 0u;
    unsigned long long key = i < n_comp ? job.r_key[i] : 0ull;
    for (; i < n_comp; i += stride) {
        const bool root = par == (int32_t)i && cnt != 0u;   // else: not a root / unused component id
        const uint32_t nx = i + stride;
        const int32_t par_next = nx < n_comp ? job.parent[nx] : -1;
        const uint32_t cnt_next = nx < n_comp ? job.r_n[nx] : 0u;
        const unsigned long long key_next = nx < n_comp ? job.r_key[nx] : 0ull;
        if (root) emit_row(job, g, s_pre, whole_map, key_base1, i, cnt, key, fix_load(job, i), job.r_c[i], job.r_r[i], job.r_s[i]);
        par = par_next;
        cnt = cnt_next;
        key = key_next;
    }
}

// Generic jobs -- thread per ROOT component: its rank and its blob table row.  Block 0 publishes the blob counts.  A root's
// work is a chain of dependent memory round trips -- (parent, n) -> (key + the whole record) -> (counters + bitmap words) --
// so each step issues everything the next one needs at once.
__global__ void __launch_bounds__(256) k_emit(Job job, const Geom *__restrict__ gp) {
    __shared__ uint32_t s_pre[KEY_GROUPS];
    __shared__ uint32_t s_wave[4];
    const uint32_t total = rank_table_lds(job, s_pre, s_wave);
    emit_ids(job, *gp, s_pre, false, INT64_MAX, 0u, n_components(job), blockIdx.x, gridDim.x);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        job.ctr->n_blobs = total;
        job.ctr->n_blobs_vol0 = total;
        if (total > job.blob_cap) atomicOr(&job.ctr->overflow, 2u);
    }
}

// Whole-map jobs: half a workgroup per tile, thread k of the half takes the tile's k-th root (k_resolve_tiles left a 256-bit
// root mask per tile), so ONE pass covers the job.  A root's sums are its own record plus what the other tiles' members of
// its blob POSTED to its tile's inbox (k_resolve_tiles): the inbox is summed in LDS first (round 4: this was a kernel of its
// own, k_paint_tiles, which also folded the first keys -- the roots hold them by construction now).  The unit components of
// the job (ids above the tiles' ranges; rare) follow, id by id.
__global__ void __launch_bounds__(256) k_emit_tiles(Job job, const Geom *__restrict__ gp) {
    __shared__ uint32_t s_pre[KEY_GROUPS];
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_acc[2][10][TILE_COMPS];   // per half: the seven FixSums fields, sum c / r / s
    __shared__ uint32_t s_accn[2][TILE_COMPS];
    const int half = threadIdx.x >> 7, k0 = threadIdx.x & 127;
    const int n_tiles = job.n_tiles;
    int tile = (int)blockIdx.x * 2 + half;
    uint64_t rm[4] = {0ull, 0ull, 0ull, 0ull};
    uint32_t n_in = 0;
    if (tile < n_tiles) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rm[q] = job.root_mask[(size_t)tile * 4 + q];
        n_in = job.inbox_count[(size_t)tile * INBOX_STRIDE];
    }
    // first key of volume 1 (fused green / red job); loaded before the table so that nothing below waits for it
    const int64_t key_base1 = job.n_vols > 1 ? job.vols[1].key_base : INT64_MAX;
    const uint32_t total = rank_table_lds(job, s_pre, s_wave);
    const Geom &g = *gp;
    for (int t0 = (int)blockIdx.x * 2; t0 < n_tiles; t0 += (int)gridDim.x * 2) {   // (block-uniform trip count: barriers inside)
        tile = t0 + half;
        n_in = min(n_in, (uint32_t)INBOX_CAP);
        const int next = tile + (int)gridDim.x * 2;
        uint64_t nm[4] = {0ull, 0ull, 0ull, 0ull};
        uint32_t n_in_next = 0;
        if (next < n_tiles) {
#pragma unroll
            for (int q = 0; q < 4; ++q) nm[q] = job.root_mask[(size_t)next * 4 + q];
            n_in_next = job.inbox_count[(size_t)next * INBOX_STRIDE];
        }
        // my inbox entries (<= 2 per thread), in flight beside the roots' records below
        InboxEntry e0, e1;
        const bool have0 = (uint32_t)k0 < n_in, have1 = (uint32_t)k0 + 128u < n_in;
        if (have0) e0 = job.inbox[(size_t)tile * INBOX_CAP + k0];
        if (have1) e1 = job.inbox[(size_t)tile * INBOX_CAP + k0 + 128];
        if (n_in != 0u) {   // (uniform over the half)
#pragma unroll
            for (int f = 0; f < 10; ++f) { s_acc[half][f][k0] = 0ull; s_acc[half][f][k0 + 128] = 0ull; }
            s_accn[half][k0] = 0u; s_accn[half][k0 + 128] = 0u;
        }
        __syncthreads();
        auto absorb = [&](const InboxEntry &e) {
            const uint32_t l = e.local;
            atomicAdd(&s_accn[half][l], e.n);
            atomicAdd(&s_acc[half][0][l], (unsigned long long)e.sum.rho);
            atomicAdd(&s_acc[half][1][l], (unsigned long long)e.sum.c_lo); atomicAdd(&s_acc[half][2][l], (unsigned long long)e.sum.c_hi);
            atomicAdd(&s_acc[half][3][l], (unsigned long long)e.sum.r_lo); atomicAdd(&s_acc[half][4][l], (unsigned long long)e.sum.r_hi);
            atomicAdd(&s_acc[half][5][l], (unsigned long long)e.sum.s_lo); atomicAdd(&s_acc[half][6][l], (unsigned long long)e.sum.s_hi);
            atomicAdd(&s_acc[half][7][l], e.c); atomicAdd(&s_acc[half][8][l], e.r); atomicAdd(&s_acc[half][9][l], e.s);
        };
        if (have0) absorb(e0);
        if (have1) absorb(e1);
        __syncthreads();
        const int c0 = popc64(rm[0]), c1 = c0 + popc64(rm[1]), c2 = c1 + popc64(rm[2]), n_roots = c2 + popc64(rm[3]);
        for (int k = k0; k < n_roots; k += 128) {
            const int q = k < c0 ? 0 : (k < c1 ? 1 : (k < c2 ? 2 : 3));
            const int before = q == 0 ? 0 : (q == 1 ? c0 : (q == 2 ? c1 : c2));
            const uint64_t word = q == 0 ? rm[0] : (q == 1 ? rm[1] : (q == 2 ? rm[2] : rm[3]));
            const uint32_t l = (uint32_t)(64 * q + nth_set_bit(word, k - before));
            const uint32_t id = (uint32_t)tile * (uint32_t)TILE_COMPS + l;
            uint32_t n_vox = job.r_n[id];
            const unsigned long long key = job.r_key[id];
            FixSums fs = fix_load(job, id);
            long long ic = job.r_c[id], ir = job.r_r[id], is = job.r_s[id];
            if (n_in != 0u) {
                n_vox += s_accn[half][l];
                fs.rho += (long long)s_acc[half][0][l];
                fs.c_lo += (long long)s_acc[half][1][l]; fs.c_hi += (long long)s_acc[half][2][l];
                fs.r_lo += (long long)s_acc[half][3][l]; fs.r_hi += (long long)s_acc[half][4][l];
                fs.s_lo += (long long)s_acc[half][5][l]; fs.s_hi += (long long)s_acc[half][6][l];
                ic += (long long)s_acc[half][7][l]; ir += (long long)s_acc[half][8][l]; is += (long long)s_acc[half][9][l];
            }
            emit_row(job, g, s_pre, true, key_base1, id, n_vox, key, fs, ic, ir, is);
        }
        __syncthreads();   // (the next round clears the tables)
#pragma unroll
        for (int q = 0; q < 4; ++q) rm[q] = nm[q];
        n_in = n_in_next;
    }
    {   // unit components (every run of a tile that overflowed LDS its own component): none on ordinary maps
        const uint32_t n_comp = n_components(job), first = (uint32_t)n_tiles * (uint32_t)TILE_COMPS;
        if (n_comp > first && job.ctr->overflow == 0u) emit_ids(job, g, s_pre, true, key_base1, first, n_comp, blockIdx.x, gridDim.x);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the table's totals, for the host and for k_labels_tiles
        uint32_t below1 = total;                 // blobs before volume 1; every blob when there is one volume
        if (key_base1 != INT64_MAX) below1 = rank_of_key(job, s_pre, (unsigned long long)key_base1);
        job.ctr->n_blobs = total;
        job.ctr->n_blobs_vol0 = below1;
        if (total > job.blob_cap) atomicOr(&job.ctr->overflow, 2u);
    }
}

// Voxel lists grouped by blob: offsets = exclusive scan of b_n (single block), then each
// voxel takes a slot in its blob with an atomic cursor (order inside a blob is a set).
__global__ void __launch_bounds__(1024) k_blob_offsets(Job job, int64_t *__restrict__ offsets, unsigned int *__restrict__ cursor) {   // (cursor: zeroed here -- a fill of its own was a launch)
    __shared__ long long s_w[16];
    __shared__ long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb = (int)job.ctr->n_blobs;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + tid;
        long long v = i < nb ? job.b_n[i] : 0;
        long long x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long long y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        long long pre = s_carry;
        for (int k = 0; k < wv; ++k) pre += s_w[k];
        if (i < nb) { offsets[i] = pre + x - v; cursor[i] = 0u; }
        __syncthreads();
        if (tid == 1023) s_carry = pre + x;
        __syncthreads();
    }
    if (tid == 0) { offsets[nb] = s_carry; job.ctr->n_voxels = (unsigned long long)s_carry; }
}

__global__ void __launch_bounds__(256) k_voxel_lists(Job job, const int64_t *__restrict__ offsets,
                                                     unsigned int *__restrict__ cursor, int32_t *__restrict__ crs_out) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t w = wave; w < job.total_words; w += n_waves) {
        const uint64_t m = job.mask[w];
        if (m == 0ull) continue;
        const bool bit = (m >> lane) & 1ull;
        const VolDesc vd = job.vols[find_vol(job.vols, job.n_vols, w)];
        const int64_t rem = w - vd.word_base;
        const int wq = (int)(rem % vd.row_words);
        const int64_t row = rem / vd.row_words;
        const uint64_t starts = run_starts(m);
        const int at_bit = bit ? lane : ctz64(m);        // (a lane without a voxel follows the word's first one: valid indices, nothing stored)
        const uint32_t run = job.run_base[w] + run_ordinal(starts, at_bit);
        const uint32_t comp = job.comp_of_run ? job.comp_of_run[run] : run;
        const uint32_t root = (uint32_t)job.parent[comp];
        const uint32_t rank = job.r_rank[root];
        // one cursor bump per RUN, or per word when all its voxels belong to one blob (a domain union is 10^5 voxels of a
        // single blob: a bump per voxel was 10^5 atomics on one address, 33 us of an aggregateCloud in round 4)
        const uint32_t rank0 = __shfl(rank, ctz64(m));
        const bool one_blob = __all(!bit || rank == rank0);
        const int first = one_blob ? ctz64(m) : run_start_of(m, at_bit);
        const uint32_t count = one_blob ? (uint32_t)popc64(m) : (uint32_t)(run_end_of(m, first) - first + 1);
        uint32_t at = 0;
        if (bit && lane == first) at = atomicAdd(&cursor[rank], count);
        at = __shfl(at, first);
        if (!bit) continue;
        const int64_t pos = offsets[rank] + at + (one_blob ? (uint32_t)popc64(m & bits_below(lane)) : (uint32_t)(lane - first));
        crs_out[3 * pos + 0] = vd.org[0] + wq * 64 + lane;
        crs_out[3 * pos + 1] = vd.org[1] + (int)(row % vd.dim[1]);
        crs_out[3 * pos + 2] = vd.org[2] + (int)(row / vd.dim[1]);
    }
}

// Several small result arrays packed into one block of a device staging buffer, so that ONE copy brings them to the host: every
// device -> host copy is a launch of the runtime's own on the stream (~5 us + the gap), and an accessor that returns seven columns
// of a blob table paid seven of them (round 4: 52 of an analysis entry's 114 launches were such copies).  Sizes and offsets are
// multiples of 4 bytes.
// k_copy_bytes: a small copy done by a kernel -- between device scratch and the context's PINNED staging block, which the device reads and
// writes over the link directly.  (A hipMemcpyAsync of a few hundred bytes is a packet on the SDMA engine: it waits behind every 4-8 MiB
// chunk the upload engine has queued there, and each hop between the compute queue and the copy engine is a semaphore.)
__global__ void __launch_bounds__(256) k_copy_bytes(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, unsigned long long bytes) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
        const unsigned long long n16 = bytes >> 4;
        const uint4 *s16 = reinterpret_cast<const uint4 *>(src);
        uint4 *d16 = reinterpret_cast<uint4 *>(dst);
        for (unsigned long long i = t; i < n16; i += stride) d16[i] = s16[i];
        for (unsigned long long i = (n16 << 4) + t; i < bytes; i += stride) dst[i] = src[i];
    } else {
        for (unsigned long long i = t; i < bytes; i += stride) dst[i] = src[i];
    }
}
// k_job_init: what stands in front of a grouped job's first kernel, in one launch -- the volume descriptors into the job's arena and zeroes over its
// counters, masks, first-key bitmap and rank counters (a copy and a fill before).  Both regions start 16-byte aligned; lengths in 16-byte units.
// in_*: the batch's staged inputs (pinned block -> device scratch) when their copy was left to this launch (group_setup's per-atom path, aggregateCloud's aux block).
__global__ void __launch_bounds__(256) k_job_init(const uint4 *__restrict__ in_src, uint4 *__restrict__ in_dst, unsigned long long in16,
                                                  const uint4 *__restrict__ vol_src, uint4 *__restrict__ vol_dst, unsigned long long vol16,
                                                  uint4 *__restrict__ zero, unsigned long long zero16) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long i = t; i < in16; i += stride) in_dst[i] = in_src[i];
    for (unsigned long long i = t; i < vol16; i += stride) vol_dst[i] = vol_src[i];
    const uint4 z = {0u, 0u, 0u, 0u};
    for (unsigned long long i = t; i < zero16; i += stride) zero[i] = z;
}
struct PackSeg { const uint32_t *src; unsigned long long words, dst_word; };
struct PackArgs { PackSeg seg[8]; int n; };
__global__ void __launch_bounds__(256) k_pack(PackArgs a, uint32_t *__restrict__ dst) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int s = 0; s < a.n; ++s) {
        const PackSeg sg = a.seg[s];
        for (unsigned long long i = t; i < sg.words; i += stride) dst[sg.dst_word + i] = sg.src[i];
    }
}

// ------------------------------------------------------------------------------------
// Sphere batches (getSphereCrsFromXyz, cutils.pyx:220-248).
// ------------------------------------------------------------------------------------
struct AtomBox {
    int32_t lo[3];
    int32_t hi[3];  // inclusive; hi < lo => empty
};

// Thread per atom: centre C = xyz2crs(xyz), R = xyz2crs(origin + r); box [C-R-1, C+R] (Q4).
__global__ void k_atom_boxes(const Geom *__restrict__ gp, const double *__restrict__ xyz, const float *__restrict__ radii,
                             const int32_t *__restrict__ atom_group, int64_t n_atoms, AtomBox *__restrict__ boxes,
                             int32_t *__restrict__ g_lo, int32_t *__restrict__ g_hi) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_atoms) return;
    const Geom &g = *gp;
    const double p[3] = {xyz[3 * a], xyz[3 * a + 1], xyz[3 * a + 2]};
    const double rad = (double)radii[a];
    int32_t C[3], R[3];
    xyz2crs(g, p, C);
    const double o[3] = {g.origin[0] + rad, g.origin[1] + rad, g.origin[2] + rad};
    xyz2crs(g, o, R);
    AtomBox bx;
    bool empty = false;
    for (int k = 0; k < 3; ++k) {
        bx.lo[k] = C[k] - R[k] - 1;
        bx.hi[k] = C[k] + R[k];
        empty = empty || (bx.hi[k] < bx.lo[k]);
    }
    if (empty) { for (int k = 0; k < 3; ++k) { bx.lo[k] = 0; bx.hi[k] = -1; } }
    boxes[a] = bx;
    if (!empty) {
        const int gidx = atom_group[a];
        for (int k = 0; k < 3; ++k) {
            atomicMin(&g_lo[3 * gidx + k], bx.lo[k]);
            atomicMax(&g_hi[3 * gidx + k], bx.hi[k]);
        }
    }
}

// Explicit voxels: group bounding boxes for list jobs.  A thread takes a contiguous share of the list; voxels arrive
// grouped (a domain union is a single group of 10^5 voxels), so a thread keeps the box of its current group in registers and
// sends it when the group changes; at the end a wave -- and then the block -- whose threads all hold the same group reduce
// first and send six atomics (a wave per 64 voxels sent 6 x 1000 atomics to the same six addresses: 33 us, round 4).
__global__ void __launch_bounds__(256) k_list_boxes(const int32_t *__restrict__ crs, const int32_t *__restrict__ vox_group, int64_t n,
                                                    int32_t *__restrict__ g_lo, int32_t *__restrict__ g_hi) {
    __shared__ int s_box[4][7];
    const int64_t n_threads = (int64_t)gridDim.x * blockDim.x, t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t share = (n + n_threads - 1) / n_threads, i0 = t * share, i1 = i0 + share < n ? i0 + share : n;
    int gidx = -1;
    int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (int64_t i = i0; i < i1; ++i) {
        const int gi = vox_group[i];
        if (gi != gidx) {
            if (gidx >= 0)
                for (int k = 0; k < 3; ++k) { atomicMin(&g_lo[3 * gidx + k], lo[k]); atomicMax(&g_hi[3 * gidx + k], hi[k]); }
            gidx = gi;
            for (int k = 0; k < 3; ++k) { lo[k] = INT32_MAX; hi[k] = INT32_MIN; }
        }
        for (int k = 0; k < 3; ++k) {
            const int v = crs[3 * i + k];
            lo[k] = v < lo[k] ? v : lo[k];
            hi[k] = v > hi[k] ? v : hi[k];
        }
    }
    // the wave: one group among the lanes that hold one?
    const unsigned long long have = __ballot(gidx >= 0);
    const int g0 = have ? __shfl(gidx, ctz64(have)) : -1;
    const bool wave_one = __all(gidx < 0 || gidx == g0);
    if (wave_one) {
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const int a = __shfl_xor(lo[k], d), c = __shfl_xor(hi[k], d);
                lo[k] = a < lo[k] ? a : lo[k];
                hi[k] = c > hi[k] ? c : hi[k];
            }
        }
    } else if (gidx >= 0) {
        for (int k = 0; k < 3; ++k) { atomicMin(&g_lo[3 * gidx + k], lo[k]); atomicMax(&g_hi[3 * gidx + k], hi[k]); }
    }
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 0) {
        s_box[wv][0] = wave_one ? g0 : -1;
        for (int k = 0; k < 3; ++k) { s_box[wv][1 + k] = lo[k]; s_box[wv][4 + k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // waves of one group each: merge neighbours of the same group, send the rest
        int cur = -1, clo[3] = {0, 0, 0}, chi[3] = {0, 0, 0};
        for (int q = 0; q <= 4; ++q) {
            const int gq = q < 4 ? s_box[q][0] : -1;
            if (q < 4 && gq >= 0 && gq == cur) {
                for (int k = 0; k < 3; ++k) { clo[k] = s_box[q][1 + k] < clo[k] ? s_box[q][1 + k] : clo[k]; chi[k] = s_box[q][4 + k] > chi[k] ? s_box[q][4 + k] : chi[k]; }
                continue;
            }
            if (cur >= 0)
                for (int k = 0; k < 3; ++k) { atomicMin(&g_lo[3 * cur + k], clo[k]); atomicMax(&g_hi[3 * cur + k], chi[k]); }
            cur = gq;
            if (q < 4 && gq >= 0)
                for (int k = 0; k < 3; ++k) { clo[k] = s_box[q][1 + k]; chi[k] = s_box[q][4 + k]; }
        }
    }
}

__global__ void k_init_bounds(int32_t *g_lo, int32_t *g_hi, int64_t n3, Counters *ctr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {   // (the batch's counters: k_make_vols fills the totals, raises the flag -- saves a fill launch)
        Counters c;
        memset(&c, 0, sizeof c);
        *ctr = c;
    }
    if (i >= n3) return;
    g_lo[i] = INT32_MAX;
    g_hi[i] = INT32_MIN;
}

// Single block: group bounding boxes -> volume descriptors with word / key offsets.
// cap_words / cap_keys: what the HOST sized the job for when it did not wait for these totals (group_setup: per-atom spheres, whose
// box sizes follow from the radius alone).  The two agree by construction (the same IEEE arithmetic on both sides); should they
// ever not, every volume is emptied before anything is painted and the flag makes the call fail -- nothing is written out of bounds.
__global__ void __launch_bounds__(1024) k_make_vols(const int32_t *__restrict__ g_lo, const int32_t *__restrict__ g_hi,
                                                     int n_groups, VolDesc *__restrict__ vols, Counters *__restrict__ ctr,
                                                     long long cap_words, long long cap_keys) {
    __shared__ long long s_w[16], s_k[16];
    __shared__ long long s_cw, s_ck;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_cw = 0; s_ck = 0; }
    __syncthreads();
    for (int base = 0; base < n_groups; base += 1024) {
        const int i = base + tid;
        VolDesc vd;
        long long words = 0, keys = 0;
        if (i < n_groups) {
            bool empty = false;
            for (int k = 0; k < 3; ++k) {
                int lo = g_lo[3 * i + k], hi = g_hi[3 * i + k];
                empty = empty || hi < lo;
                vd.org[k] = lo;
                vd.dim[k] = hi - lo + 1;
            }
            if (empty) { for (int k = 0; k < 3; ++k) { vd.org[k] = 0; vd.dim[k] = 0; } }
            vd.row_words = (vd.dim[0] + 63) / 64;
            vd.group = i;
            words = (long long)vd.row_words * vd.dim[1] * vd.dim[2];
            keys = (long long)vd.dim[0] * vd.dim[1] * vd.dim[2];
        }
        long long xw = words, xk = keys;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long long yw = __shfl_up(xw, d), yk = __shfl_up(xk, d);
            if (lane >= d) { xw += yw; xk += yk; }
        }
        if (lane == 63) { s_w[wv] = xw; s_k[wv] = xk; }
        __syncthreads();
        long long pw = s_cw, pk = s_ck;
        for (int k = 0; k < wv; ++k) { pw += s_w[k]; pk += s_k[k]; }
        if (i < n_groups) {
            vd.word_base = pw + xw - words;
            vd.key_base = pk + xk - keys;
            vols[i] = vd;
        }
        __syncthreads();
        if (tid == 1023) { s_cw = pw + xw; s_ck = pk + xk; }
        __syncthreads();
    }
    if (tid == 0) { ctr->total_words = s_cw; ctr->total_keys = s_ck; }
    if (s_cw > cap_words || s_ck > cap_keys) {   // block-uniform
        if (tid == 0) ctr->overflow = 1u;
        for (int i = tid; i < n_groups; i += 1024) {
            VolDesc vd = vols[i];
            vd.dim[0] = vd.dim[1] = vd.dim[2] = 0; vd.row_words = 0; vd.word_base = 0; vd.key_base = 0;
            vols[i] = vd;
        }
    }
}

// Block per atom, thread per 16-voxel piece of a box row: wrapped fetch, strict density filter (Q2), fp64 distance
// (cutils.pyx:205-218, 244-245); the hits of a piece are collected in a register and OR-ed into the group's volume with
// one atomic per mask word (a lane per voxel sent ~1400 atomics per atom to ~256 words), which deduplicates sphere unions
// on raw crs exactly like the reference's set (cutils.pyx:268-271).
__global__ void __launch_bounds__(256) k_sphere_paint(const Geom *__restrict__ gp, const float *__restrict__ dens,
                                                      const double *__restrict__ xyz, const float *__restrict__ radii,
                                                      const int32_t *__restrict__ atom_group, const AtomBox *__restrict__ boxes,
                                                      const VolDesc *__restrict__ vols, uint64_t *__restrict__ mask, float cutoff,
                                                      const Counters *__restrict__ setup_ctr, Counters *__restrict__ job_ctr) {
    if (setup_ctr->overflow != 0u) {   // (k_make_vols: the volumes outgrew what the host sized the job for -- cannot happen; fails the call)
        if (blockIdx.x == 0 && threadIdx.x == 0 && job_ctr) job_ctr->unit_wait_failed = 1u;
        return;
    }
    const int64_t a = blockIdx.x;
    const AtomBox bx = boxes[a];
    const int dc = bx.hi[0] - bx.lo[0] + 1, dr = bx.hi[1] - bx.lo[1] + 1, dsz = bx.hi[2] - bx.lo[2] + 1;
    if (dc <= 0 || dr <= 0 || dsz <= 0) return;
    const Geom &g = *gp;
    const VolDesc vd = vols[atom_group[a]];
    const double px = xyz[3 * a], py = xyz[3 * a + 1], pz = xyz[3 * a + 2];
    const double rad = (double)radii[a], cut = (double)cutoff;
    const unsigned pieces = ((unsigned)dc + 15u) / 16u, udr = (unsigned)dr;
    const int64_t n_items64 = (int64_t)pieces * dr * dsz;
    if (n_items64 >= (1ll << 31)) return;   // (unreachable: a box that large is rejected with its radius on the host side)
    const unsigned n_items = (unsigned)n_items64;
    for (unsigned it = threadIdx.x; it < n_items; it += blockDim.x) {
        const unsigned row = it / pieces, piece = it - row * pieces, sl = row / udr;
        const int r = bx.lo[1] + (int)(row - sl * udr), s = bx.lo[2] + (int)sl;
        const int c0 = bx.lo[0] + 16 * (int)piece;
        const int cnt = dc - 16 * (int)piece < 16 ? dc - 16 * (int)piece : 16;
        float dv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) dv[u] = u < cnt ? fetch_wrapped(g, dens, c0 + u, r, s) : 0.0f;
        const int lc0 = c0 - vd.org[0];
        const int64_t row_word = vd.word_base + ((int64_t)(s - vd.org[2]) * vd.dim[1] + (r - vd.org[1])) * vd.row_words;
        uint64_t bits = 0;
        int cur = lc0 >> 6;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u < cnt) {
                const int lc = lc0 + u;
                if ((lc >> 6) != cur) {
                    if (bits) atomicOr((unsigned long long *)&mask[row_word + cur], bits);
                    bits = 0;
                    cur = lc >> 6;
                }
                const double d = (double)dv[u];
                if ((0.0 < cut && cut < d) || (d < cut && cut < 0.0) || cut == 0.0) {
                    double q[3];
                    crs2xyz(g, c0 + u, r, s, q);
                    const double dx = q[0] - px, dy = q[1] - py, dz = q[2] - pz;
                    const double dist = __dsqrt_rn((dx * dx + dy * dy) + dz * dz);
                    if (dist <= rad) bits |= 1ull << (lc & 63);
                }
            }
        }
        if (bits) atomicOr((unsigned long long *)&mask[row_word + cur], bits);
    }
}

// Lock-free union in an LDS parent table (parent[x] <= x, roots point at themselves): find both roots, hang the larger
// under the smaller with an atomic min; if the larger was no root any more, carry on with its new parent.
// (find splits the path it walks: every node on it is re-pointed at its grandparent with a plain store.  A store that
//  overwrites a concurrent hook of a NON-root loses nothing: whoever hooks a non-root goes on to unite its old parent.)
__device__ __forceinline__ uint32_t lds_find(uint32_t *parent, uint32_t x) {
    uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (p != x) {
        const uint32_t gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void lds_unite(uint32_t *parent, uint32_t a, uint32_t b) {
    while (true) {
        a = lds_find(parent, a);
        b = lds_find(parent, b);
        if (a == b) return;
        const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
        const uint32_t old = atomicMin(&parent[hi], lo);
        if (old == hi) return;
        a = old;
        b = lo;
    }
}

// ------------------------------------------------------------------------------------
// Round 6: the whole labelling of a PER-ATOM sphere batch (the clouds of aggregateCloud, findAberrantBlobs on single atoms) in ONE
// launch -- k_sphere_paint, k_run_index, k_union, k_resolve and k_paint_keys were five, 5 us of device work each and a host
// launch apart.  An atom's volume is its own sphere box, nothing of the labelling crosses volumes, and a box is a few hundred
// mask words of one word a row: a workgroup per atom does the five phases on its volume with workgroup barriers where the
// generic kernels have launch boundaries.  The arithmetic is the generic kernels' own, statement by statement (the voxel test
// of k_sphere_paint, narrow_run_records, the neighbour rows of k_union, resolve_wave, paint_key): the results are the same
// integers.  Host side: group_setup knows the boxes (it made them) and takes this path when every box has one word a row and
// at most ATOM_WORDS rows; anything else goes through the five kernels as before.
//   * the masks live in LDS (and are written out for the accessors: voxel lists, pool look-ups);
//   * run ids: one atomicAdd on the job's counter per volume -- the runs of a volume are consecutive ids;
//   * a volume of at most ATOM_RUNS runs and ATOM_COMPS blobs (every atom cloud) never touches the global tables before its results
//     are final: the unions run on an LDS parent table, the runs' integer sums are folded into LDS accumulators per blob, and what
//     is written out is what k_resolve leaves -- a record per root, every run's root in parent[] -- plus the painted first keys.
//     (The first form of this kernel ran the five phases on the global tables with barriers in between: 140 us for 2 000 atoms where
//     the five kernels take 73 -- a chain of ~25 dependent memory round trips per workgroup, four waves each.)
//   * a larger volume runs those phases on the job's global tables as the generic kernels do.  A find may read a stale parent from
//     this CU's L1 (atomics execute in L2): harmless during the unions (uf_unite2 ends on the value its atomic min returns), but
//     resolve and the painting of the keys need what the atomics left -- an agent-scope acquire (L1 invalidate) follows the barrier
//     in front of either phase, which is what the launch boundaries did.
// ------------------------------------------------------------------------------------
constexpr int ATOM_WORDS = 512;    // rows of a box the fused kernel holds in LDS (a 3.5 A sphere at 0.5 A spacing: 16 x 16 = 256; at 0.35 A: 22 x 22)
constexpr int ATOM_RUNS = 1024;    // runs of a volume whose unions run in LDS ...
constexpr int ATOM_COMPS = 64;     // ... and blobs of a volume whose sums are folded in LDS (an atom has one to three clouds); beyond either: the global tables
                                   // (16.5 KB of LDS a workgroup: eight workgroups a CU, 2 048 atoms in one round)
// What a run [ra, rb] of the row (rl, sl) of volume vd contributes: the integers narrow_run_records / word_run_records_of store
// (sums rounded once to the job's quantum), computed with the same statements.  v: the row's densities by voxel (rows of at most
// NARROW_ROW voxels), or nullptr: fetched here.
struct RunRecord { uint32_t n; FixSums fs; long long c, r, s; unsigned long long key; };
__device__ __forceinline__ RunRecord atom_run_record(const Job &job, const Geom &g, const float *__restrict__ dens, const VolDesc &vd, int rl, int sl,
                                                     int ra, int rb, const float *v) {
    const int rawc0 = vd.org[0], rawr = vd.org[1] + rl, raws = vd.org[2] + sl, len = rb - ra + 1;
    double s_rho = 0.0, s_rl = 0.0;
    if (v) {
#pragma unroll
        for (int k = 0; k < NARROW_ROW; ++k)
            if (k >= ra && k <= rb) { s_rho += (double)v[k]; s_rl += (double)v[k] * (double)k; }
    } else {
        for (int k = ra; k <= rb; ++k) { const double x = (double)fetch_wrapped(g, dens, rawc0 + k, rawr, raws); s_rho += x; s_rl += x * (double)k; }
    }
    RunRecord rec;
    rec.n = (uint32_t)len;
    rec.fs = fix_sums(fix_of(s_rho, job.fix_mul), fix_of(s_rl, job.fix_mul), 0, 0, rawc0, rawr, raws);
    const long long first = (long long)rawc0 + ra;
    rec.c = (long long)len * first + (long long)len * (len - 1) / 2;
    rec.r = (long long)len * rawr;
    rec.s = (long long)len * raws;
    rec.key = (unsigned long long)(vd.key_base + ((int64_t)ra * vd.dim[1] + rl) * vd.dim[2] + sl);
    return rec;
}
__device__ __forceinline__ void atom_store_record(const Job &job, uint32_t idx, const RunRecord &rec) {
    job.parent[idx] = (int32_t)idx;
    job.r_n[idx] = rec.n;
    fix_store(job, idx, rec.fs);
    job.r_c[idx] = rec.c; job.r_r[idx] = rec.r; job.r_s[idx] = rec.s;
    job.r_key[idx] = rec.key;
}

__global__ void __launch_bounds__(256) k_atom_engine(Job job, const Geom *__restrict__ gp, const float *__restrict__ dens,
                                                     const double *__restrict__ xyz, const float *__restrict__ radii,
                                                     const AtomBox *__restrict__ boxes, float cutoff, int run_cap, int comp_cap) {   // (the caps: ATOM_RUNS / ATOM_COMPS; tests shrink them -- PDBEDA_DEBUG_ATOM_CAPS -- so that small inputs take the global-table path)
    __shared__ unsigned long long s_m[ATOM_WORDS];
    __shared__ uint32_t s_rb[ATOM_WORDS];     // first run of the row, counted from the volume's first run
    __shared__ uint32_t s_par[ATOM_RUNS];     // LDS path: parent of run k of the volume; after the numbering of the roots: component of the run
    __shared__ unsigned long long s_acc[10][ATOM_COMPS], s_key[ATOM_COMPS];   // LDS path: the seven FixSums fields, sum c / r / s; first key
    __shared__ uint32_t s_accn[ATOM_COMPS], s_root[ATOM_COMPS];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_carry, s_base, s_ncomp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t a = blockIdx.x;
    const AtomBox bx = boxes[a];
    const int dc = bx.hi[0] - bx.lo[0] + 1, dr = bx.hi[1] - bx.lo[1] + 1, dsz = bx.hi[2] - bx.lo[2] + 1;
    if (dc <= 0 || dr <= 0 || dsz <= 0) return;   // (block-uniform: an empty box has no words)
    const Geom &g = *gp;
    const VolDesc vd = job.vols[a];                // (a group per atom: the volume IS the box)
    const int W = dr * dsz;                        // one word a row (the host checked: dc <= 64, W <= ATOM_WORDS)
    for (int w = tid; w < W; w += 256) s_m[w] = 0ull;
    if (tid == 0) { s_carry = 0u; s_ncomp = 0u; }
    __syncthreads();
    {   // ---- paint (k_sphere_paint): a thread per 16-voxel piece of a box row ----
        const double px = xyz[3 * a], py = xyz[3 * a + 1], pz = xyz[3 * a + 2];
        const double rad = (double)radii[a], cut = (double)cutoff;
        const unsigned pieces = ((unsigned)dc + 15u) / 16u, n_items = pieces * (unsigned)W;
        for (unsigned it = tid; it < n_items; it += 256u) {
            const unsigned row = it / pieces, piece = it - row * pieces, sl = row / (unsigned)dr;
            const int r = bx.lo[1] + (int)(row - sl * (unsigned)dr), s = bx.lo[2] + (int)sl;
            const int c0 = bx.lo[0] + 16 * (int)piece;
            const int cnt = dc - 16 * (int)piece < 16 ? dc - 16 * (int)piece : 16;
            float dv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) dv[u] = u < cnt ? fetch_wrapped(g, dens, c0 + u, r, s) : 0.0f;
            unsigned long long bits = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u < cnt) {
                    const double d = (double)dv[u];
                    if ((0.0 < cut && cut < d) || (d < cut && cut < 0.0) || cut == 0.0) {
                        double q[3];
                        crs2xyz(g, c0 + u, r, s, q);
                        const double dx = q[0] - px, dy = q[1] - py, dz = q[2] - pz;
                        const double dist = __dsqrt_rn((dx * dx + dy * dy) + dz * dz);
                        if (dist <= rad) bits |= 1ull << (16 * (int)piece + u);
                    }
                }
            }
            if (bits) atomicOr(&s_m[row], bits);
        }
    }
    __syncthreads();
    // ---- run index (k_run_index): runs per row, a scan over the volume's rows, ONE id range for the volume ----
    for (int w0 = 0; w0 < W; w0 += 256) {   // (block-uniform trip count)
        const int w = w0 + tid;
        const uint32_t cnt = w < W ? (uint32_t)popc64(run_starts(s_m[w])) : 0u;
        uint32_t x = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wsum[wv] = x;
        __syncthreads();
        uint32_t pre = s_carry;
        for (int k = 0; k < wv; ++k) pre += s_wsum[k];
        if (w < W) s_rb[w] = pre + x - cnt;
        __syncthreads();
        if (tid == 255) s_carry = pre + x;
    }
    __syncthreads();
    const uint32_t T = s_carry;   // runs of the volume (block-uniform)
    // (the id range: one returning atomic per volume, in flight while the unions run in LDS)
    if (tid == 0) s_base = T ? atomicAdd(&job.ctr->n_runs, T) : 0u;
    if (T <= (uint32_t)run_cap) {
        // ---- LDS path: unions on an LDS parent table ----
        for (uint32_t i = tid; i < T; i += 256u) s_par[i] = i;
        __syncthreads();
        for (int t = tid; t < 4 * W; t += 256) {   // (k_union: a thread per (row, earlier neighbour row))
            const int w = t >> 2, nb = t & 3;
            const unsigned long long m = s_m[w];
            if (m == 0ull) continue;
            const int sl = w / dr, rl = w - sl * dr;
            const int r2 = rl + (nb == 3 ? 1 : (nb == 2 ? 0 : -1)), s2 = sl + (nb == 0 ? 0 : -1);
            if (r2 < 0 || r2 >= dr || s2 < 0) continue;
            const int w2 = s2 * dr + r2;
            const unsigned long long nm = s_m[w2];
            if (nm == 0ull) continue;
            uint64_t todo = run_starts(m);
            uint32_t k = 0;
            while (todo) {
                const int ra = ctz64(todo);
                todo &= todo - 1;
                const int rb = run_end_of(m, ra);
                const uint32_t me = s_rb[w] + k;
                ++k;
                int lo = ra - 1, hi = rb + 1;
                if (lo < 0) lo = 0;
                if (hi > 63) hi = 63;
                uint64_t hit = nm & (bits_below(hi + 1) & ~bits_below(lo));
                while (hit) {
                    const int p = ctz64(hit);
                    int st;
                    const uint32_t other = run_of_bit(nm, s_rb[w2], p, &st);
                    lds_unite(s_par, me, other);
                    hit &= ~bits_below(run_end_of(nm, st) + 1);
                }
            }
        }
        __syncthreads();
        // the roots take component numbers (the smallest run id of a blob is its root: parents point downwards)
        for (uint32_t i = tid; i < T; i += 256u)
            if (s_par[i] == i) { const uint32_t c = atomicAdd(&s_ncomp, 1u); if (c < (uint32_t)comp_cap) s_root[c] = i; }
        for (int c = tid; c < ATOM_COMPS; c += 256) {
#pragma unroll
            for (int f = 0; f < 10; ++f) s_acc[f][c] = 0ull;
            s_accn[c] = 0u; s_key[c] = ~0ull;
        }
        __syncthreads();
    }
    __syncthreads();   // (s_base: thread 0's atomic has returned)
    const uint32_t n_comp = s_ncomp;
    const uint32_t base = s_base;
    const bool in_lds = T <= (uint32_t)run_cap && n_comp <= (uint32_t)comp_cap;   // block-uniform
    if (in_lds) {
        // every run finds its root; the roots' places among the components go where the parents were
        uint32_t my_root[ATOM_RUNS / 256];
#pragma unroll
        for (int k = 0; k < ATOM_RUNS / 256; ++k) { const uint32_t i = (uint32_t)tid + 256u * k; my_root[k] = i < T ? lds_find(s_par, i) : 0u; }
        __syncthreads();
        for (uint32_t c = tid; c < n_comp; c += 256u) s_par[s_root[c]] = 0x80000000u | c;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ATOM_RUNS / 256; ++k) { const uint32_t i = (uint32_t)tid + 256u * k; if (i < T && !(s_par[i] & 0x80000000u)) s_par[i] = s_par[my_root[k]]; }
        __syncthreads();   // s_par[i] = 0x80000000 | component of run i
    }
    // ---- the runs' records: a thread per row.  LDS path: folded into the components' accumulators (integer adds: any order);
    //      otherwise stored run by run (the generic kernels' tables) ----
    for (int w = tid; w < W; w += 256) {
        const unsigned long long m = s_m[w];
        job.mask[vd.word_base + w] = m;
        job.run_base[vd.word_base + w] = base + s_rb[w];
        if (m == 0ull) continue;
        const int sl = w / dr, rl = w - sl * dr;
        float v[NARROW_ROW];
        const bool narrow = dc <= NARROW_ROW;
        if (narrow) {
#pragma unroll
            for (int k = 0; k < NARROW_ROW; ++k) v[k] = ((m >> k) & 1ull) ? fetch_wrapped(g, dens, vd.org[0] + k, vd.org[1] + rl, vd.org[2] + sl) : 0.0f;
        }
        uint64_t todo = run_starts(m);
        uint32_t k = s_rb[w];
        while (todo) {
            const int ra = ctz64(todo);
            todo &= todo - 1;
            const RunRecord rec = atom_run_record(job, g, dens, vd, rl, sl, ra, run_end_of(m, ra), narrow ? v : nullptr);
            if (in_lds) {
                const uint32_t c = s_par[k] & 0x7fffffffu;
                atomicAdd(&s_accn[c], rec.n);
                atomicAdd(&s_acc[0][c], (unsigned long long)rec.fs.rho);
                atomicAdd(&s_acc[1][c], (unsigned long long)rec.fs.c_lo); atomicAdd(&s_acc[2][c], (unsigned long long)rec.fs.c_hi);
                atomicAdd(&s_acc[3][c], (unsigned long long)rec.fs.r_lo); atomicAdd(&s_acc[4][c], (unsigned long long)rec.fs.r_hi);
                atomicAdd(&s_acc[5][c], (unsigned long long)rec.fs.s_lo); atomicAdd(&s_acc[6][c], (unsigned long long)rec.fs.s_hi);
                atomicAdd(&s_acc[7][c], (unsigned long long)rec.c); atomicAdd(&s_acc[8][c], (unsigned long long)rec.r); atomicAdd(&s_acc[9][c], (unsigned long long)rec.s);
                atomicMin(&s_key[c], rec.key);
                // (what the accessors read of a run: its root -- and a count that says the id is in use)
                job.parent[base + k] = (int32_t)(base + s_root[c]);
                job.r_n[base + k] = rec.n;
            } else {
                atom_store_record(job, base + k, rec);
            }
            ++k;
        }
    }
    __syncthreads();
    if (in_lds) {   // ---- one record per blob, its first key painted (k_resolve's result + k_paint_keys) ----
        for (uint32_t c = tid; c < n_comp; c += 256u) {
            const uint32_t id = base + s_root[c];
            FixSums fs;
            fs.rho = (long long)s_acc[0][c];
            fs.c_lo = (long long)s_acc[1][c]; fs.c_hi = (long long)s_acc[2][c];
            fs.r_lo = (long long)s_acc[3][c]; fs.r_hi = (long long)s_acc[4][c];
            fs.s_lo = (long long)s_acc[5][c]; fs.s_hi = (long long)s_acc[6][c];
            job.r_n[id] = s_accn[c];
            fix_store(job, id, fs);
            job.r_c[id] = (long long)s_acc[7][c]; job.r_r[id] = (long long)s_acc[8][c]; job.r_s[id] = (long long)s_acc[9][c];
            job.r_key[id] = s_key[c];
            paint_key(job, s_key[c]);
        }
        return;
    }
    // ---- a volume beyond the LDS tables: the generic kernels' phases on the job's global tables ----
    // (the records and parents of the volume are in L2 behind the barrier above: the unions' atomics find them)
    for (int t = tid; t < 4 * W; t += 256) {
        const int w = t >> 2, nb = t & 3;
        const unsigned long long m = s_m[w];
        if (m == 0ull) continue;
        const int sl = w / dr, rl = w - sl * dr;
        const int r2 = rl + (nb == 3 ? 1 : (nb == 2 ? 0 : -1)), s2 = sl + (nb == 0 ? 0 : -1);
        if (r2 < 0 || r2 >= dr || s2 < 0) continue;
        const int w2 = s2 * dr + r2;
        const unsigned long long nm = s_m[w2];
        if (nm == 0ull) continue;
        const uint32_t nbase = base + s_rb[w2];
        uint64_t todo = run_starts(m);
        uint32_t k = 0;
        while (todo) {
            const int ra = ctz64(todo);
            todo &= todo - 1;
            const int rb = run_end_of(m, ra);
            const int me = (int)(base + s_rb[w] + k);
            ++k;
            int lo = ra - 1, hi = rb + 1;
            if (lo < 0) lo = 0;
            if (hi > 63) hi = 63;
            uint64_t hit = nm & (bits_below(hi + 1) & ~bits_below(lo));
            while (hit) {
                const int p = ctz64(hit);
                int st;
                const uint32_t other = run_of_bit(nm, nbase, p, &st);
                uf_unite2(job.parent, me, (int)other);
                hit &= ~bits_below(run_end_of(nm, st) + 1);
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (see the head of this kernel: the finds below must see what the atomics left)
    for (uint32_t i0 = (uint32_t)(tid & ~63); i0 < T; i0 += 256u)   // (k_resolve; wave-uniform trip count)
        resolve_wave(job, base + i0 + (uint32_t)lane, i0 + (uint32_t)lane < T, lane);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint32_t i = (uint32_t)tid; i < T; i += 256u) {   // (k_paint_keys)
        const uint32_t id = base + i;
        if (job.parent[id] != (int32_t)id || job.r_n[id] == 0u) continue;
        paint_key(job, job.r_key[id]);
    }
}

__global__ void k_list_paint(const int32_t *__restrict__ crs, const int32_t *__restrict__ vox_group, int64_t n,
                             const VolDesc *__restrict__ vols, uint64_t *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const VolDesc vd = vols[vox_group[i]];
    const int lc = crs[3 * i] - vd.org[0], lr = crs[3 * i + 1] - vd.org[1], ls = crs[3 * i + 2] - vd.org[2];
    const int64_t w = vd.word_base + ((int64_t)ls * vd.dim[1] + lr) * vd.row_words + (lc >> 6);
    atomicOr((unsigned long long *)&mask[w], 1ull << (lc & 63));
}

// Regional sums over painted (cutoff-free) sphere-union volumes (densityAnalysis.py:1183-1198) + testValidXyzList
// (cutils.pyx:273-313).  Workgroup per VOLUME (= group: the painted box of an atom or of a residue's atoms), thread per box
// voxel in box order: an atom box is 16 voxels wide, so a lane per mask bit of one word would idle three lanes in four and
// walk 256 words in a row; here every lane tests its own voxel's bit, four voxels in flight per thread, partial sums in
// registers, ONE publish per volume.  The order of the fp64 additions is fixed (thread stride, shuffle tree, wave order).
__global__ void __launch_bounds__(256) k_region_reduce(const Geom *__restrict__ gp, const float *__restrict__ dens,
                                                       const VolDesc *__restrict__ vols, int n_vols, const uint64_t *__restrict__ mask,
                                                       int64_t total_words, float cutoff, double *__restrict__ pos,
                                                       double *__restrict__ neg, unsigned long long *__restrict__ cnt,
                                                       unsigned int *__restrict__ invalid) {
    __shared__ double s_p[4], s_q[4];
    __shared__ unsigned int s_n[4], s_bad[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const Geom &g = *gp;
    const double cut = (double)cutoff;
    for (int v = blockIdx.x; v < n_vols; v += gridDim.x) {
        const VolDesc vd = vols[v];
        const unsigned dc = (unsigned)vd.dim[0], dr = (unsigned)vd.dim[1];
        const unsigned nvox = dc * dr * (unsigned)vd.dim[2];      // (a volume is one group's bounding box: far below 2^32 voxels)
        double p = 0.0, q = 0.0;
        unsigned int n = 0;
        bool bad = false;
        for (unsigned i0 = tid; i0 < nvox; i0 += 1024) {
            unsigned c[4], row[4];
            uint64_t m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned i = i0 + 256u * u;
                row[u] = i / dc;
                c[u] = i - row[u] * dc;
                m[u] = i < nvox ? mask[vd.word_base + (int64_t)row[u] * vd.row_words + (c[u] >> 6)] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!((m[u] >> (c[u] & 63)) & 1ull)) continue;
                const unsigned sl = row[u] / dr, rl = row[u] - sl * dr;
                bool ok = true;
                const double d = (double)fetch_wrapped(g, dens, vd.org[0] + (int)c[u], vd.org[1] + (int)rl, vd.org[2] + (int)sl, &ok);
                ++n;
                if (d > cut) p += d;
                if (d < -cut) q += d;
                bad = bad || !ok;
            }
        }
        const unsigned long long any_bad = __ballot(bad);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            p += __shfl_down(p, off);
            q += __shfl_down(q, off);
            n += __shfl_down(n, off);
        }
        if (lane == 0) { s_p[wv] = p; s_q[wv] = q; s_n[wv] = n; s_bad[wv] = any_bad ? 1u : 0u; }
        __syncthreads();
        if (tid == 0) {
            const double tp = ((s_p[0] + s_p[1]) + s_p[2]) + s_p[3], tq = ((s_q[0] + s_q[1]) + s_q[2]) + s_q[3];
            const unsigned int tn = s_n[0] + s_n[1] + s_n[2] + s_n[3];
            if (tp != 0.0) unsafeAtomicAdd(&pos[vd.group], tp);
            if (tq != 0.0) unsafeAtomicAdd(&neg[vd.group], tq);
            if (tn) atomicAdd(&cnt[vd.group], (unsigned long long)tn);
            if (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) atomicOr(&invalid[vd.group], 1u);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// aggregateCloud (densityAnalysis.py:571-731) on device-resident voxel lists: the clouds of the atoms stay where the
// sphere batch left them (voxel list grouped by cloud); pooled clouds are gathered straight into a second job whose
// groups are the residues (all-pairs testOverlap clustering + merge == 26-connected components of the union of the
// pooled voxels, 663-687) plus ONE extra group holding everything pooled (the domain clouds, 692-712).
// ------------------------------------------------------------------------------------
// Thread per gathered voxel: item i < V goes to the residue group of its cloud's atom, item V + i to the domain group.
// Round 6: the regional sums of a PER-ATOM batch (calculateAtomRegionDensity / Discrepancy: a group per atom) in ONE launch.  A group of one atom needs
// no mask -- nothing is united, no voxel is shared -- so the sphere test of k_sphere_paint (cutoff 0: the distance alone) and the sums of k_region_reduce
// run in one pass over the atom's box: the same voxel -> thread mapping, the same order of the fp64 additions, the same reduction tree, hence the same
// sums to the last bit (PDBEDA_ATOM_REGION=0: the four launches, A/B).  Inputs (coordinates, radii, the host-made volumes) are read where the host staged
// them, in the pinned block -- a workgroup reads ITS atom's 80 bytes over the link once --, results are written there: no input copy, no zeroing, no pack.
__global__ void __launch_bounds__(256) k_atom_region(const Geom *__restrict__ gp, const float *__restrict__ dens, const double *__restrict__ xyz,
                                                     const float *__restrict__ radii, const VolDesc *__restrict__ vols, int n_vols, float cutoff,
                                                     double *__restrict__ pos, double *__restrict__ neg, unsigned long long *__restrict__ cnt,
                                                     unsigned int *__restrict__ invalid) {
    __shared__ double s_p[4], s_q[4];
    __shared__ unsigned int s_n[4], s_bad[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const Geom &g = *gp;
    const double cut = (double)cutoff;
    for (int v = blockIdx.x; v < n_vols; v += gridDim.x) {
        const VolDesc vd = vols[v];
        const double px = xyz[3 * v], py = xyz[3 * v + 1], pz = xyz[3 * v + 2], rad = (double)radii[v];
        const unsigned dc = (unsigned)vd.dim[0], dr = (unsigned)vd.dim[1];
        const unsigned nvox = dc * dr * (unsigned)vd.dim[2];
        double p = 0.0, q = 0.0;
        unsigned int n = 0;
        bool bad = false;
        for (unsigned i0 = tid; i0 < nvox; i0 += 1024) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned i = i0 + 256u * u;
                if (i >= nvox) continue;
                const unsigned row = i / dc, c = i - row * dc, sl = row / dr, rl = row - sl * dr;
                const int rc = vd.org[0] + (int)c, rr = vd.org[1] + (int)rl, rs = vd.org[2] + (int)sl;
                double w[3];
                crs2xyz(g, rc, rr, rs, w);
                const double dx = w[0] - px, dy = w[1] - py, dz = w[2] - pz;
                if (!(__dsqrt_rn((dx * dx + dy * dy) + dz * dz) <= rad)) continue;      // (k_sphere_paint at cutoff 0: inside the sphere)
                bool ok = true;
                const double d = (double)fetch_wrapped(g, dens, rc, rr, rs, &ok);
                ++n;
                if (d > cut) p += d;
                if (d < -cut) q += d;
                bad = bad || !ok;
            }
        }
        const unsigned long long any_bad = __ballot(bad);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            p += __shfl_down(p, off);
            q += __shfl_down(q, off);
            n += __shfl_down(n, off);
        }
        if (lane == 0) { s_p[wv] = p; s_q[wv] = q; s_n[wv] = n; s_bad[wv] = any_bad ? 1u : 0u; }
        __syncthreads();
        if (tid == 0) {
            pos[v] = ((s_p[0] + s_p[1]) + s_p[2]) + s_p[3];
            neg[v] = ((s_q[0] + s_q[1]) + s_q[2]) + s_q[3];
            cnt[v] = (unsigned long long)(s_n[0] + s_n[1] + s_n[2] + s_n[3]);
            invalid[v] = s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_pool_gather(const int32_t *__restrict__ src_crs, const int64_t *__restrict__ src_off,
                                                     const int32_t *__restrict__ pool_cloud, const int64_t *__restrict__ pool_voff,
                                                     const int32_t *__restrict__ pool_group, int n_pool, int64_t V, int domain_group,
                                                     int32_t *__restrict__ out_crs, int32_t *__restrict__ out_group) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * V) return;
    const bool dom = i >= V;
    const int64_t j = dom ? i - V : i;
    int lo = 0, hi = n_pool - 1;   // the pooled cloud that holds voxel j: pool_voff[p] <= j < pool_voff[p + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pool_voff[mid] <= j) lo = mid; else hi = mid - 1;
    }
    const int64_t src = src_off[pool_cloud[lo]] + (j - pool_voff[lo]);
    out_crs[3 * i] = src_crs[3 * src];
    out_crs[3 * i + 1] = src_crs[3 * src + 1];
    out_crs[3 * i + 2] = src_crs[3 * src + 2];
    out_group[i] = dom ? domain_group : pool_group[lo];
}

// Thread per (pooled cloud, union kind): the rank (row of the union job's blob table) of the component that contains the
// cloud -- looked up through its first voxel.  out[kind * n_pool + p].
__global__ void __launch_bounds__(256) k_pool_component(Job job, const int32_t *__restrict__ src_crs, const int64_t *__restrict__ src_off,
                                                        const int32_t *__restrict__ pool_cloud, const int32_t *__restrict__ pool_group, int n_pool,
                                                        int domain_group, int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n_pool) return;
    const int kind = i >= n_pool ? 1 : 0, p = kind ? i - n_pool : i;
    const int64_t v = src_off[pool_cloud[p]];
    const VolDesc vd = job.vols[kind ? domain_group : pool_group[p]];
    const int lc = src_crs[3 * v] - vd.org[0], lr = src_crs[3 * v + 1] - vd.org[1], ls = src_crs[3 * v + 2] - vd.org[2];
    const int64_t w = vd.word_base + ((int64_t)ls * vd.dim[1] + lr) * vd.row_words + (lc >> 6);
    const uint32_t run = job.run_base[w] + run_ordinal(run_starts(job.mask[w]), lc & 63);
    const uint32_t comp = job.comp_of_run ? job.comp_of_run[run] : run;
    out[i] = (int32_t)job.r_rank[(uint32_t)job.parent[comp]];
}

// The last launch of aggregateCloud's union job in its UNORDERED form (round 6; k_paint_keys, k_emit, k_pool_component and the pack of
// the results were four): behind k_resolve(slots) every root holds its blob's final sums and a table row of its own, so ONE grid
//   * writes the roots' rows -- voxels, total density, density-weighted centroid (emit_row_ranked's arithmetic), group --,
//   * looks up the row of the component that holds each pooled cloud, per union kind (k_pool_component),
//   * copies the bonded pairs' touch flags and the job's counters
// STRAIGHT into the context's pinned block, where the host reads them after its one wait for this job.
struct UnionFinish {
    const int32_t *src_crs; const int64_t *src_off;                 // the clouds' voxel lists
    const int32_t *pool_cloud, *pool_group; int n_pool, domain_group;
    const unsigned int *touch; int n_pairs;
    Counters *out_ctr; long long *out_n; double *out_total, *out_centroid; int32_t *out_group; unsigned int *out_touch; int32_t *out_comp;
    unsigned int cap;                                               // rows the host's arrays hold (a union component holds a pooled cloud: 2 n_pool bound them)
};
__global__ void __launch_bounds__(256) k_union_finish(Job job, const Geom *__restrict__ gp, UnionFinish a) {
    const uint32_t n_runs = n_components(job);
    const uint32_t stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    const Geom &g = *gp;
    for (uint32_t i = t0; i < n_runs; i += stride) {
        if (job.parent[i] != (int32_t)i) continue;
        const uint32_t row = job.r_rank[i];
        if (row >= a.cap) continue;      // (cannot happen: see cap)
        const uint32_t n_vox = job.r_n[i];
        const FixSums fs = fix_load(job, i);
        const double tot_q = (double)fs.rho, rc = fix_moment(fs.c_lo, fs.c_hi), rr = fix_moment(fs.r_lo, fs.r_hi), rs = fix_moment(fs.s_lo, fs.s_hi);
        const double wc[3] = {rc / tot_q, rr / tot_q, rs / tot_q};   // (the quantum cancels)
        double xyz[3];
        crs2xyz_frac(g, wc, xyz);
        a.out_n[row] = (long long)n_vox;
        a.out_total[row] = tot_q / job.fix_mul;      // (a power of two: exact)
        a.out_centroid[3 * row + 0] = xyz[0]; a.out_centroid[3 * row + 1] = xyz[1]; a.out_centroid[3 * row + 2] = xyz[2];
        a.out_group[row] = job.vols[find_vol_by_key(job.vols, job.n_vols, (int64_t)job.r_key[i])].group;
    }
    for (uint32_t i = t0; i < 2u * (uint32_t)a.n_pool; i += stride) {
        const int kind = i >= (uint32_t)a.n_pool ? 1 : 0, p = kind ? (int)i - a.n_pool : (int)i;
        const int64_t v = a.src_off[a.pool_cloud[p]];
        const VolDesc vd = job.vols[kind ? a.domain_group : a.pool_group[p]];
        const int lc = a.src_crs[3 * v] - vd.org[0], lr = a.src_crs[3 * v + 1] - vd.org[1], ls = a.src_crs[3 * v + 2] - vd.org[2];
        const int64_t w = vd.word_base + ((int64_t)ls * vd.dim[1] + lr) * vd.row_words + (lc >> 6);
        const uint32_t run = job.run_base[w] + run_ordinal(run_starts(job.mask[w]), lc & 63);
        a.out_comp[i] = (int32_t)job.r_rank[(uint32_t)job.parent[run]];
    }
    for (uint32_t i = t0; i < (uint32_t)a.n_pairs; i += stride) a.out_touch[i] = a.touch[i];
    if (t0 == 0) *a.out_ctr = *job.ctr;
}

// One 16-bit digit of a radix select over the masked voxels of the unique box (see pdbeda_abs_select_hist).
__global__ void __launch_bounds__(256) k_abs_select_hist(const Geom *__restrict__ gp, const float *__restrict__ a, const float *__restrict__ b,
                                                         double alpha, double cut_a, double cut_b, int which, int shift,
                                                         unsigned long long prefix, unsigned long long prefix_mask, unsigned int *__restrict__ hist) {
    const Geom &g = *gp;
    const int uc = g.unique_ncrs[0], ur = g.unique_ncrs[1], us = g.unique_ncrs[2];
    const int64_t n = (int64_t)uc * ur * us;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % uc);
        const int64_t row = i / uc;
        const int64_t at = ((row / ur) * g.ncrs[1] + (row % ur)) * g.ncrs[0] + c;
        const float va = a[at];
        if (!(fabs((double)va) < cut_a)) continue;
        double vc = 0.0;
        if (b) {
            vc = fabs((double)va + alpha * (double)b[at]);
            if (!(vc < cut_b)) continue;
        }
        const unsigned long long key = which ? (unsigned long long)__double_as_longlong(vc) : (unsigned long long)__float_as_uint(fabsf(va));
        if ((key & prefix_mask) != prefix) continue;
        atomicAdd(&hist[(key >> shift) & 0xffffull], 1u);
    }
}

// out = float32(double(a) + alpha * double(b)), float4 per thread (the Fc map: alpha = -2).
// In-place 32-bit byte swap of an uploaded grid (a CCP4 file written on a machine of the other endianness).
__global__ void __launch_bounds__(256) k_byteswap32(uint32_t *__restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = __builtin_bswap32(x[i]);
}

__global__ void __launch_bounds__(256) k_map_combine(const float *__restrict__ a, const float *__restrict__ b, double alpha, int64_t n,
                                                     float *__restrict__ out) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
        float4 r;
        r.x = (float)((double)x.x + alpha * (double)y.x);
        r.y = (float)((double)x.y + alpha * (double)y.y);
        r.z = (float)((double)x.z + alpha * (double)y.z);
        r.w = (float)((double)x.w + alpha * (double)y.w);
        reinterpret_cast<float4 *>(out)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        out[i] = (float)((double)a[i] + alpha * (double)b[i]);
    }
}

// ------------------------------------------------------------------------------------
// Whole-map reductions.  Deterministic two-stage fp64 sums: a fixed grid writes one
// partial per block, a single block folds the partials in index order.
// mode 0: sum(x)   mode 1: sum((x-mean)^2)   mode 2: sum(|x|) for |x| > cutoff (strict)
// ------------------------------------------------------------------------------------
__device__ inline double block_sum(double v, double *s_part) {
    const int lane = lane_id(), wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) s_part[wv] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += s_part[k];
    return t;
}

__global__ void __launch_bounds__(256) k_reduce_partials(const float *__restrict__ x, int64_t n, int mode, const double *__restrict__ mean_p,
                                                         double cutoff, double *__restrict__ partials) {
    __shared__ double s_part[4];
    const int64_t n4 = n >> 2;
    const double mean = mode == 1 ? mean_p[0] : 0.0;
    double acc = 0.0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        const double e[4] = {(double)v.x, (double)v.y, (double)v.z, (double)v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (mode == 0) acc += e[k];
            else if (mode == 1) { const double d = e[k] - mean; acc += d * d; }
            else { const double a = fabs(e[k]); if (a > cutoff) acc += a; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const double e = (double)x[(n4 << 2) + threadIdx.x];
        if (mode == 0) acc += e;
        else if (mode == 1) { const double d = e - mean; acc += d * d; }
        else { const double a = fabs(e); if (a > cutoff) acc += a; }
    }
    const double t = block_sum(acc, s_part);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// The range of a map in ONE pass: per block sum |x| (fixed order inside the block) and max |x|; k_range_final folds the
// blocks' partials in index order.  out[0] = sum |x|, out[1] = max |x| (NaNs are not magnitudes).
__global__ void __launch_bounds__(256) k_range_partials(const float *__restrict__ x, int64_t n, double *__restrict__ part_sum, double *__restrict__ part_max) {
    __shared__ double s_part[4];
    __shared__ float s_max[4];
    double acc = 0.0;
    float mx = 0.0f;
    const int64_t n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    // (a NaN or an infinity makes the maximum infinite: the host refuses to derive a quantum from such a map)
    auto take = [&](float v) { const float a = fabsf(v); if (a < INFINITY) { acc += (double)a; mx = a > mx ? a : mx; } else mx = INFINITY; };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        take(v.x); take(v.y); take(v.z); take(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) take(x[(n4 << 2) + threadIdx.x]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_down(mx, off); mx = o > mx ? o : mx; }
    if (lane_id() == 0) s_max[threadIdx.x >> 6] = mx;
    const double t = block_sum(acc, s_part);   // (its barrier also publishes s_max)
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = t;
        part_max[blockIdx.x] = (double)fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    }
}
__global__ void __launch_bounds__(256) k_range_final(const double *__restrict__ part_sum, const double *__restrict__ part_max, int n_part, double *__restrict__ out) {
    __shared__ double s_part[4];
    __shared__ double s_max[4];
    double acc = 0.0, mx = 0.0;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) { acc += part_sum[i]; mx = part_max[i] > mx ? part_max[i] : mx; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(mx, off); mx = o > mx ? o : mx; }
    if (lane_id() == 0) s_max[threadIdx.x >> 6] = mx;
    const double t = block_sum(acc, s_part);
    if (threadIdx.x == 0) { out[0] = t; out[1] = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3])); }
}

// Single block: out[0] = sum(partials) [/ n]  [sqrt].
__global__ void __launch_bounds__(256) k_reduce_final(const double *__restrict__ partials, int n_part, double scale, int take_sqrt,
                                                      double *__restrict__ out) {
    __shared__ double s_part[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) acc += partials[i];
    const double t = block_sum(acc, s_part);
    if (threadIdx.x == 0) {
        double v = t * scale;
        out[0] = take_sqrt ? __dsqrt_rn(v) : v;
    }
}

// ------------------------------------------------------------------------------------
// meanDensity / stdDensity exactly as numpy computes them (ccp4.py:343-363: np.mean / np.std of the voxel tuple, i.e. of a
// contiguous float64 array).  The default cutoffs (densityAnalysis.py:131-132,148) are float32(mean + k std): one ulp of
// the mean can flip a voxel, so the summation TREE is reproduced, not just the value to 1e-12:
//   add.reduce feeds the inner loop 8192 elements at a time and adds the calls up in order; each call is numpy's
//   pairwise sum: blocks of 128 summed with 8 interleaved accumulators, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), blocks
//   combined by a balanced binary tree (8192 = 64 leaves); an array tail < 8192 follows the general recursion
//   n2 = (n/2) & ~7.  np.std: m = sum/n; x = a - m; x = x*x; sqrt(sum(x)/n)  (_methods._var).
// k_np_chunk_sums: one workgroup per 8192-element chunk: coalesced 16-B loads -> LDS (leaf stride 136 floats: the
// 8 x 4 (lane j, leaf) reads of a 32-lane group hit 32 different banks), thread (leaf, j) adds its 16 values in order,
// wave butterflies build the tree.  k_np_final: the tail chunk (its recursion tree as a heap in LDS, built and combined level by level
// by the workgroup) + the in-order accumulation of the chunk sums (one thread: the order IS the result), then the division / square root.
// mode 0: x   mode 1: (x - mean)^2
// ------------------------------------------------------------------------------------
constexpr int NP_CHUNK = 8192, NP_LSTRIDE = 136;

__device__ __forceinline__ double np_elem(float v, int mode, double shift) {
    double d = (double)v;
    if (mode == 1) { d = d - shift; d = d * d; }
    return d;
}

// range_sum / range_max (round 6; nullable, mode 0): per chunk, sum |x| and max |x| of the same 8192 values -- the range pass of a fresh map (the quantum
// of its blob sums) rides in the mean's pass over the map instead of a pass and two launches of its own (k_range_partials, k_range_final)
__global__ void __launch_bounds__(256) k_np_chunk_sums(const float *__restrict__ x, int64_t n_full, int mode, const double *__restrict__ mean_p,
                                                       double *__restrict__ chunk_sums, double *__restrict__ range_sum, double *__restrict__ range_max,
                                                       Geom geom_val, Geom *__restrict__ geom_out) {   // geom_out (nullable): the map's geometry, by value -- a copy launch of its own before
    __shared__ __attribute__((aligned(16))) float s_x[64 * NP_LSTRIDE];
    __shared__ double s_node[8];
    __shared__ double s_rs[4];
    __shared__ float s_rm[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double shift = mode == 1 ? mean_p[0] : 0.0;
    if (geom_out && blockIdx.x == 0 && tid == 0) *geom_out = geom_val;      // (nothing of this chain reads it; the jobs behind it on the stream do)
    for (int64_t chunk = blockIdx.x; chunk < n_full; chunk += gridDim.x) {   // block-uniform
        const float4 *src = reinterpret_cast<const float4 *>(x + chunk * NP_CHUNK);
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[tid + 256 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = 4 * (tid + 256 * k);
            *reinterpret_cast<float4 *>(&s_x[(idx >> 7) * NP_LSTRIDE + (idx & 127)]) = v[k];
        }
        if (range_sum) {   // (block-uniform; a fixed order: thread, then the shuffle tree, then the waves in order)
            double acc = 0.0;
            float mx = 0.0f;
            auto take = [&](float q) { const float a = fabsf(q); if (a < INFINITY) { acc += (double)a; mx = a > mx ? a : mx; } else mx = INFINITY; };   // (a NaN or an infinity makes the maximum infinite: the host refuses such a map)
#pragma unroll
            for (int k = 0; k < 8; ++k) { take(v[k].x); take(v[k].y); take(v[k].z); take(v[k].w); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { acc += __shfl_down(acc, off); const float o = __shfl_down(mx, off); mx = o > mx ? o : mx; }
            if (lane == 0) { s_rs[wv] = acc; s_rm[wv] = mx; }
        }
        __syncthreads();
        if (range_sum && tid == 0) {
            range_sum[chunk] = ((s_rs[0] + s_rs[1]) + s_rs[2]) + s_rs[3];
            range_max[chunk] = (double)fmaxf(fmaxf(s_rm[0], s_rm[1]), fmaxf(s_rm[2], s_rm[3]));
        }
        double node[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float *p = s_x + (h * 32 + (tid >> 3)) * NP_LSTRIDE + (tid & 7);
            double r = np_elem(p[0], mode, shift);
#pragma unroll
            for (int i = 1; i < 16; ++i) r += np_elem(p[8 * i], mode, shift);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) r += __shfl_xor(r, d);   // 8 accumulators -> leaf -> the 8 leaves of this wave
            node[h] = r;
        }
        if (lane == 0) { s_node[wv] = node[0]; s_node[4 + wv] = node[1]; }
        __syncthreads();
        if (tid == 0) chunk_sums[chunk] = ((s_node[0] + s_node[1]) + (s_node[2] + s_node[3])) + ((s_node[4] + s_node[5]) + (s_node[6] + s_node[7]));
    }
}

// range_sum / range_max / range_out (round 6; nullable, mode 0): the chunks' range partials folded in index order, the tail's added -> range_out[0] = sum |x|,
// range_out[1] = max |x|.  host_out (nullable): six doubles -- mean_p[0] or this launch's result, this launch's result, -, -, and the range (this launch's, or range_in) -- written
// straight into the context's pinned block by the LAST launch of the chain (a copy launch of its own before).
__global__ void __launch_bounds__(256) k_np_final(const float *__restrict__ x, int64_t n, int64_t n_full, int mode, const double *__restrict__ mean_p,
                                                  const double *__restrict__ chunk_sums, int take_sqrt, double *__restrict__ out,
                                                  const double *__restrict__ range_sum, const double *__restrict__ range_max, double *__restrict__ range_out,
                                                  const double *__restrict__ range_in, double *__restrict__ host_out) {
    // The tail's recursion tree as a heap in LDS (node i -> 2i, 2i + 1): a right child is at most l / 2 + 7.5 long, so from 8 191 elements
    // the lengths are <= 4 103, 2 059, 1 037, 526, 270, 142, 78: every node of level 7 (ids 128..255) is a leaf.  Built level by level, leaves
    // summed by eight lanes each, combined level by level (left + right): round 5 -- one thread walking explicit stacks (private arrays =
    // scratch memory, then LDS) was 60 of this kernel's 80 us behind a 200^3 upload.
    __shared__ int s_off[256], s_len[256];
    __shared__ double s_val[256];
    __shared__ double s_buf[1024];
    __shared__ float s_tail[NP_CHUNK];
    const int tid = threadIdx.x;
    const double shift = mode == 1 ? mean_p[0] : 0.0;
    const int rem = (int)(n - n_full * NP_CHUNK);
    const float *xg = x + n_full * NP_CHUNK;
    for (int i = tid; i < rem; i += 256) s_tail[i] = xg[i];
    s_len[tid] = tid == 1 ? rem : 0;
    s_off[tid] = 0;
    __syncthreads();
    for (int d = 0; d < 7; ++d) {
        const int i = (1 << d) + tid;
        if (tid < (1 << d) && s_len[i] > 128) {
            const int l = s_len[i], o = s_off[i];
            int n2 = l / 2;
            n2 -= n2 % 8;
            s_off[2 * i] = o; s_len[2 * i] = n2;
            s_off[2 * i + 1] = o + n2; s_len[2 * i + 1] = l - n2;
        }
        __syncthreads();
    }
    for (int task = tid; task < 256 * 8; task += 256) {   // (node, accumulator j): the 8 lanes of a leaf are neighbours
        const int node = task >> 3, j = task & 7, o = s_off[node], l = s_len[node];
        const bool leaf = l > 0 && l <= 128;               // (uniform over the 8 lanes of a node)
        double r = 0.0;
        if (leaf && l >= 8) {
            r = np_elem(s_tail[o + j], mode, shift);
            for (int i = 8; i < l - (l % 8); i += 8) r += np_elem(s_tail[o + i + j], mode, shift);
        }
        r += __shfl_xor(r, 1);
        r += __shfl_xor(r, 2);
        r += __shfl_xor(r, 4);
        if (leaf && j == 0) {
            double res = l >= 8 ? r : 0.0;
            for (int i = l >= 8 ? l - (l % 8) : 0; i < l; ++i) res += np_elem(s_tail[o + i], mode, shift);
            s_val[node] = res;
        }
    }
    __syncthreads();
    for (int d = 6; d >= 0; --d) {
        const int i = (1 << d) + tid;
        if (tid < (1 << d) && s_len[i] > 128) s_val[i] = s_val[2 * i] + s_val[2 * i + 1];
        __syncthreads();
    }
    double total = 0.0;
    // ---- chunk sums in order (staged through LDS, added by ONE thread: the order IS the result) ----
    for (int64_t base = 0; base < n_full; base += 1024) {
        for (int k = tid; k < 1024; k += 256) s_buf[k] = base + k < n_full ? chunk_sums[base + k] : 0.0;
        __syncthreads();
        if (tid == 0) {
            const int cnt = n_full - base < 1024 ? (int)(n_full - base) : 1024;
            int k = 0;
            for (; k + 8 <= cnt; k += 8) {    // eight loads in flight, then the chain of adds
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = s_buf[k + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) total += v[q];
            }
            for (; k < cnt; ++k) total += s_buf[k];
        }
        __syncthreads();
    }
    double r_total = 0.0, r_max = 0.0;
    if (range_out) {   // (block-uniform) the range: the tail's magnitudes by a fixed tree, the chunks' partials in index order behind them
        __shared__ double s_rpart[4], s_rmax[4];
        double acc = 0.0, mx = 0.0;
        for (int i = tid; i < rem; i += 256) { const float a = fabsf(s_tail[i]); if (a < INFINITY) { acc += (double)a; mx = (double)a > mx ? (double)a : mx; } else mx = INFINITY; }
        if (range_max)
            for (int64_t k = tid; k < n_full; k += 256) { const double q = range_max[k]; mx = q > mx ? q : mx; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(mx, off); mx = o > mx ? o : mx; }
        if ((tid & 63) == 0) s_rmax[tid >> 6] = mx;
        const double tail_sum = block_sum(acc, s_rpart);   // (its barrier also publishes s_rmax)
        if (tid == 0) {
            r_total = tail_sum;
            if (range_sum)
                for (int64_t k = 0; k < n_full; ++k) r_total += range_sum[k];
            r_max = fmax(fmax(s_rmax[0], s_rmax[1]), fmax(s_rmax[2], s_rmax[3]));
            range_out[0] = r_total; range_out[1] = r_max;
        }
    }
    if (tid == 0) {
        if (rem > 0) total += s_val[1];
        const double v = total / (double)n;
        const double res = take_sqrt ? __dsqrt_rn(v) : v;
        out[0] = res;
        if (host_out) {
            host_out[0] = mode == 1 ? mean_p[0] : res;
            host_out[1] = res;
            host_out[4] = range_out ? r_total : (range_in ? range_in[0] : 0.0);      // (range_in: what an earlier launch of the chain left in range_out)
            host_out[5] = range_out ? r_max : (range_in ? range_in[1] : 0.0);
        }
    }
}

// ------------------------------------------------------------------------------------
// Batched point helpers.
// ------------------------------------------------------------------------------------
__global__ void k_point_density(const Geom *__restrict__ gp, const float *__restrict__ dens, const int32_t *__restrict__ crs, int64_t n,
                                double *__restrict__ out, uint8_t *__restrict__ valid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool ok;
    const float v = fetch_wrapped(*gp, dens, crs[3 * i], crs[3 * i + 1], crs[3 * i + 2], &ok);
    if (out) out[i] = (double)v;
    if (valid) valid[i] = ok ? 1 : 0;
}

__global__ void k_crs2xyz(const Geom *__restrict__ gp, const int32_t *__restrict__ crs, int64_t n, double *__restrict__ xyz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double q[3];
    crs2xyz(*gp, crs[3 * i], crs[3 * i + 1], crs[3 * i + 2], q);
    xyz[3 * i] = q[0]; xyz[3 * i + 1] = q[1]; xyz[3 * i + 2] = q[2];
}

__global__ void k_xyz2crs(const Geom *__restrict__ gp, const double *__restrict__ xyz, int64_t n, int32_t *__restrict__ crs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    int32_t c[3];
    xyz2crs(*gp, p, c);
    crs[3 * i] = c[0]; crs[3 * i + 1] = c[1]; crs[3 * i + 2] = c[2];
}

// utils.testOverlap batched (cutils.pyx:8-25): block per pair, threads over |A| x |B|.
__device__ __forceinline__ void test_overlap_pair(int p, const int32_t *__restrict__ crs, const int64_t *__restrict__ set_off,
                                                  const int32_t *__restrict__ a_idx, const int32_t *__restrict__ b_idx, unsigned int *__restrict__ out);
__global__ void __launch_bounds__(256) k_test_overlap(const int32_t *__restrict__ crs, const int64_t *__restrict__ set_off,
                                                      const int32_t *__restrict__ a_idx, const int32_t *__restrict__ b_idx,
                                                      unsigned int *__restrict__ out) {
    test_overlap_pair((int)blockIdx.x, crs, set_off, a_idx, b_idx, out);
}
__device__ __forceinline__ void test_overlap_pair(int p, const int32_t *__restrict__ crs, const int64_t *__restrict__ set_off,
                                                  const int32_t *__restrict__ a_idx, const int32_t *__restrict__ b_idx, unsigned int *__restrict__ out) {
    const int64_t a0 = set_off[a_idx[p]], a1 = set_off[a_idx[p] + 1];
    const int64_t b0 = set_off[b_idx[p]], b1 = set_off[b_idx[p] + 1];
    const int64_t na = a1 - a0, nb = b1 - b0;
    bool hit = false;
    for (int64_t t = threadIdx.x; t < na * nb && !hit; t += blockDim.x) {
        const int32_t *x = crs + 3 * (a0 + t / nb);
        const int32_t *y = crs + 3 * (b0 + t % nb);
        const int d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2];
        hit = d0 >= -1 && d0 <= 1 && d1 >= -1 && d1 <= 1 && d2 >= -1 && d2 <= 1;
    }
    if (hit) atomicOr(&out[p], 1u);
}

// The union job of aggregateCloud in ONE launch in front of its labelling kernels (round 5; three before: k_pool_gather, k_test_overlap, k_list_paint):
// blocks below paint_blocks take a pooled voxel each per thread -- item i < V goes to the residue group of its cloud's atom, item V + i to the domain
// group -- and set its bit in the group's volume straight from the clouds' voxel lists (no gathered copy of the coordinates); the blocks above test one
// bonded pair each (utils.testOverlap).  Behind k_job_init: the volume descriptors and the zeroed masks are the job's.
struct PoolPaint {
    const int32_t *src_crs; const int64_t *src_off;                                  // the clouds' voxel lists
    const int32_t *pool_cloud; const int64_t *pool_voff; const int32_t *pool_group;  // pooled clouds, their voxel offsets and residue groups
    int n_pool; int domain_group; long long V;
    const int64_t *set_off; const int32_t *pair_a, *pair_b; unsigned int *touch; int n_pairs;
    unsigned int paint_blocks;
    Counters *ctr;      // the union job's: a voxel outside its group's host-made volume raises unit_wait_failed (the host turns that into an error)
};
__global__ void __launch_bounds__(256) k_pool_paint(PoolPaint a, const VolDesc *__restrict__ vols, uint64_t *__restrict__ mask) {
    if (blockIdx.x >= a.paint_blocks) {      // block-uniform
        test_overlap_pair((int)(blockIdx.x - a.paint_blocks), a.src_crs, a.set_off, a.pair_a, a.pair_b, a.touch);
        return;
    }
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * a.V) return;
    const bool dom = i >= a.V;
    const long long j = dom ? i - a.V : i;
    int lo = 0, hi = a.n_pool - 1;   // the pooled cloud that holds voxel j: pool_voff[p] <= j < pool_voff[p + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.pool_voff[mid] <= j) lo = mid; else hi = mid - 1;
    }
    const long long src = a.src_off[a.pool_cloud[lo]] + (j - a.pool_voff[lo]);
    const VolDesc vd = vols[dom ? a.domain_group : a.pool_group[lo]];
    const int lc = a.src_crs[3 * src] - vd.org[0], lr = a.src_crs[3 * src + 1] - vd.org[1], ls = a.src_crs[3 * src + 2] - vd.org[2];
    // The volumes are the HOST's (the box around the sphere boxes of a group's pooled atoms, pdbeda_aggregate_cloud); the voxels come from
    // the clouds the device painted inside those sphere boxes.  Same xyz2crs, same radius, same alias rule -- should the two ever drift
    // apart, nothing is written outside the arena: the job is flagged and the call fails
    if ((unsigned)lc >= (unsigned)vd.dim[0] || (unsigned)lr >= (unsigned)vd.dim[1] || (unsigned)ls >= (unsigned)vd.dim[2]) {
        a.ctr->unit_wait_failed = 1u;
        return;
    }
    const int64_t w = vd.word_base + ((int64_t)ls * vd.dim[1] + lr) * vd.row_words + (lc >> 6);
    atomicOr((unsigned long long *)&mask[w], 1ull << (lc & 63));
}

// utils.createSymmetryAtoms (cutils.pyx:73-103): candidate t = ((cell * n_ops) + op) * n_atoms + atom in the reference's
// product order.  k_symmetry_keep tests every candidate (a keep BIT each: the flags are all the host needs to number the
// survivors in order), k_symmetry_pick evaluates the coordinates of the listed survivors with the same arithmetic.
__device__ inline bool symmetry_candidate(int64_t t, const double *__restrict__ xyz, int64_t n_atoms, const double *__restrict__ rot, int n_ops,
                                          const double *__restrict__ ortho, const double *__restrict__ lo, const double *__restrict__ hi, double v[3]) {
    const int64_t a = t % n_atoms;
    const int64_t cell_op = t / n_atoms;
    const int op = (int)(cell_op % n_ops);
    const int cell = (int)(cell_op / n_ops);
    const int i = cell / 9 - 1, j = (cell / 3) % 3 - 1, k = cell % 3 - 1;
    const double p[3] = {xyz[3 * a], xyz[3 * a + 1], xyz[3 * a + 2]};
    if (i == 0 && j == 0 && k == 0 && op == 0) {
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        return true;
    }
    const double ijk[3] = {(double)i, (double)j, (double)k};
    double ot[3];
    matvec3(ortho, ijk, ot);
    const double *rm = rot + 12 * op;
    bool in = true;
    for (int q = 0; q < 3; ++q) {
        double w = ((rm[4 * q] * p[0] + rm[4 * q + 1] * p[1]) + rm[4 * q + 2] * p[2]);
        w = (w + rm[4 * q + 3]) + ot[q];
        v[q] = w;
        in = in && (lo[q] - 5 <= w) && (w <= hi[q] + 5);
    }
    return in;
}

__global__ void __launch_bounds__(256) k_symmetry_keep(const double *__restrict__ xyz, int64_t n_atoms, const double *__restrict__ rot, int n_ops,
                                                       const double *__restrict__ ortho, const double *__restrict__ lo, const double *__restrict__ hi,
                                                       unsigned long long *__restrict__ keep_bits) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = 27ll * n_ops * n_atoms;
    double v[3];
    const bool keep = t < total && symmetry_candidate(t, xyz, n_atoms, rot, n_ops, ortho, lo, hi, v);
    const unsigned long long bits = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && t < total) keep_bits[t >> 6] = bits;
}

__global__ void __launch_bounds__(256) k_symmetry_pick(const double *__restrict__ xyz, int64_t n_atoms, const double *__restrict__ rot, int n_ops,
                                                       const double *__restrict__ ortho, const double *__restrict__ lo, const double *__restrict__ hi,
                                                       const int64_t *__restrict__ picked, int64_t n_picked, double *__restrict__ out_xyz) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_picked) return;
    double v[3];
    symmetry_candidate(picked[k], xyz, n_atoms, rot, n_ops, ortho, lo, hi, v);
    out_xyz[3 * k] = v[0]; out_xyz[3 * k + 1] = v[1]; out_xyz[3 * k + 2] = v[2];
}

// Nearest atom per centroid (scipy cdist + argmin, densityAnalysis.py:934-935): block per
// centroid, fp64 Euclidean, first index on ties.
__global__ void __launch_bounds__(256) k_nearest_atom(const double *__restrict__ cen, const double *__restrict__ atoms, int64_t n_atoms,
                                                      int64_t *__restrict__ index, double *__restrict__ distance) {
    __shared__ double s_d[256];
    __shared__ long long s_i[256];
    const int64_t c = blockIdx.x;
    const double cx = cen[3 * c], cy = cen[3 * c + 1], cz = cen[3 * c + 2];
    double best = INFINITY;
    long long bi = -1;
    for (int64_t a = threadIdx.x; a < n_atoms; a += blockDim.x) {
        const double dx = cx - atoms[3 * a], dy = cy - atoms[3 * a + 1], dz = cz - atoms[3 * a + 2];
        const double d = __dsqrt_rn((dx * dx + dy * dy) + dz * dz);
        if (d < best) { best = d; bi = a; }
    }
    s_d[threadIdx.x] = best;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double d2 = s_d[threadIdx.x + off];
            const long long i2 = s_i[threadIdx.x + off];
            if (i2 >= 0 && (d2 < s_d[threadIdx.x] || (d2 == s_d[threadIdx.x] && i2 < s_i[threadIdx.x]) || s_i[threadIdx.x] < 0)) {
                s_d[threadIdx.x] = d2;
                s_i[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { index[c] = s_i[0]; distance[c] = s_d[0]; }
}

}  // namespace pdbeda
