// pdbeda_tile.h -- the whole-map fast path (pdbeda_full_blobs / pdbeda_full_blobs_pm).
//
// k_tile_label<CW> (one 512-thread workgroup per tile of CW words x 8 rows x 8 sections, i.e. up to
// 256 c x 8 r x 8 s = 16 Ki voxels; WAVE w owns SECTION w of the tile from the first load to the last fold):
//   A1 the wave streams its section once from HBM (coalesced 256-B wave loads); the compare masks ARE the wave
//      ballots -> bit masks of both signs (fused green / red); the significant values are parked in the wave's
//      own slice of LDS, compacted in lane order
//   -- barrier 1 (the masks of all sections are in LDS) --
//   A2 lane = (sign, row, word) of the section: the lane counts the word-runs of its ROW's mask word; ONE packed
//      32-lane DPP scan numbers them (ids follow wave, sign, row, word) and places the parked values: these are the ids the
//      later kernels look voxels up by, and what C2 hands out -- the unions do not run on them (round 5)
//   B  26-connected components inside the tile on BLOCK NODES: the word-runs of the OR of the four row masks of a 2 x 2
//      (row, section) group -- connected sets under 26-connectivity; 16 groups a tile instead of 64 rows, a third of the nodes,
//      no pair inside a group.  Wave = (sign, one of the four earlier groups), lane = (group, word): the touching pairs are the
//      set bits of one bit expression over the rows on the shared face (a pair is charged to the later of its two run starts,
//      which makes it unique), either side's node is the last start <= the bit in its group's OR mask.  The lanes of a wave LIST
//      their pairs, then lane k unites pair k, k + 64, ... in a lock-free union-find in LDS (optimistic atomic min on the larger
//      id; find splits the path it walks).  Waves do not wait for each other: every wave of a sign numbers the nodes for itself.
//   -- barrier 2 (all unions done) --
//   C1 roots (among the node ids) take component numbers; every lane describes the row-runs of its word in the idle upper
//      halves of the parent table
//   -- barrier 3 --
//   C2 a THREAD PER ROW-RUN: its node (the group's starts, kept in LDS), fp64 (sum rho, sum rho * c) over the run's parked values in order, rounded once to the
//      job's quantum and folded into the component's accumulators with integer LDS atomics (FixSums: the result
//      does not depend on the order); run -> component ids are published for the label writer, and per mask word
//      the components of its first 7 runs and of the run at its last bit, a byte each, for the face merge
//   -- barrier 4 --  one record per tile component is flushed to HBM.
// Four barriers on a tile's path (round 2: 24); per-run sums never leave the chip.  The kernel is bound by
// instruction issue: a divergent per-lane loop issues for its slowest lane, which is why
// pairs and runs are handed out one per lane instead of being walked word by word.
// Only component pairs that touch across a tile face are united globally (k_face_merge, after an LDS
// de-duplication per tile), and only non-root tile components cost global atomics (k_resolve_tiles, after an
// LDS pre-reduction per tile).  Tiles beyond the LDS tables (round 4): more than CCAP components -> a WIDE tile (tile_mode 2: its
// components take ids above the tiles' own ranges, its sums are made CCAP components at a time); more than RCAP word-runs -> a
// DENSE tile (the parked values' LDS becomes parent slots RCAP .. RCAP_DENSE - 1, the values are re-read from L2); both keep
// their in-LDS unions.  Only a tile with more than RCAP_DENSE word-runs (both signs of a checkerboard), or one that finds no ids,
// is a "unit tile" (tile_mode 1 / 3): labelled run by run, every run its own component, and united globally by two launches
// of their own (k_unit_label, k_unit_pairs) in a SECOND run of the job -- the first run, enqueued without them, flags itself
// (Counters::overflow bit 2; round 5: no workgroup waits for another any more).  Slower, same result.
//
// The step is FOUR launches (round 4; six in round 3): k_tile_label -> k_face_merge -> k_resolve_tiles -> k_labels_tiles<fused>
// (+ the two unit launches behind k_face_merge in the second run of a job that has unit tiles).
// The cross-tile unions hang by FIRST KEY (kpar[]: first key << 32 | id, the later first voxel under the earlier one), so the
// root of a blob holds the blob's first key the moment the unions are done: k_resolve_tiles paints the keys while it finds the
// roots (no fold of keys, no painting kernel), and the label writer -- every component's packed parent carries its root's
// key -- ranks the keys of its own tile's components itself and writes the blob table rows of the tile's roots (no emit
// kernel, no label table from another launch).
#pragma once
#include "pdbeda_kernels.h"
#include <type_traits>

namespace pdbeda {

constexpr int TILE_R = 8, TILE_S = 8;
constexpr int RCAP = 1408;  // word-runs of a tile (both signs) whose parents fit the parent table proper
constexpr int RCAP_DENSE = 4096;   // ... and with the parked values' LDS as more of the same table (a tile that dense re-reads its values from L2 anyway)
constexpr int CCAP = 256;   // tile-local components (both signs together) handled in LDS
static_assert(CCAP == TILE_COMPS, "k_emit walks the tiles' component ranges");
constexpr int VCAP = 3584;  // significant values of a tile parked in LDS: one private region of VCAP / 8 per wave (= section)
constexpr int FACE_K = 7;    // word-runs of a word whose components k_face_merge finds in the word's byte record (the 8th: by run id)

struct TileDims {
    int cw;                       // words per tile along c (1..4)
    int ctiles, rtiles, stiles;   // tile grid
    int n_planes;
    int uc, ur, us, row_words;    // the unique box (a copy of the volume descriptor: kernel argument, not a dependent load)
    int nc, nr;                   // stored row / section pitch of the grid (header.ncrs[0], [1])
    float cut[2];                 // threshold of plane p
    int sign[2];                  // +1: density >= cut, -1: density <= cut
};

__host__ __device__ inline int64_t tile_index(const TileDims &td, int plane, int wq, int r, int s) {
    return (((int64_t)plane * td.stiles + (s >> 3)) * td.rtiles + (r >> 3)) * td.ctiles + wq / td.cw;
}

// What workgroup 0 of k_tile_label publishes for the later kernels of the job (descriptors by value: no host
// staging buffer, no memset / init launch, no sync): the volume descriptors and the id counters.  Run / component
// ids below runs0 / comps0 are owned tile by tile; unit tiles take theirs above them (unit_label_tile).
struct JobInit {
    VolDesc v[2];
    unsigned int runs0, comps0;
};

// number of set bits of a wave-uniform 64-bit mask at lane positions < the calling lane
__device__ inline uint32_t mbcnt_lt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ inline uint32_t mbcnt_lt_add(uint64_t mask, uint32_t add) {   // ... + add (the instruction has an accumulator operand)
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, add));
}

// lane `sel` of `old` <- the wave-uniform value `sval` (v_writelane_b32; this clang has no builtin for it).
// A select on `lane == sel` would do, but its 64-bit masks are loop invariant: hoisted, 16 of them spill the SGPRs.
// (`sel` must fold to a constant 0..63 -- an inline constant: a second SGPR would break the constant-bus limit.)
__device__ __forceinline__ uint32_t wave_writelane(uint32_t old, uint32_t sval, int sel) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(__builtin_amdgcn_readfirstlane((int)sval)), "n"(sel));
    return old;
}

// orders this wave's LDS accesses for its own lanes (LDS executes a wave's instructions in order: nothing to wait for,
// the fences only pin the compiler)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A workgroup barrier for LDS traffic only: __syncthreads() also waits for the wave's global stores to be acknowledged
// (s_waitcnt vmcnt(0)), which is what a kernel that has fire-and-forget stores in flight must NOT do.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Tile t owns component ids [t * CCAP, (t+1) * CCAP) and run ids [t * runs_per_tile, ...): no
// allocation atomics on the fast path.  Unused component ids are marked empty (r_n = 0).
template <typename JobRef>
__device__ inline void mark_comps_unused(const JobRef &job, uint32_t cb, uint32_t from, int tid, int nt) {
    unsigned long long *kpar = job.kpar;
    uint32_t *r_n = job.r_n;
    for (uint32_t i = from + tid; i < (uint32_t)CCAP; i += nt) {
        kpar[cb + i] = KP_UNUSED;   // (parent[] of every id of a tile is written by k_resolve_tiles)
        r_n[cb + i] = 0u;
    }
}

// The Job of a kernel whose FIRST argument it is, read from the kernel-argument segment at the point of use.  A by-value
// argument is loaded into scalar registers at kernel entry and stays live until its last use: the ~20 pointers that only
// the tail of k_tile_label stores through would sit on 40 SGPRs during the hot phases.  The empty asm keeps the loads below it.
typedef const Job __attribute__((address_space(4))) *JobKernarg;
#define PDBEDA_LATE_JOB(name)                                                      \
    JobKernarg name##_p = (JobKernarg)__builtin_amdgcn_kernarg_segment_ptr();      \
    asm volatile("" : "+s"(name##_p));                                             \
    const Job __attribute__((address_space(4))) &name = *name##_p

// A kernel that reads its arguments late reads them in many small scalar loads, each behind a wait; the kernel-argument
// segment is cold in the scalar cache at kernel start, so every new 64-byte line of it costs a trip to memory -- ten of them in
// a row held the fused label writer's first loads back by ~5 us (stamps).  Touching all lines at once, at entry, makes that
// one trip: the later loads hit the scalar cache.  (eight lines: Job + TileDims + the two pointers behind them)
__device__ __forceinline__ void kernarg_prefetch() {
    JobKernarg ka = (JobKernarg)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
    asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\t"
                 "s_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\t"
                 "s_waitcnt lgkmcnt(0)"   // (inside the statement: the registers are the compiler's again behind it, and a load must not land later)
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7) : "s"(ka) : "memory");
}
static_assert(sizeof(Job) + sizeof(TileDims) + 16 > 0x1c0 && sizeof(Job) + sizeof(TileDims) + 16 <= 0x200, "eight 64-byte lines of kernel arguments");

// Generic labelling of a tile k_tile_label could not hold in LDS ("unit tile"): every run is its own component with its own
// record (wave prefix sums per word, as k_run_index), ids from the job's counters above the per-tile ranges; the pairs of such a
// tile are then ALL united globally.  Called by the tile's own workgroup of k_face_merge (round 3: this used to be 128 extra
// workgroups of that kernel behind a grid barrier of their own -- 1.3 ms for a map whose tiles all overflow; and inside
// k_tile_label, where the tile has everything at hand, the cold code cost the other tiles 3 us of spilled scalars).  A quarter
// of the tile (16 rows) at a time; `scratch`: 1 KiB of LDS nobody else uses.
template <int CW, typename JobRef>
__device__ void unit_label_tile(const JobRef &job, const float *__restrict__ dens, const Geom *__restrict__ gp, const TileDims &td,
                                int w0, int r0, int s0, unsigned char *scratch) {
    constexpr int QU = 16 * CW;   // units of a quarter tile
    uint64_t *s_m = reinterpret_cast<uint64_t *>(scratch);          // [64]
    uint32_t *s_off = reinterpret_cast<uint32_t *>(scratch + 512);  // [64]
    uint32_t *s_base = reinterpret_cast<uint32_t *>(scratch + 768); // [3]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n_waves = (int)(blockDim.x >> 6);
    const Geom &g = *gp;
    const int ur = td.ur, us = td.us, row_words = td.row_words;
    const int64_t plane_words = (int64_t)row_words * ur * us;
    for (int q = 0; q < td.n_planes; ++q) {
        const VolDesc vd = job.vols[q];
        for (int quarter = 0; quarter < 4; ++quarter) {
            // thread tid < QU owns unit quarter * QU + tid
            const int u = quarter * QU + tid;
            const int my_wl = u % CW, my_rowl = (u / CW) & 63;
            const int my_rl = my_rowl & 7, my_sl = my_rowl >> 3;
            const bool my_valid = (tid < QU) && (r0 + my_rl < ur) && (s0 + my_sl < us) && (w0 + my_wl < row_words);
            const int64_t my_word = (int64_t)q * plane_words + ((int64_t)(s0 + my_sl) * ur + (r0 + my_rl)) * row_words + (w0 + my_wl);
            const uint64_t m = my_valid ? job.mask[my_word] : 0ull;
            const uint32_t cnt = (uint32_t)popc64(run_starts(m));
            uint32_t x = (wv == 0) ? cnt : 0u;   // QU <= 64: all owners sit in wave 0
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            if (tid < 64) { s_m[tid] = (tid < QU) ? m : 0ull; s_off[tid] = x - cnt; }
            if (tid == 63) {
                s_base[0] = x ? atomicAdd(&job.ctr->n_runs, x) : 0u;
                s_base[1] = x ? atomicAdd(&job.ctr->n_comps, x) : 0u;
                // the ids this job has may not cover a map this dense: nothing is written beyond them -- the job is flagged, the
                // unit work of every later kernel is skipped, and the host runs the job again in a worst-case arena
                s_base[2] = (s_base[0] + x > job.run_cap || s_base[1] + x > job.comp_cap) ? 1u : 0u;
                if (s_base[2]) atomicOr(&job.ctr->overflow, 1u);
            }
            __syncthreads();
            const uint32_t run_first = s_base[0], comp_first = s_base[1];
            const bool no_room = s_base[2] != 0u;   // block-uniform
            if (my_valid) job.run_base[my_word] = no_room ? 0u : run_first + s_off[tid];
            for (int j = wv; j < QU && !no_room; j += n_waves) {   // a wave per unit
                const uint64_t mw = s_m[j];
                if (mw == 0ull) continue;
                const int uu = quarter * QU + j;
                const int wl = uu % CW, rowl = (uu / CW) & 63;
                const int r = r0 + (rowl & 7), s = s0 + (rowl >> 3), c0 = (w0 + wl) * 64;
                const uint32_t run0 = run_first + s_off[j], comp0 = comp_first + s_off[j];
                word_run_records(job, g, dens, vd, mw, lane, c0, r, s, c0, r, s, comp0);
                const uint64_t starts = run_starts(mw);
                if ((starts >> lane) & 1ull) {
                    const uint32_t k = (uint32_t)popc64(starts & bits_below(lane));
                    job.comp_of_run[run0 + k] = comp0 + k;
                }
            }
            __syncthreads();
        }
    }
}

template <int CW>
__global__ void __launch_bounds__(512, 8) k_tile_label(Job job, const float *__restrict__ dens, const Geom *__restrict__ gp, TileDims td, JobInit init) {
    constexpr int NT = 512, NW = 8;
    constexpr int NU = 64 * CW;     // (row, word) units of the tile
    constexpr int USEC = 8 * CW;    // units of a section = of a wave
    constexpr int CHU = USEC < 16 ? USEC : ((CW == 3) ? 12 : 16);  // units per chunk of the stream: whole rows
    constexpr int VREG = VCAP / NW;
    constexpr uint32_t ROOT16 = 0x8000u;   // C1: the low half of parent[root] = ROOT16 | component number
    static_assert(USEC % CHU == 0 && CHU % CW == 0, "chunks are whole rows");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    __shared__ uint64_t s_mask[2][256];
    // one block: the parent table and, behind it, the parked values.  A tile with more than RCAP word-runs (noise below ~1.1
    // sigma: 1 500-2 700 a tile) has sections far too dense to park (VREG values each) and reads its values from L2 in C2
    // whatever happens -- so the values' 14 KiB are its parent slots RCAP .. RCAP_DENSE - 1 (round 4: such a tile was a unit
    // tile before, labelled run by run and united through global atomics -- 1.6 ms a step for noise at 1 sigma against 0.09 at 1.5)
    __shared__ __attribute__((aligned(16))) uint32_t s_pv[RCAP + VCAP];
    static_assert(RCAP_DENSE <= RCAP + VCAP && RCAP_DENSE <= 32768 && RCAP_DENSE <= 64 * 4 * 32, "parent slots of a dense tile; 15-bit component numbers; the tile's own run id range");
    uint32_t *s_parent = s_pv;
    float *s_val = reinterpret_cast<float *>(s_pv + RCAP);
    // phase B: the waves' edge lists; phase C: the component accumulators (sums relative to the tile origin)
    __shared__ __attribute__((aligned(16))) unsigned char s_blob[CCAP * 48];
    // (integers: multiples of the job's quantum, see FixSums -- LDS atomics fold them in whatever order, the result is the same)
    unsigned long long *s_rho = reinterpret_cast<unsigned long long *>(s_blob), *s_rho_c = s_rho + CCAP, *s_rho_r = s_rho + 2 * CCAP, *s_rho_s = s_rho + 3 * CCAP;
    // integer sums of a component, relative to the tile origin, packed so that a run costs two LDS atomics, not four:
    // s_pk = voxels (20 bits) | sum (r - r0) << 20 (20 bits) | sum (s - s0) << 40;  s_crel = sum (c - c_tile)
    unsigned long long *s_pk = reinterpret_cast<unsigned long long *>(s_rho + 4 * CCAP);
    uint32_t *s_crel = reinterpret_cast<uint32_t *>(s_pk + CCAP), *s_key = s_crel + CCAP;   // s_key: plane << 31 | c-major key inside the plane (min = first voxel)
    __shared__ __attribute__((aligned(16))) uint32_t s_wtot[NW];   // word-runs of section w (both signs); bit 31: it could not park all its values
    __shared__ uint32_t s_ub[2][256];  // per sign and unit: first parked value of the unit | first word-run of the word << 16
    __shared__ uint32_t s_ncomp;
    // the union-find NODES (round 5): per sign, 2 x 2 (row, section) group and word, the starts of the word-runs of the OR of the
    // group's four row masks, and the id of the first of them; s_gtot: nodes of either sign
    __shared__ uint64_t s_sg[2][64];
    __shared__ uint16_t s_nb[2][64];
    __shared__ uint32_t s_gtot[2];

    const int uc = td.uc, ur = td.ur, us = td.us;   // (kernel arguments: no dependent load through gp before the stream can start)
    const int nc = td.nc, nr = td.nr;
    // (a 3-D grid: the tile coordinates come with the workgroup -- two integer divisions here were ~50 scalar instructions in
    //  every wave, and the scalar unit of a CU serves all 32 of them.  Workgroups are dealt x fastest: tile order, as before)
    const int ct = (int)blockIdx.x, rt = (int)blockIdx.y, st = (int)blockIdx.z;
    const uint32_t bid = ((uint32_t)st * (uint32_t)td.rtiles + (uint32_t)rt) * (uint32_t)td.ctiles + (uint32_t)ct;
    const int w0 = ct * CW, r0 = rt * TILE_R, s0 = st * TILE_S;
    const int row_words = (uc + 63) >> 6;
    const int n_planes = td.n_planes;

    {
        PDBEDA_LATE_JOB(pj);
        if (tid == 0) s_ncomp = 0;
        if (bid == 0 && tid == 0) {   // read by the kernels that follow; nothing in this kernel touches them
            pj.vols[0] = init.v[0];
            if (td.n_planes > 1) pj.vols[1] = init.v[1];
            Counters c;
            memset(&c, 0, sizeof c);
            c.n_runs = init.runs0;
            c.n_comps = init.comps0;
            *pj.ctr = c;
            // a tile with more than CCAP components takes ids from c.n_comps further down (the wide tiles): not before this
            // flag says that the counters are this job's (epoch: stale values of a recycled arena never match)
            __hip_atomic_store(&pj.unit_flag[1], pj.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (the first-key bitmap, the rank counters and the inbox counters are cleared by k_face_merge: this kernel is bound by
        //  instruction issue, that one by memory round trips -- its issue slots are free)
    }
    for (int i = tid; i < RCAP; i += NT) s_parent[i] = (uint32_t)i;

    // the streaming phase runs at raised wave priority: a tile whose data arrives late shares its CU with tiles that are already
    // in their LDS phases, and it is the late tile that ends the kernel
    const uint64_t t_in = __builtin_amdgcn_s_memrealtime();   // (100 MHz; see the priorities behind barrier 1)
    __builtin_amdgcn_s_setprio(3);
    // ---- A1: the wave streams its section: compare, ballot, park the significant values (lane order) in its LDS region.
    //      A tile that lies wholly inside the grid (all but the last ones along each axis) takes the unguarded path:
    //      scalar row bases + immediate offsets, no per-load address arithmetic or exec juggling.  Eleven vector
    //      instructions per 64 voxels: this phase is issue bound next to the HBM stream.
    {
        const bool interior = (r0 + TILE_R <= ur) && (s0 + TILE_S <= us) && ((w0 + CW) * 64 <= uc);   // block-uniform
        uint32_t vcnt = 0;      // wave-uniform: significant values so far (> VREG: the region overflowed)
        uint32_t rtot = 0;      // lanes < CHU: word-runs of the units they collected
        const uint32_t vreg = (uint32_t)wvs * VREG;
        const float c0 = td.cut[0], c1 = td.cut[1];
        auto stream = [&](auto interior_tag, auto pos_tag, auto two_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value, POS = decltype(pos_tag)::value, TWO = decltype(two_tag)::value;
#pragma unroll 1
            for (int chunk = 0; chunk < USEC / CHU; ++chunk) {
                const int u0 = wvs * USEC + chunk * CHU;   // first unit of the chunk (scalar; a multiple of CW)
                const int rowl0 = u0 / CW;                 // its CHU / CW rows lie in one section
                const float *cbase = dens + ((int64_t)(s0 + (rowl0 >> 3)) * nr + (r0 + (rowl0 & 7))) * nc + w0 * 64 + lane;
                float v[CHU];
#pragma unroll
                for (int jj = 0; jj < CHU; ++jj) {
                    const float *ptr = cbase + (int64_t)(jj / CW) * nc + (jj % CW) * 64;
                    if (INTERIOR) {
                        v[jj] = *ptr;
                    } else {
                        const int rowl = rowl0 + jj / CW;
                        const bool in = (r0 + (rowl & 7) < ur) && (s0 + (rowl >> 3) < us) && ((w0 + jj % CW) * 64 + lane < uc);
                        v[jj] = in ? *ptr : 0.0f;
                    }
                }
                // lane jj of the chunk collects the masks of unit jj (v_writelane from the scalar ballots)
                uint32_t k0lo = 0, k0hi = 0, k1lo = 0, k1hi = 0;
#pragma unroll
                for (int jj = 0; jj < CHU; ++jj) {
                    const int u = u0 + jj;
                    const int wl = u % CW, rowl = u / CW;
                    const float x = v[jj];
                    bool hit0 = POS ? x >= c0 : x <= c0;
                    bool hit1 = TWO && x <= c1;
                    if (!INTERIOR) {
                        const bool in = (r0 + (rowl & 7) < ur) && (s0 + (rowl >> 3) < us) && ((w0 + wl) * 64 + lane < uc);
                        hit0 = hit0 && in;
                        hit1 = hit1 && in;
                    }
                    const uint64_t b0 = __ballot(hit0);
                    const uint64_t b1 = __ballot(hit1);
                    const uint64_t bb = b0 | b1;
                    const uint32_t nv = (uint32_t)popc64(bb);
                    // branch-free: a value beyond the region lands on its last slot, and vcnt > VREG at the end says that the
                    // section was too dense to park (C2 then re-reads the tile's values from global memory / L2)
                    if (hit0 || hit1) s_val[vreg + min(mbcnt_lt_add(bb, vcnt), (uint32_t)(VREG - 1))] = x;
                    vcnt += nv;
                    k0lo = wave_writelane(k0lo, (uint32_t)b0, jj);
                    k0hi = wave_writelane(k0hi, (uint32_t)(b0 >> 32), jj);
                    k1lo = wave_writelane(k1lo, (uint32_t)b1, jj);
                    k1hi = wave_writelane(k1hi, (uint32_t)(b1 >> 32), jj);
                }
                if (lane < CHU) {
                    const uint64_t k0 = ((uint64_t)k0hi << 32) | k0lo, k1 = ((uint64_t)k1hi << 32) | k1lo;
                    s_mask[0][u0 + lane] = k0;
                    s_mask[1][u0 + lane] = k1;
                    rtot += (uint32_t)popc64(run_starts(k0)) + (uint32_t)popc64(run_starts(k1));
                }
            }
        };
        const bool pos = td.sign[0] > 0;
        if (n_planes > 1) {   // (a fused job: plane 0 is the ">= cut" plane)
            if (interior) stream(std::true_type{}, std::true_type{}, std::true_type{}); else stream(std::false_type{}, std::true_type{}, std::true_type{});
        } else if (interior) {
            if (pos) stream(std::true_type{}, std::true_type{}, std::false_type{}); else stream(std::true_type{}, std::false_type{}, std::false_type{});
        } else {
            if (pos) stream(std::false_type{}, std::true_type{}, std::false_type{}); else stream(std::false_type{}, std::false_type{}, std::false_type{});
        }
        // word-runs of the section: the units sat on lanes < 16 (one DPP row)
        rtot += dpp0<DPP_ROW_SHR + 1>(rtot);
        rtot += dpp0<DPP_ROW_SHR + 2>(rtot);
        rtot += dpp0<DPP_ROW_SHR + 4>(rtot);
        rtot += dpp0<DPP_ROW_SHR + 8>(rtot);
        if (lane == 15) s_wtot[wvs] = rtot | (vcnt > (uint32_t)VREG ? 0x80000000u : 0u);
    }
    __syncthreads();   // ---- barrier 1 ----
    {   // The tile whose data arrived LATE is the one that ends the kernel: HBM serves the workgroups in dispatch order (stamps by
        // tile id: the first eighth of the tiles has its section after 5.9 us, the last after 10.7), every CU holds a tile of each
        // quarter, and from 10 to 25 us all four are in the vector-bound phases below at once.  So the later a tile left the
        // stream, the higher its priority from here on: the early tiles have the slack (r04 A/B: -1.8 us of 34; by tile id
        // instead of by the clock: -1.3; two levels instead of four: -1.0).  Thresholds in 10 ns ticks, for MI355X's HBM.
        const uint32_t dt = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_in);
        if (dt > 1000u) __builtin_amdgcn_s_setprio(3);
        else if (dt > 850u) __builtin_amdgcn_s_setprio(2);
        else if (dt > 650u) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }

    // ---- A2: lane = (sign q, row rl, word wl) of section wvs: the word-runs of the ROWS are numbered (ids follow wave, sign, row,
    //      word): they are what the later kernels look voxels up by, and what C2 sums; the unions no longer run on them ----
    const int q = lane >> 5, usec = lane & 31;
    const bool act = usec < USEC;
    const int rl = usec / CW, wl = usec % CW;
    const int u = wvs * USEC + (act ? usec : 0);
    uint32_t wbase = 0, n_runs = 0;
    bool from_global = false;
    {   // lanes 0..7 hold the sections' totals: count | overflow << 16 (a tile has < 2^16 word-runs), scanned inside the DPP row
        const uint32_t raw = lane < NW ? s_wtot[lane] : 0u;
        const uint32_t v = (raw & 0x7fffffffu) | ((raw >> 31) << 16);
        uint32_t x = v;
        x += dpp0<DPP_ROW_SHR + 1>(x);
        x += dpp0<DPP_ROW_SHR + 2>(x);
        x += dpp0<DPP_ROW_SHR + 4>(x);
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)x, NW - 1);
        const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)x, wvs), mine = (uint32_t)__builtin_amdgcn_readlane((int)v, wvs);
        n_runs = all & 0xffffu;
        from_global = (all >> 16) != 0u;
        wbase = (upto - mine) & 0xffffu;
    }
    const uint64_t m = act ? s_mask[q][u] : 0ull, mo = act ? s_mask[q ^ 1][u] : 0ull;
    // my word
    const bool my_valid = act && q < n_planes && (r0 + rl < ur) && (s0 + wvs < us) && (w0 + wl < row_words);
    const int64_t plane_words = (int64_t)row_words * ur * us;
    const int64_t my_word = (int64_t)q * plane_words + ((int64_t)(s0 + wvs) * ur + (r0 + rl)) * row_words + (w0 + wl);
    const int64_t tile_id = (int64_t)bid;

    if (n_runs > (uint32_t)RCAP && n_runs <= (uint32_t)RCAP_DENSE) {   // block-uniform: a dense tile (see s_pv)
        from_global = true;
        for (uint32_t i = (uint32_t)RCAP + tid; i < n_runs; i += NT) s_parent[i] = i;
        __syncthreads();
    }
    if (n_runs == 0 || n_runs > (uint32_t)RCAP_DENSE) {   // block-uniform
        PDBEDA_LATE_JOB(lj);
        // nothing significant, or too many runs for LDS: publish the masks; an overflowing tile is labelled run by run
        // (every run its own component) by its workgroup of k_face_merge
        if (my_valid) { lj.mask[my_word] = m; lj.run_base[my_word] = 0u; }
        if (tid == 0) { lj.tile_mode[tile_id] = n_runs ? 1 : 0; lj.tile_runs[bid] = 0u; if (n_runs) { lj.unit_flag[0] = lj.epoch; lj.unit_flag[2] = lj.epoch; } }
        mark_comps_unused(lj, (uint32_t)bid * CCAP, 0u, tid, NT);
        return;
    }

    const uint64_t SA = run_starts(m);
    // one packed scan per half wave: word-runs of my section [0, 11), parked values [22, 32)
    const uint32_t packed = (uint32_t)popc64(SA) | ((uint32_t)popc64(m | mo) << 22);
    const uint32_t incl = half_scan(packed), excl = incl - packed;
    const uint32_t half0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);   // totals of the sign-0 lanes
    const uint32_t my_base = wbase + (q ? (half0 & 0x7ffu) : 0u) + (excl & 0x7ffu);                         // id of my first run
    const uint32_t vb = (uint32_t)wvs * VREG + (excl >> 22);   // my unit's first parked value (both sign lanes of a unit agree)

    // ---- B: 26-connected components inside the tile, on BLOCK nodes (round 5).  Under 26-connectivity every voxel of a 2 x 2
    // group of rows (rows 2R, 2R + 1 of sections 2S, 2S + 1) at column p touches every voxel of the group at p - 1, p, p + 1: the
    // runs of the OR of the group's four row masks are connected sets, and they are the union-find nodes -- 16 groups a tile
    // instead of 64 rows, a third of the nodes, no pairs inside a group (the block-based idea of the GPU labelling literature
    // on this kernel's bit masks; rounds 1-4 united the word-runs of single rows: 2.3 pairs per word and sign).
    // Lane = (group, word).  A group meets four earlier groups -- (R - 1, S)
    // across its r face, (R, S - 1) across its s face, (R - 1, S - 1) and (R + 1, S - 1) across an edge -- and only the rows on
    // that face can touch: with FA / FB the OR of the face rows of either side, a touching pair of face runs is charged to the
    // later of its two starts, as before --
    //     EA = starts(FA) & (FB | FB << 1 | carry FB)      EB = starts(FB) & (FA << 1 | carry FA)
    // and at a set bit p either side's NODE is the last start <= p of its group's OR mask (a face run lies inside one node:
    // repeats are harmless).  Node ids: sign 0 counts up from 0, sign 1 ends at n_runs - 1 (a node holds at least one whole
    // word-run of a row: there are never more nodes than those) -- the signs do not wait for each other.
    // Wave = (sign, which of the four earlier groups): all eight waves work, each lists and unites the pairs of ONE relation
    // (r05 A/B: one wave per sign doing all four took 5.4 us per tile on the critical path, the other six waiting at the barrier).
    // Every wave of a sign numbers the nodes for itself -- the same scan over the same masks -- so nobody waits for ids.
    const int gq = wvs >> 2, grel = wvs & 3;   // wave-uniform
    if (gq < n_planes) {
        constexpr int ECAPW = CCAP * 48 / 4 / NW;   // pairs a wave lists before it unites them (the lists live where the component sums go later)
        uint32_t *g_edge = reinterpret_cast<uint32_t *>(s_blob) + wvs * ECAPW;
        const int gi = lane / CW, gwl = lane % CW;
        const bool gact = gi < 16;
        const int R = gi & 3, S = (gi >> 2) & 3;
        auto MK = [&](int sec, int row) -> uint64_t { return s_mask[gq][sec * USEC + row * CW + gwl]; };
        const uint64_t a = gact ? MK(2 * S, 2 * R) : 0ull, b = gact ? MK(2 * S, 2 * R + 1) : 0ull;
        const uint64_t c = gact ? MK(2 * S + 1, 2 * R) : 0ull, d = gact ? MK(2 * S + 1, 2 * R + 1) : 0ull;
        const uint64_t G = (a | b) | (c | d);
        const uint64_t SG = run_starts(G);
        const uint32_t gcnt = (uint32_t)popc64(SG);
        const uint32_t gincl = wave_scan(gcnt);
        const uint32_t gtot = (uint32_t)__builtin_amdgcn_readlane((int)gincl, 63);
        const uint32_t nbase = (gq ? n_runs - gtot : 0u) + gincl - gcnt;
        if (grel == 0) {   // for C1 / C2
            s_sg[gq][lane] = SG;
            s_nb[gq][lane] = (uint16_t)nbase;
            if (lane == 0) s_gtot[gq] = gtot;
        }
        // my relation: the earlier group (R + dR, S + dS), my face rows towards it and its face rows towards me
        const int dR = grel == 1 ? 0 : (grel == 3 ? 1 : -1), dS = grel == 0 ? 0 : -1;
        const int R2 = R + dR, S2 = S + dS;
        const bool has = gact && R2 >= 0 && R2 < 4 && S2 >= 0;   // (else: in another tile -- k_face_merge's pairs)
        const int R2c = has ? R2 : R, S2c = has ? S2 : S;
        uint64_t a2 = MK(2 * S2c, 2 * R2c), b2 = MK(2 * S2c, 2 * R2c + 1), c2 = MK(2 * S2c + 1, 2 * R2c), d2 = MK(2 * S2c + 1, 2 * R2c + 1);
        if (!has) { a2 = 0ull; b2 = 0ull; c2 = 0ull; d2 = 0ull; }
        const uint64_t SG2 = run_starts((a2 | b2) | (c2 | d2));
        const uint32_t nbase2 = (uint32_t)__shfl((int)nbase, lane + (dS * 4 + dR) * CW);
        //                            (R - 1, S)        (R, S - 1)        (R - 1, S - 1)   (R + 1, S - 1)
        const uint64_t FA = grel == 0 ? (a | c) : (grel == 1 ? (a | b) : (grel == 2 ? a : b));
        const uint64_t FB = grel == 0 ? (b2 | d2) : (grel == 1 ? (c2 | d2) : (grel == 2 ? d2 : c2));
        // bit 63 of the word to the left, for every mask that takes part (cross-lane reads sit outside every conditional)
        const uint32_t hbits = (uint32_t)(G >> 63) | ((uint32_t)(FA >> 63) << 1) | ((uint32_t)(FB >> 63) << 2);
        const uint32_t upb = dpp0<DPP_WAVE_SHR1>(hbits);
        const uint32_t car = gwl > 0 ? upb : 0u;
        const uint64_t E = (run_starts(FA) & (FB | (FB << 1) | (uint64_t)((car >> 2) & 1u))) | (run_starts(FB) & ((FA << 1) | (uint64_t)((car >> 1) & 1u)));   // EA and EB are disjoint
        const bool same_row = grel == 3 && (G & 1ull) && (car & 1u);   // my group's run continues across the word boundary
        const uint32_t ecount = (uint32_t)popc64(E) + (same_row ? 1u : 0u);
        const uint32_t eincl = wave_scan(ecount);
        const uint32_t etot = (uint32_t)__builtin_amdgcn_readlane((int)eincl, 63);
        const uint32_t mb1 = nbase - 1u, bb1 = nbase2 - 1u;
        if (etot <= (uint32_t)ECAPW) {   // wave-uniform
            // The lanes list their pairs (mine << 16 | earlier: ids of earlier groups are smaller), then lane k unites pair k, k + 64, ...:
            // a union is a chain of dependent LDS trips, and a lane with six pairs must not keep 63 others waiting
            uint32_t epos = eincl - ecount;
            if (same_row) g_edge[epos++] = (nbase << 16) | mb1;
            uint64_t todo = E;
            while (todo) {
                const uint64_t below = todo - 1, upto = todo ^ below;   // bits <= p
                todo &= below;
                g_edge[epos++] = ((mb1 + (uint32_t)popc64(SG & upto)) << 16) | (bb1 + (uint32_t)popc64(SG2 & upto));
            }
            wave_lds_sync();
            for (uint32_t e = lane; e < etot; e += 64) {
                const uint32_t pk = g_edge[e], x = pk >> 16, y = pk & 0xffffu;
                const uint32_t old = atomicMin(&s_parent[x], y);   // optimistic: most nodes are still roots when their first pair arrives
                if (old != x) lds_unite(s_parent, old, y);         // x hung under `old` already: unite that tree with y's
            }
        } else {   // a very dense tile: unite on the spot
            if (same_row) lds_unite(s_parent, nbase, mb1);
            uint64_t todo = E;
            while (todo) {
                const uint64_t below = todo - 1, upto = todo ^ below;
                todo &= below;
                lds_unite(s_parent, mb1 + (uint32_t)popc64(SG & upto), bb1 + (uint32_t)popc64(SG2 & upto));
            }
        }
    } else if (grel == 0 && lane == 0) {
        s_gtot[1] = 0u;
    }
    __syncthreads();   // ---- barrier 2: all unions done ----
    // ---- C1: the accumulators take the place of the edge lists; roots take component numbers; every lane describes the
    //      runs of its word (unit | sign << 8 | first bit << 9) in the idle upper halves of the parent table, so that C2 can
    //      hand ONE RUN to each thread: a lane-per-word loop issues for the word with the most runs and the longest run ----
    for (uint32_t i = tid; i < (uint32_t)CCAP; i += NT) {
        s_rho[i] = 0ull; s_rho_c[i] = 0ull; s_rho_r[i] = 0ull; s_rho_s[i] = 0ull;
        s_pk[i] = 0ull; s_crel[i] = 0u; s_key[i] = 0xffffffffu;
    }
    uint16_t *s_half = reinterpret_cast<uint16_t *>(s_parent);   // [2 i]: parent / ROOT16 | component, [2 i + 1]: descriptor of run i
    const uint32_t n_node0 = s_gtot[0], n_nodes = n_node0 + s_gtot[1], node_skip = n_runs - n_nodes;   // (sign 1's nodes end at n_runs - 1)
    for (uint32_t i = tid; i < n_nodes; i += NT) {   // (one LDS atomic per wave: the compiler aggregates)
        const uint32_t id = i < n_node0 ? i : i + node_skip;
        if (s_half[2 * id] == id) s_half[2 * id] = (uint16_t)(ROOT16 | atomicAdd(&s_ncomp, 1u));
    }
    {
        const uint32_t dw = (uint32_t)u | ((uint32_t)q << 8);
        uint64_t todo = SA;
        uint32_t id = my_base;
        while (todo) {
            s_half[2 * id + 1] = (uint16_t)(dw | ((uint32_t)ctz64(todo) << 9));
            todo &= todo - 1;
            ++id;
        }
        if (act) s_ub[q][u] = vb | (my_base << 16);
    }
    PDBEDA_LATE_JOB(lj);          // (everything below stores through pointers nothing above needs)
    const uint32_t cb = (uint32_t)bid * CCAP, rb = (uint32_t)bid * (uint32_t)(NU * 32);
    {   // what the later kernels read per word and per row (run bases are wrong for a tile that turns out to be a unit tile
        // below: that path writes them again)
        if (my_valid) { lj.mask[my_word] = m; lj.run_base[my_word] = rb + my_base; }
    }
    __syncthreads();   // ---- barrier 3 ----
    const uint32_t n_comp = s_ncomp;
    // More components than the accumulators hold (CCAP): a WIDE tile.  Its unions are done all the same -- in LDS, above --, so
    // it keeps them: the components take ids from the job's counter (above the tiles' own ranges, where the unit tiles' runs take
    // theirs), and the sums below are made CCAP components at a time (round 4; before, such a tile fell back to run-by-run
    // labelling and global unions of ALL its pairs in k_face_merge -- a protein-like map at 0.5 sigma, every tile of it: 1.2 ms
    // a step).  For everybody else it is a unit tile (tile_mode 2: look the component up by run id) whose in-tile pairs are done.
    const bool wide = n_comp > (uint32_t)CCAP;   // block-uniform
    uint32_t gb = cb;
    if (wide) {
        uint32_t *s_gb = s_wtot;   // (free since A2)
        if (tid == 0) {
            // the counters are written by workgroup 0 at its very start, some 15 us before any tile gets here; the flag makes that
            // order a fact instead of a habit of the dispatcher (bounded: a flag that does not come up sends the tile the old way)
            unsigned spins = 0;
            bool ready = true;
            while (__hip_atomic_load(&lj.unit_flag[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != lj.epoch) {
                if (++spins > (1u << 16)) { ready = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            uint32_t first = 0xffffffffu;
            if (ready) {
                first = atomicAdd(&lj.ctr->n_comps, n_comp);
                if (first + n_comp > lj.comp_cap) { atomicOr(&lj.ctr->overflow, 1u); first = 0xffffffffu; }   // (no ids left: flagged; the host runs the job again in the worst-case arena)
            }
            s_gb[0] = first;
        }
        __syncthreads();
        gb = s_gb[0];
        if (gb == 0xffffffffu) {   // block-uniform: a unit tile, labelled run by run by its workgroup of k_face_merge
            if (my_valid) lj.run_base[my_word] = 0u;
            if (tid == 0) { lj.tile_mode[tile_id] = 3; lj.tile_runs[bid] = 0u; lj.unit_flag[0] = lj.epoch; lj.unit_flag[2] = lj.epoch; }
            mark_comps_unused(lj, cb, 0u, tid, NT);
            return;
        }
    }
    uint32_t *comp_of_run = lj.comp_of_run + rb;
    // k_face_merge reads, per word, the tile-local components of its first FACE_K word-runs and (byte 7) of the run that
    // reaches the word's last bit, as the bytes of one 64-bit load (a wide tile's are not read: tile_mode)
    uint8_t *word_comps = lj.word_comps + (size_t)bid * (2 * 256 * 8);
    const int64_t keys_pp = (int64_t)uc * ur * us;
    for (uint32_t pass0 = 0;; pass0 += (uint32_t)CCAP) {   // (one pass unless the tile is wide)
        // ---- C2: a thread per run: sums over the parked values, fold into the component ----
        {
            const int ctile = w0 * 64;
            const uint32_t urus = (uint32_t)ur * (uint32_t)us;
            const double fix_mul = lj.fix_mul;
            for (uint32_t i = tid; i < n_runs; i += NT) {
                const uint32_t desc = s_half[2 * i + 1];
                const int ru = (int)(desc & 0xffu), rq = (int)((desc >> 8) & 1u), a = (int)(desc >> 9);
                const int rsl = ru / USEC, rrl = (ru % USEC) / CW, rwl = ru % CW;
                // my node: the run of my group's OR mask that covers my first bit (the last start <= a)
                const int gl = ((rsl >> 1) * 4 + (rrl >> 1)) * CW + rwl;
                uint32_t x = (uint32_t)s_nb[rq][gl] + (uint32_t)popc64(s_sg[rq][gl] & ((2ull << a) - 1ull)) - 1u;
                do x = s_half[2 * x]; while (!(x & ROOT16));
                const uint32_t comp_all = x & 0x7fffu;
                if (wide && comp_all - pass0 >= (uint32_t)CCAP) continue;   // (not this pass's)
                const uint32_t comp = comp_all - pass0;
                const uint64_t rm = s_mask[rq][ru];
                const uint32_t ub = s_ub[rq][ru];
                const uint64_t inv = ~(rm >> a);
                const int len = inv ? ctz64(inv) : 64;   // (a run that fills bits a..63: rm >> a has 64 - a ones and zeros above)
                // exact sequential fp64 sums: S = sum v_i, T = sum of the running S = sum (len - i) v_i, so sum i v_i = len S - T
                double S = 0.0, T = 0.0;
                if (!from_global) {
                    const uint32_t off = (ub & 0xffffu) + (uint32_t)popc64((rm | s_mask[rq ^ 1][ru]) & bits_below(a));
                    int k = 0;
                    for (; k + 2 <= len; k += 2) {   // two parked values in flight
                        const float v0 = s_val[off + k], v1 = s_val[off + k + 1];
                        S += (double)v0; T += S;
                        S += (double)v1; T += S;
                    }
                    if (k < len) { S += (double)s_val[off + k]; T += S; }
                } else {   // dense tile: the values were not parked; re-read from L2
                    const float *rowptr = dens + ((int64_t)(s0 + rsl) * nr + (r0 + rrl)) * nc + ctile + rwl * 64;
                    for (int k = 0; k < len; ++k) { S += (double)rowptr[a + k]; T += S; }
                }
                const int p0 = rwl * 64 + a;
                // the run's sums become integers here (rounded once to the job's quantum); moments relative to the tile origin
                const long long F = fix_of(S, fix_mul), Fc = fix_of((double)p0 * S + ((double)len * S - T), fix_mul);
                atomicAdd(&s_rho[comp], (unsigned long long)F);
                atomicAdd(&s_rho_c[comp], (unsigned long long)Fc);
                atomicAdd(&s_rho_r[comp], (unsigned long long)(F * rrl));
                atomicAdd(&s_rho_s[comp], (unsigned long long)(F * rsl));
                // (keys of a plane are below 2^31: the host checks)
                atomicMin(&s_key[comp], ((uint32_t)rq << 31) | ((uint32_t)(ctile + p0) * urus + (uint32_t)(r0 + rrl) * (uint32_t)us + (uint32_t)(s0 + rsl)));
                const uint32_t ulen = (uint32_t)len;
                atomicAdd(&s_pk[comp], (unsigned long long)ulen | ((unsigned long long)(ulen * (uint32_t)rrl) << 20) | ((unsigned long long)(ulen * (uint32_t)rsl) << 40));
                atomicAdd(&s_crel[comp], ulen * (uint32_t)p0 + ulen * (ulen - 1u) / 2u);
                comp_of_run[i] = gb + comp_all;
                const uint32_t kw = i - (ub >> 16);   // my place among the word-runs of my word
                uint8_t *rec = word_comps + ((desc & 0x1ffu) << 3);
                if (kw < (uint32_t)FACE_K) rec[kw] = (uint8_t)comp;
                if (a + len == 64) rec[7] = (uint8_t)comp;
            }
            if (tid == 0 && pass0 == 0u) {
                lj.tile_mode[tile_id] = wide ? 2 : 0;
                if (wide) *lj.unit_flag = lj.epoch;
            }
        }
        __syncthreads();   // ---- barrier 4 ----
        const uint32_t n_here = n_comp - pass0 < (uint32_t)CCAP ? n_comp - pass0 : (uint32_t)CCAP;
        for (uint32_t i = tid; i < n_here; i += NT) {
            const uint32_t g = gb + pass0 + i;
            const unsigned long long pk = s_pk[i];
            const long long n = (long long)(pk & 0xfffffull);
            lj.r_n[g] = (uint32_t)n;
            fix_store(lj, g, fix_sums((long long)s_rho[i], (long long)s_rho_c[i], (long long)s_rho_r[i], (long long)s_rho_s[i], w0 * 64, r0, s0));
            lj.r_c[g] = (long long)s_crel[i] + n * (w0 * 64);
            lj.r_r[g] = (long long)((pk >> 20) & 0xfffffull) + n * r0;
            lj.r_s[g] = (long long)(pk >> 40) + n * s0;
            const uint32_t key = s_key[i];
            lj.r_key[g] = (unsigned long long)((key >> 31) ? keys_pp : 0) + (key & 0x7fffffffu);
            lj.kpar[g] = ((unsigned long long)key << 32) | g;   // a root, named by its first key: the cross-tile unions hang the later first voxel under the earlier
            if (wide) lj.parent[g] = (int32_t)g;                 // (ids above the tiles' ranges: k_resolve_tiles writes parent[] of non-roots only)
        }
        if (pass0 + (uint32_t)CCAP >= n_comp) break;
        __syncthreads();   // (the accumulators have been read)
        for (uint32_t i = tid; i < (uint32_t)CCAP; i += NT) {
            s_rho[i] = 0ull; s_rho_c[i] = 0ull; s_rho_r[i] = 0ull; s_rho_s[i] = 0ull;
            s_pk[i] = 0ull; s_crel[i] = 0u; s_key[i] = 0xffffffffu;
        }
        __syncthreads();
    }
    mark_comps_unused(lj, cb, wide ? 0u : n_comp, tid, NT);
    if (tid == 0) lj.tile_runs[bid] = n_runs;
}

// Cross-tile pairs of one mask word.  All global loads (the 13 neighbour masks and run bases)
// are issued up front and unconditionally -- one memory latency instead of one per neighbour --
// then the touching RUN pairs are enumerated from registers.  Pairs inside one normally
// processed tile were already united in LDS and are skipped (their neighbour mask is zeroed).
// emit(runA, runB) with global run ids (the caller maps them to components).
struct NbWords {
    uint64_t m[13];     // [0] = previous word of my row; [1 + nb*3 + (dw+1)] = neighbour rows
    uint32_t base[13];
};

// Which of a word's pairs a caller wants (the tiles' modes: 0 united in LDS, 2 wide -- united in LDS, components by run id --,
// 1 / 3 unit tiles, labelled by k_unit_label):
//   PAIRS_ALL   every cross-tile pair (the generic path)
//   PAIRS_WIDE  the pairs with a WIDE tile on either side and no unit tile on the other: everything they need was written by
//               k_tile_label, so k_face_merge unites them itself
//   PAIRS_UNIT  the pairs with a UNIT tile on either side: k_unit_pairs, behind k_unit_label
enum { PAIRS_ALL = 0, PAIRS_WIDE = 1, PAIRS_UNIT = 2 };
template <typename JobRef>
__device__ inline bool load_cross_tile(const JobRef &job, const TileDims &td, int64_t w, uint64_t m, NbWords &nw, uint32_t &my_base, int which = PAIRS_ALL) {
    const int plane = (job.n_vols > 1 && w >= job.vols[1].word_base) ? 1 : 0;
    const VolDesc vd = job.vols[plane];
    const int64_t rem = w - vd.word_base;
    const int wq = (int)(rem % vd.row_words);
    const int64_t row = rem / vd.row_words;
    const int rl = (int)(row % vd.dim[1]);
    const int sl = (int)(row / vd.dim[1]);
    const uint32_t my_mode = job.tile_mode[tile_index(td, 0, wq, rl, sl)];
    const bool in_lds = my_mode == 0u || my_mode == 2u;   // a WIDE tile (2) united its own pairs in LDS, like a normal one
    if (which == PAIRS_WIDE && (my_mode & 1u)) return false;   // (a unit tile's pairs: all of them the other pass's)
    // does any of the 13 earlier neighbours live in ANOTHER tile?
    const bool edge = ((rl & 7) == 0 && rl > 0) || ((sl & 7) == 0 && sl > 0) || ((rl & 7) == 7 && rl + 1 < vd.dim[1] && sl > 0) ||
                      (wq % td.cw == 0 && wq > 0) || (wq % td.cw == td.cw - 1 && wq + 1 < vd.row_words && (rl > 0 || sl > 0));
    if (!edge && in_lds) return false;
    // does the pair with a neighbour in a tile of mode `nm` belong to this pass?
    auto wanted = [&](uint32_t nm) {
        if (which == PAIRS_ALL) return true;
        if (which == PAIRS_UNIT) return ((my_mode | nm) & 1u) != 0u;
        return (my_mode == 2u || nm == 2u) && (nm & 1u) == 0u;
    };
    const bool all_nb = which == PAIRS_ALL || (which == PAIRS_UNIT && (my_mode & 1u));   // otherwise a neighbour counts by ITS tile's mode
    bool want[13];
    int64_t at[13];
    want[0] = (m & 1ull) && wq > 0 && (!in_lds || (wq % td.cw == 0));
    if (want[0] && !all_nb) want[0] = wanted(job.tile_mode[tile_index(td, 0, wq - 1, rl, sl)]);
    at[0] = w - 1;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int dr = nb == 2 ? 0 : (nb == 3 ? 1 : -1);
        const int ds = nb == 0 ? 0 : -1;
        const int r2 = rl + dr, s2 = sl + ds;
        const bool row_ok = r2 >= 0 && r2 < vd.dim[1] && s2 >= 0;
        const bool row_same_tile = ((r2 >> 3) == (rl >> 3)) && ((s2 >> 3) == (sl >> 3));
        const int64_t rowbase = vd.word_base + ((int64_t)s2 * vd.dim[1] + r2) * vd.row_words;
#pragma unroll
        for (int dw = -1; dw <= 1; ++dw) {
            const int w2 = wq + dw;
            const int i = 1 + nb * 3 + (dw + 1);
            bool ok = row_ok && w2 >= 0 && w2 < vd.row_words;
            if (ok && in_lds && row_same_tile && (w2 / td.cw == wq / td.cw)) ok = false;  // united in LDS
            if (dw < 0 && !(m & 1ull)) ok = false;
            if (dw > 0 && !(m >> 63)) ok = false;
            if (ok && !all_nb) ok = wanted(job.tile_mode[tile_index(td, 0, w2, r2, s2)]);
            want[i] = ok;
            at[i] = rowbase + w2;
        }
    }
    bool any = false;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int64_t a = want[i] ? at[i] : w;  // unconditional load (own word when unwanted)
        nw.m[i] = job.mask[a];
        nw.base[i] = job.run_base[a];
    }
    my_base = job.run_base[w];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        if (!want[i]) nw.m[i] = 0ull;
        any = any || nw.m[i] != 0ull;
    }
    return any;
}

template <typename F>
__device__ inline void cross_tile_pairs(uint64_t m, uint32_t my_base, const NbWords &nw, F &&emit) {
    const uint64_t mstarts = run_starts(m);
    const uint64_t dil = m | (m << 1) | (m >> 1);
    if (nw.m[0] >> 63) {
        const uint64_t pm = nw.m[0];
        emit(my_base, nw.base[0] + (uint32_t)popc64(run_starts(pm) & bits_below(run_start_of(pm, 63))));
    }
#pragma unroll
    for (int i = 1; i < 13; ++i) {
        const int dw = (i - 1) % 3 - 1;
        const uint64_t nm = nw.m[i];
        uint64_t hit = dw == 0 ? (nm & dil) : dw < 0 ? (nm & (1ull << 63)) : (nm & 1ull);
        while (hit) {
            const int qb = ctz64(hit);
            const int stb = run_start_of(nm, qb);
            const int enb = run_end_of(nm, stb);
            hit &= ~bits_below(enb + 1);
            const uint32_t other = nw.base[i] + (uint32_t)popc64(run_starts(nm) & bits_below(stb));
            int lo = stb - 1 + 64 * dw, hi = enb + 1 + 64 * dw;
            if (lo < 0) lo = 0;
            if (hi > 63) hi = 63;
            uint64_t mine = m & (bits_below(hi + 1) & ~bits_below(lo));
            while (mine) {
                const int pa = ctz64(mine);
                const int sta = run_start_of(m, pa);
                mine &= ~bits_below(run_end_of(m, sta) + 1);
                emit(my_base + (uint32_t)popc64(mstarts & bits_below(sta)), other);
            }
        }
    }
}

// Cross-tile unions, one workgroup per tile and ONE round trip to memory before the pairs are known.  A tile meets
// earlier tiles across its r / s faces in 46 pairs of rows (a row on a face, an earlier neighbour row in ANOTHER tile of the
// same c column); a lane owns one mask word of such a pair of rows and loads, all at once, the word of either row, the two
// words' component records (k_tile_label: the tile-local components of a word's first FACE_K word-runs, a byte each) and the
// two tile modes.  The touching word-runs then come out of bit arithmetic (as inside k_tile_label): with S = the word-local
// run starts, every touching pair shows once, at the later of its two starts --
//     EA = SA & (B | B << 1 | carry B)      EB = SB & (A << 1 | carry A)
// and at a set bit p both runs are "the last start <= p" of their row: popc(S & bits up to p) numbers them inside the word,
// count 0 = the run that ends the previous word of the row (one shuffle brings its component).  No run lists, no merge
// loop: the two-pointer merge of two exported run lists that this replaces spent 7.6 us walking them and 4.2 us loading
// them behind a first trip for their offsets (r03 stamps: 20.4 us per workgroup, now the trip, ~1 us of arithmetic and the
// unions).  The distinct COMPONENT pairs of the tile go into an LDS hash set (the runs of a blob cross a face row after row:
// ~350 touching run pairs per tile are ~110 distinct component pairs), and the set's slots are then united in the global
// union-find, about one pair per thread -- no pair buffer in HBM, no second launch.  Pairs with a unit tile on either side
// belong to the unit path (skipped here by the tile modes).  Grids wider than one tile add the c faces (waves 6 / 7): the
// run at position 0 of a row against the runs that end the 9 rows around it in the tile to the left.
//
// Tiles that overflowed LDS in k_tile_label ("unit tiles") are not this kernel's work (k_unit_label / k_unit_pairs, behind it,
// for the jobs that have any): here their pairs are skipped, and a job enqueued without the two launches is flagged for a second run.
constexpr int PAIR_SLOTS = 1024; // LDS hash set of a tile's distinct cross-face component pairs (a full set unites on the spot)
constexpr int FM_THREADS = 384, FM_THREADS_WIDE = 512;  // 2 signs x 46 pairs of rows x 4 word slots = 368 lanes; grids wider than a tile: 128 more for the c faces
constexpr int FACE_PAIRS = 46;

// One mask word of the unit-tile companion of k_face_merge: it owns EVERY pair that has a unit tile on either side
// (all rows), and unites on the spot.
template <typename JobRef>
__device__ void unit_edges_word(const JobRef &job, const TileDims &td, const VolDesc &v0, int plane, int sl, int r, int wq, int which) {
    const int rw = v0.row_words;
    const int64_t w = (int64_t)plane * rw * v0.dim[1] * v0.dim[2] + ((int64_t)sl * v0.dim[1] + r) * rw + wq;
    const uint64_t m = job.mask[w];
    if (m == 0ull) return;
    NbWords nw;
    uint32_t my_base;
    if (!load_cross_tile(job, td, w, m, nw, my_base, which)) return;
    cross_tile_pairs(m, my_base, nw, [&](uint32_t a, uint32_t b) { kuf_unite(job.kpar, job.comp_of_run[a], job.comp_of_run[b]); });
}

template <int CW, int NTH>
__global__ void __launch_bounds__(NTH, 8) k_face_merge(Job job_arg, const float *__restrict__ dens, const Geom *__restrict__ gp, TileDims td, int pair_slots) {
    // (the job is read from the kernel-argument segment where it is used: the ~45 pointers of a by-value Job sat in scalar registers
    //  -- and spilled into vector lanes -- through the whole kernel for the sake of its cold tail)
    kernarg_prefetch();
    PDBEDA_LATE_JOB(lj);
    /*@F0*/
    const bool any_unit = *lj.unit_flag == lj.epoch;   // block-uniform: some tile of this job is a unit tile (rare)
    static_assert(FACE_K == 7, "a word's component record is one 64-bit load: seven runs and the run at the last bit");
    __shared__ unsigned long long s_set[PAIR_SLOTS], s_pairs[PAIR_SLOTS];
    __shared__ uint32_t s_wsum[NTH / 64];
    // what kpar[] holds for the components of this tile and of the four tiles it meets across its r / s faces, fetched with
    // everything else: a union starts from two VALUES (an ancestor-or-self of either component, named by its first key), and
    // these are such values whatever the other workgroups have united meanwhile -- no trip for them in front of the first hook
    constexpr int KPH = 128;   // ids of a tile whose packed parents ride in the first trip (a tile has ~100 components: the others are fetched on demand; r04 A/B: -0.6 us, -5 MB)
    __shared__ kp_t s_kp[5][KPH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int ur = td.ur, us = td.us, row_words = td.row_words;
    // (3-D grid: the tile coordinates come with the workgroup, x fastest = tile order)
    const int ct = (int)blockIdx.x, rt = (int)blockIdx.y, st = (int)blockIdx.z;
    const int tile = (st * td.rtiles + rt) * td.ctiles + ct;
    const int w0 = ct * CW;
    const int64_t plane_words = (int64_t)row_words * ur * us;
    const unsigned long long *comps64 = reinterpret_cast<const unsigned long long *>(lj.word_comps);
    auto word_at = [&](int q, int r, int s, int wq) { return (int64_t)q * plane_words + ((int64_t)s * ur + r) * row_words + wq; };
    auto comps_at = [&](uint32_t tl, int q, int r, int s, int wl) { return ((size_t)tl * 2 + q) * 256 + ((s & 7) * TILE_R + (r & 7)) * CW + wl; };
    // component of word-run k of mask word w by its run id (words with more than FACE_K runs: two dependent loads, rare)
    auto comp_by_run = [&](int64_t w, uint32_t k) { return lj.comp_of_run[lj.run_base[w] + k]; };

    // ---- the one trip: everything a lane needs, issued before anything waits ----
    // r / s faces: lanes 0 .. 367 = sign x pair of rows x word slot
    const int fq = tid / (FACE_PAIRS * 4), frem = tid % (FACE_PAIRS * 4), ft = frem >> 2, fwl = frem & 3;
    int rl = 0, sl = 0, dr = 0, ds = -1;
    bool task = tid < 2 * FACE_PAIRS * 4 && fq < td.n_planes && fwl < CW && w0 + fwl < row_words;
    if (ft < 8) { rl = 0; sl = ft; dr = -1; ds = 0; }
    else if (ft < 16) { rl = 0; sl = ft - 8; dr = -1; }
    else if (ft < 23) { rl = ft - 15; sl = 0; dr = -1; }
    else if (ft < 31) { rl = ft - 23; sl = 0; dr = 0; }
    else if (ft < 39) { rl = ft - 31; sl = 0; dr = 1; }
    else { rl = 7; sl = ft - 38; dr = 1; }
    const int r = rt * TILE_R + rl, s = st * TILE_S + sl, r2 = r + dr, s2 = s + ds;
    if (r >= ur || s >= us || r2 < 0 || r2 >= ur || s2 < 0) task = false;
    const uint32_t tile_b = task ? (uint32_t)(((s2 >> 3) * td.rtiles + (r2 >> 3)) * td.ctiles + ct) : 0u;   // the r / s faces join tiles of one c column
    const int64_t wA = task ? word_at(fq, r, s, w0 + fwl) : 0, wB = task ? word_at(fq, r2, s2, w0 + fwl) : 0;
    uint64_t mA = 0ull, mB = 0ull;
    unsigned long long cA = 0ull, cB = 0ull;
    uint32_t modes = 0u;
    if (task) {
        mA = lj.mask[wA]; mB = lj.mask[wB];
        cA = comps64[comps_at((uint32_t)tile, fq, r, s, fwl)]; cB = comps64[comps_at(tile_b, fq, r2, s2, fwl)];
        modes = (uint32_t)lj.tile_mode[tile] | (uint32_t)lj.tile_mode[tile_b];
    }
    // the five tables: this tile, (rt - 1, st), (rt - 1, st - 1), (rt, st - 1), (rt + 1, st - 1)
    uint32_t tiles5[5];
    {
        const int drs[5] = {0, -1, -1, 0, 1}, dss[5] = {0, 0, -1, -1, -1};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int r5 = rt + drs[j], s5 = st + dss[j];
            tiles5[j] = (r5 >= 0 && r5 < td.rtiles && s5 >= 0) ? (uint32_t)((s5 * td.rtiles + r5) * td.ctiles + ct) : 0xffffffffu;
        }
    }
    constexpr int KPL = (5 * KPH + NTH - 1) / NTH;
    kp_t kp_pre[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int e = tid + k * NTH, j = e / KPH;
        const uint32_t t5 = j == 0 ? tiles5[0] : (j == 1 ? tiles5[1] : (j == 2 ? tiles5[2] : (j == 3 ? tiles5[3] : tiles5[4])));
        kp_pre[k] = (e < 5 * KPH && t5 != 0xffffffffu) ? kuf_load(lj.kpar, t5 * (uint32_t)CCAP + (uint32_t)(e % KPH)) : KP_UNUSED;
    }
    {   // every tile clears its slice of the first-key bitmap and of the rank counters (saves a memset launch; painted by the next
        // kernel) and its inbox counter -- HERE, under the first trip: fire-and-forget stores, and the barriers of this kernel's hot
        // path wait for LDS only (lds_barrier), so nothing ever waits for them (r03 had them behind the last barrier: 0.9 us at
        // the end of every workgroup; in front of a __syncthreads() they cost 2 us)
        const int64_t key_words = lj.key_words;
        const int64_t lo = (int64_t)lj.clear_bits * tile, hi = lo + lj.clear_bits < key_words ? lo + lj.clear_bits : key_words;
        for (int64_t i = lo + tid; i < hi; i += NTH) lj.key_bits[i] = 0ull;
        const int64_t nfc = lj.n_fine_alloc / 2;   // (16-bit counters, two per word)
        const int64_t clo = (int64_t)lj.clear_fine * tile, chi = clo + lj.clear_fine < nfc ? clo + lj.clear_fine : nfc;
        for (int64_t i = clo + tid; i < chi; i += NTH) lj.fine_count[i] = 0u;
        const int64_t nmid = (key_words + KEY_FINE - 1) / KEY_FINE * (KEY_FINE / 16);   // (a byte per 4 key words, four per word; whole buckets)
        const int64_t mlo = (int64_t)lj.clear_mid * tile, mhi = mlo + lj.clear_mid < nmid ? mlo + lj.clear_mid : nmid;
        for (int64_t i = mlo + tid; i < mhi; i += NTH) lj.mid_count[i] = 0u;
        if (tid == 0) lj.inbox_count[(size_t)tile * INBOX_STRIDE] = 0u;
    }
    for (int i = tid; i < pair_slots; i += NTH) s_set[i] = 0ull;   // 0 = empty: a pair (lo << 32 | hi) has hi > lo >= 0
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int e = tid + k * NTH;
        if (e < 5 * KPH) (&s_kp[0][0])[e] = kp_pre[k];
    }
    lds_barrier();
    /*@F1*/
    auto val_of = [&](uint32_t id) -> kp_t {   // a value that names a node of id's set
        const uint32_t t = id / (uint32_t)CCAP;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (t == tiles5[j] && (id % (uint32_t)CCAP) < (uint32_t)KPH) return s_kp[j][id % (uint32_t)CCAP];
        return kuf_load(lj.kpar, id);   // (the c faces of a grid wider than one tile)
    };
    const uint32_t slot_mask = (uint32_t)pair_slots - 1u;
    auto add_pair = [&](uint32_t ca, uint32_t cb) {
        const uint32_t lo = ca < cb ? ca : cb, hi = ca < cb ? cb : ca;
        const unsigned long long key = ((unsigned long long)lo << 32) | hi;
        uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & slot_mask;
        for (int probe = 0; probe < 8; ++probe, h = (h + 1u) & slot_mask) {
            const unsigned long long old = atomicCAS(&s_set[h], 0ull, key);
            if (old == 0ull || old == key) return;
        }
        kuf_hook_vals(lj.kpar, val_of(lo), val_of(hi));   // the neighbourhood of the slot is full: unite on the spot
    };
    {   // ---- r / s faces ----
        if (modes != 0u) { mA = 0ull; mB = 0ull; }   // a unit tile on either side: the unit path owns the pair
        const uint64_t SA = mA & ~(mA << 1), SB = mB & ~(mB << 1);   // word-local run starts
        // what the next word of the row needs of me: do my rows end set, and the components of the runs that end them
        const uint32_t pack = (uint32_t)(mA >> 63) | ((uint32_t)(mB >> 63) << 1) | ((uint32_t)(cA >> 56) << 8) | ((uint32_t)(cB >> 56) << 16);
        uint32_t prev = dpp0<DPP_WAVE_SHR1>(pack);   // (the four word slots of a pair of rows are four lanes of one wave; lane 0 reads 0)
        if (fwl == 0) prev = 0u;
        const uint64_t carA = prev & 1u, carB = (prev >> 1) & 1u;
        uint64_t todo = (SA & (mB | (mB << 1) | carB)) | (SB & ((mA << 1) | carA));
        const uint32_t baseA = (uint32_t)tile * CCAP, baseB = tile_b * CCAP;
        uint32_t last_a = ~0u, last_b = ~0u;
        while (todo) {
            const int p = ctz64(todo);
            todo &= todo - 1ull;
            const uint64_t upto = (2ull << p) - 1ull;   // (p = 63: all ones)
            const uint32_t ka = (uint32_t)popc64(SA & upto), kb = (uint32_t)popc64(SB & upto);
            uint32_t ca, cb;
            if (ka == 0u) ca = baseA + ((prev >> 8) & 0xffu);   // the run that ends the previous word
            else ca = ka <= (uint32_t)FACE_K ? baseA + (uint32_t)((cA >> ((ka - 1u) * 8u)) & 0xffull) : comp_by_run(wA, ka - 1u);
            if (kb == 0u) cb = baseB + ((prev >> 16) & 0xffu);
            else cb = kb <= (uint32_t)FACE_K ? baseB + (uint32_t)((cB >> ((kb - 1u) * 8u)) & 0xffull) : comp_by_run(wB, kb - 1u);
            if (ca != last_a || cb != last_b) {   // the runs of a blob cross a face in a row: repeats are the rule
                add_pair(ca, cb);
                last_a = ca; last_b = cb;
            }
        }
    }
    // ---- c faces (grids wider than one tile), lanes 384 .. 511 = sign x row: row `lane` of this tile may start with a run at
    // position 0; the rows around it (9 offsets, itself included) in the tile to the LEFT may end with a run at that tile's
    // last position -- they touch.  (Loaded after the barrier: the registers of the r / s lanes are free by now, and these
    // waves have nothing else to wait for.)
    const int cq = (tid - 384) >> 6, crl = lane & 7, csl = lane >> 3;
    const int cr = rt * TILE_R + crl, cs = st * TILE_S + csl;
    const bool ctask = NTH > 384 && tid >= 384 && ct > 0 && cq < td.n_planes && cr < ur && cs < us;
    const uint32_t *mask32 = reinterpret_cast<const uint32_t *>(lj.mask);
    uint32_t m0 = 0u, c0 = 0u, cmodes = 0u, hi9[9], c9[9], mode9[9];   // low half of my word, high halves of theirs; record bytes 0 / 7
#pragma unroll
    for (int k = 0; k < 9; ++k) { hi9[k] = 0u; c9[k] = 0u; mode9[k] = 0u; }
    auto tile9 = [&](int k) { return (uint32_t)((((cs + k / 3 - 1) >> 3) * td.rtiles + ((cr + k % 3 - 1) >> 3)) * td.ctiles + ct - 1); };
    if (ctask) {
        m0 = mask32[2 * word_at(cq, cr, cs, w0)];
        c0 = lj.word_comps[comps_at((uint32_t)tile, cq, cr, cs, 0) * 8];
        cmodes = lj.tile_mode[tile];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int r9 = cr + k % 3 - 1, s9 = cs + k / 3 - 1;
            if (r9 < 0 || r9 >= ur || s9 < 0 || s9 >= us) continue;
            hi9[k] = mask32[2 * word_at(cq, r9, s9, w0 - 1) + 1];
            c9[k] = lj.word_comps[comps_at(tile9(k), cq, r9, s9, CW - 1) * 8 + 7];
            mode9[k] = lj.tile_mode[tile9(k)];
        }
    }
    if (ctask && (m0 & 1u) && cmodes == 0u) {
        const uint32_t ca = (uint32_t)tile * CCAP + c0;   // (a run at position 0 is the first of its word)
        uint32_t last_b = ~0u;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (!(hi9[k] >> 31) || mode9[k] != 0u) continue;   // (a unit tile: the unit path unites)
            const uint32_t cb = tile9(k) * CCAP + c9[k];
            if (cb != last_b) { add_pair(ca, cb); last_b = cb; }
        }
    }
    lds_barrier();
    /*@F2*/
    // the distinct pairs of this tile: compacted (block prefix over the slots) so that thread k unites pair k, k + 512, ... --
    // a union is a chain of dependent memory round trips, and nobody should walk two chains while others walk none
    // (parents start as the identity, so most unions are two parallel loads and one atomic min)
    {
        constexpr int MAXPER = (PAIR_SLOTS + NTH - 1) / NTH;
        const int per = (pair_slots + NTH - 1) / NTH;   // consecutive slots per thread
        unsigned long long mine[MAXPER];
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < MAXPER; ++k) {
            const int slot = tid * per + k;
            mine[k] = (k < per && slot < pair_slots) ? s_set[slot] : 0ull;
            cnt += mine[k] != 0ull ? 1u : 0u;
        }
        const uint32_t x = wave_scan(cnt);   // (DPP: six adds; the shuffle form is six trips through the LDS crossbar)
        if (lane == 63) s_wsum[tid >> 6] = x;
        lds_barrier();
        uint32_t at = x - cnt;
        for (int k = 0; k < (tid >> 6); ++k) at += s_wsum[k];
#pragma unroll
        for (int k = 0; k < MAXPER; ++k)
            if (mine[k] != 0ull) s_pairs[at++] = mine[k];
        lds_barrier();
    }
    /*@F3*/
    uint32_t n_pairs = 0;
#pragma unroll
    for (int k = 0; k < NTH / 64; ++k) n_pairs += s_wsum[k];
    for (uint32_t k = tid; k < n_pairs; k += NTH) {
        const unsigned long long key = s_pairs[k];
        kuf_hook_vals(lj.kpar, val_of((uint32_t)(key >> 32)), val_of((uint32_t)key));
    }
    /*@F4*/
    /*@F5*/
    if (any_unit) {
        // Some tile left the fast path in k_tile_label.  A WIDE tile (more than CCAP components) is complete -- united in LDS, its
        // components under global ids: the pairs it has across its faces are united here, word by word (by the workgroup of the
        // tile that holds the later word).  A UNIT tile (more word-runs than LDS holds: checkerboards) is not labelled yet: that,
        // and every pair with such a tile on either side, is the work of k_unit_label / k_unit_pairs, two launches behind this one
        // that the host enqueues only for a job known to need them (Job::unit_form) -- a job enqueued without them says so here
        // and is run again.  Nobody waits for another workgroup (rounds 3-4: both phases ran here, behind per-tile flags the
        // workgroups polled).
        const VolDesc v0 = lj.vols[0];
        constexpr int NU = 64 * CW;
        for (int k = tid; k < td.n_planes * NU; k += NTH) {
            const int plane = k / NU, u = k % NU, wl = u % CW, rowl = u / CW;
            const int r = rt * TILE_R + (rowl & 7), s = st * TILE_S + (rowl >> 3), wq = w0 + wl;
            if (r < ur && s < us && wq < row_words) unit_edges_word(lj, td, v0, plane, s, r, wq, PAIRS_WIDE);
        }
        if (lj.unit_form == 0 && tid == 0 && lj.unit_flag[2] == lj.epoch) atomicOr(&lj.ctr->overflow, 4u);
    }
}

// The two launches of the unit path (Job::unit_form 1), one workgroup per tile each.
// k_unit_label: a unit tile is labelled run by run (every run its own component: unit_label_tile).
template <int CW>
__global__ void __launch_bounds__(512) k_unit_label(Job job_arg, const float *__restrict__ dens, const Geom *__restrict__ gp, TileDims td) {
    PDBEDA_LATE_JOB(lj);
    __shared__ __attribute__((aligned(16))) unsigned char s_scratch[1024];
    const int ct = (int)blockIdx.x, rt = (int)blockIdx.y, st = (int)blockIdx.z;
    const int tile = (st * td.rtiles + rt) * td.ctiles + ct;
    if ((lj.tile_mode[tile] & 1) == 0) return;   // block-uniform: 1 / 3 (a wide tile, 2, was labelled by k_tile_label)
    unit_label_tile<CW>(lj, dens, gp, td, ct * CW, rt * TILE_R, st * TILE_S, s_scratch);
}
// k_unit_pairs: every pair with a unit tile on either side is united, word by word, by the workgroup of the tile that holds the
// LATER word of the pair: all words of a unit tile, else the words on the tile's faces that look at one (unit_edges_word).
template <int CW>
__global__ void __launch_bounds__(512) k_unit_pairs(Job job_arg, TileDims td) {
    PDBEDA_LATE_JOB(lj);
    const int tid = threadIdx.x;
    const int ct = (int)blockIdx.x, rt = (int)blockIdx.y, st = (int)blockIdx.z;
    // (a unit tile that found no room for its ids raised the flag: the job's results are void, nothing is looked up)
    if (__hip_atomic_load(&lj.ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const VolDesc v0 = lj.vols[0];
    constexpr int NU = 64 * CW;
    for (int k = tid; k < td.n_planes * NU; k += 512) {
        const int plane = k / NU, u = k % NU, wl = u % CW, rowl = u / CW;
        const int r = rt * TILE_R + (rowl & 7), s = st * TILE_S + (rowl >> 3), wq = ct * CW + wl;
        if (r < td.ur && s < td.us && wq < td.row_words) unit_edges_word(lj, td, v0, plane, s, r, wq, PAIRS_UNIT);
    }
}

// Whole-map k_resolve: one workgroup per tile (its CCAP component ids).  Every component finds its root -- the component of
// its blob with the smallest first key (the unions hang by key) -- so a ROOT knows the blob's first key without hearing from
// anybody: it paints that key into the bitmap and bumps its rank counter here (round 4: a kernel of its own before, behind
// the fold of the members' keys).  The non-root components of a tile that share a root are first summed in LDS (a small hash
// table keyed by the root), then ONE record per (tile, root) is POSTED to the inbox of the tile that owns the root -- a
// returning atomic on a counter with a cache line of its own + plain stores; k_emit_tiles / the label writer absorb the
// inbox.  A blob that spans the map (the chain of a protein at 1.5 sigma: tens of thousands of tile components) would
// otherwise pile 9 same-address atomics per component on one record.  All loads of a component's record are issued up
// front, beside the first step of the find (the kernel is a chain of dependent memory round trips, not bandwidth).
// Workgroups beyond the tiles handle the components of unit tiles, 256 at a time.
constexpr int RSLOTS = 256;   // LDS slots for the distinct roots the members of one tile fold into (<= 256 members: always enough)
__global__ void __launch_bounds__(256) k_resolve_tiles(Job job, int n_tiles) {
    static_assert(CCAP == 256, "one thread per component id of a tile");
    const int tid = threadIdx.x;
    auto fold = [&](uint32_t root, uint32_t n, const FixSums &sum, unsigned long long c, unsigned long long r, unsigned long long s) {
        atomicAdd(&job.r_n[root], n);
        fix_fold(job, root, sum);
        atomicAdd((unsigned long long *)&job.r_c[root], c);
        atomicAdd((unsigned long long *)&job.r_r[root], r);
        atomicAdd((unsigned long long *)&job.r_s[root], s);
    };
    auto paint = [&](unsigned long long key) { paint_key(job, key); };
    __shared__ int s_root[RSLOTS];
    __shared__ FixSums s_f[RSLOTS];
    __shared__ unsigned long long s_i[3][RSLOTS];
    __shared__ uint32_t s_cnt[RSLOTS];
    auto clear_table = [&]() {
        for (int k = tid; k < RSLOTS; k += 256) {
            s_root[k] = -1;
            s_f[k] = fix_zero();
            s_i[0][k] = 0ull; s_i[1][k] = 0ull; s_i[2][k] = 0ull; s_cnt[k] = 0u;
        }
    };
    auto table_add = [&](int root, uint32_t n, const FixSums &sum, unsigned long long c, unsigned long long r, unsigned long long s) {
        uint32_t h = ((uint32_t)root * 2654435761u) >> 24;   // 8 bits
        while (true) {                                         // (<= 256 members, 256 slots: a slot always turns up)
            const int old = atomicCAS(&s_root[h], -1, root);
            if (old == -1 || old == root) break;
            h = (h + 1u) & (RSLOTS - 1);
        }
        atomicAdd(&s_cnt[h], n);
        fix_atomic_add(&s_f[h], sum);
        atomicAdd(&s_i[0][h], c);
        atomicAdd(&s_i[1][h], r);
        atomicAdd(&s_i[2][h], s);
    };
    if ((int)blockIdx.x >= n_tiles) {
        // the components of unit tiles (every run its own component): 256 at a time, and the same pre-reduction -- the members of
        // a round that share a root are summed in the LDS table and folded with ONE set of atomics.  Folding them one by one put
        // hundreds of thousands of same-address atomics on the record of a map-spanning blob (a protein-like map at 0.5 sigma:
        // 2.3 ms in this kernel alone).
        if (job.ctr->overflow != 0u) return;   // (the unit tiles ran out of ids: this job's results are void, the host runs it again)
        const uint32_t n_comp = n_components(job), stride = (gridDim.x - (uint32_t)n_tiles) * 256u;
        for (uint32_t base = (uint32_t)n_tiles * CCAP + (blockIdx.x - (uint32_t)n_tiles) * 256u; base < n_comp; base += stride) {   // block-uniform
            clear_table();
            const uint32_t i = base + tid;
            int root = -1;
            if (i < n_comp) {
                const kp_t rp = kuf_find_from(job.kpar, kuf_load(job.kpar, i));
                if (kp_id(rp) == i) paint(job.r_key[i]);   // (a run has voxels: a root among the unit components is a blob)
                else { root = (int)kp_id(rp); job.parent[i] = root; job.kpar[i] = rp; }
            }
            __syncthreads();
            if (root >= 0)
                table_add(root, job.r_n[i], fix_load(job, i), (unsigned long long)job.r_c[i], (unsigned long long)job.r_r[i], (unsigned long long)job.r_s[i]);
            __syncthreads();
            for (int k = tid; k < RSLOTS; k += 256)
                if (s_root[k] >= 0) fold((uint32_t)s_root[k], s_cnt[k], s_f[k], s_i[0][k], s_i[1][k], s_i[2][k]);
            __syncthreads();   // (the next round clears the table)
        }
        return;
    }
    /*@R0*/
    const uint32_t i = (uint32_t)blockIdx.x * CCAP + tid;
    // everything this thread may need, in flight at once -- for the lower half of the tile's ids (unused ones hold stale bytes:
    // loaded, never used).  A tile has ~100 components: ids 128 .. 255 are rarely in use, and their records (100 bytes each)
    // were 13 MB of reads per step for nothing: those threads load theirs once kpar[] says the id is in use (r04 A/B: -1.1 us)
    const kp_t p0 = kuf_load(job.kpar, i);
    uint32_t n_i = 0;
    FixSums v_sum = fix_zero();
    unsigned long long v_c = 0, v_r = 0, v_s = 0, v_key = 0;
    if (tid < 128 || p0 != KP_UNUSED) {
        n_i = job.r_n[i];
        v_sum = fix_load(job, i);
        v_c = (unsigned long long)job.r_c[i]; v_r = (unsigned long long)job.r_r[i]; v_s = (unsigned long long)job.r_s[i]; v_key = job.r_key[i];
    }
    __shared__ int s_any;
    clear_table();
    if (tid == 0) s_any = 0;
    lds_barrier();   // (early, and for LDS only: the table is clear before anybody has found anything -- a member adds itself as
                     //  soon as ITS find is done, and no barrier of this kernel waits for the stores and atomics in flight)
    const bool used = n_i > 0u && p0 != KP_UNUSED;
    const bool is_root = used && kp_id(p0) == i;
    const bool member = used && !is_root;   // non-root component with voxels
    int root = (int)i;
    /*@R1*/
    if (member) {
        const kp_t rp = kuf_find_from(job.kpar, p0);
        root = (int)kp_id(rp);
        job.kpar[i] = rp;
    }
    /*@R2*/
    job.parent[i] = root;   // (every id of the tile: the accessors and the label writer go through parent[])
    {   // the tile's roots as four ballots: k_emit_tiles maps its threads onto the set bits instead of scanning every component id
        const unsigned long long rbits = __ballot(is_root);
        if ((tid & 63) == 0) job.root_mask[(size_t)blockIdx.x * 4 + (tid >> 6)] = rbits;
    }
    if (is_root) paint(v_key);
    /*@R3*/
    if (member) { table_add(root, n_i, v_sum, v_c, v_r, v_s); s_any = 1; }
    lds_barrier();
    /*@R4*/
    if (s_any == 0) return;   // nothing to fold in this tile
    for (int k = tid; k < RSLOTS; k += 256) {
        if (s_root[k] < 0) continue;
        const uint32_t root = (uint32_t)s_root[k], rtile = root / CCAP;
        uint32_t pos = INBOX_CAP;
        if (rtile < (uint32_t)n_tiles) pos = atomicAdd(&job.inbox_count[(size_t)rtile * INBOX_STRIDE], 1u);   // (roots among the unit components have no inbox)
        if (pos < (uint32_t)INBOX_CAP) {
            InboxEntry e;
            e.local = root % CCAP; e.n = s_cnt[k];
            e.sum = s_f[k];
            e.c = s_i[0][k]; e.r = s_i[1][k]; e.s = s_i[2][k];
            job.inbox[(size_t)rtile * INBOX_CAP + pos] = e;
        } else {
            fold(root, s_cnt[k], s_f[k], s_i[0][k], s_i[1][k], s_i[2][k]);
        }
    }
    /*@R5*/
}

// Signed labels, one workgroup per tile, all look-ups in LDS: the tile's label-of-component table
// (<= CCAP ints), its run -> local component table (bytes) and its mask / run-base words; the 64 KiB
// of labels of a 256 x 8 x 8 tile are then streamed out with 16-B stores (1 KiB per wave and row).
// Unit tiles / tiles with too many runs take the global look-up path (same result).
// (Measured and dropped: two tiles per workgroup with the second tile's tables fetched under the first tile's stores --
//  27.8 us against 20.4; the store stream of a plain zero fill of the volume takes 10 us.)
//
// FUSED (round 4; the launch that labels a job as part of pdbeda_full_blobs*): the kernel also does what k_emit_tiles does.
// A component's root is named by its FIRST KEY (kpar[], compressed by k_resolve_tiles), and a label is the rank of that key:
// every tile ranks the roots of its own components itself (the prefix table over the rank counters is built by every
// workgroup, as in k_emit_tiles; rank_of_key is five loads), so nobody waits for a label table that another kernel wrote --
// the prologue stays at two memory round trips (kpar + masks + table | ranks + inbox), where a separate k_emit_tiles in
// front cost a launch, a table and three trips of its own.  Waves 4-7 meanwhile hold the records of the tile's own
// components: the roots among them add what the other tiles POSTED to the tile's inbox (summed in LDS) and write their blob
// table rows.  The non-fused form (labels asked for later, pdbeda_bloblist_labels) reads the label table k_emit_tiles wrote.
constexpr int LCAP = 4096;  // word-runs of a tile whose run -> component bytes fit the LDS table
#ifndef PDBEDA_LABELS_NT_THREADS
#define PDBEDA_LABELS_NT_THREADS 512
#endif
template <int CW, bool FUSED>
__global__ void __launch_bounds__(PDBEDA_LABELS_NT_THREADS, FUSED ? 8 : 1) k_labels_tiles(Job job_arg, TileDims td, int32_t *__restrict__ labels, const Geom *__restrict__ gp) {
    // the job is read from the kernel-argument segment where it is used (see PDBEDA_LATE_JOB): its ~45 pointers do not fit the
    // scalar registers beside the row loop -- passed by value they were copied to scratch at entry
    kernarg_prefetch();
    PDBEDA_LATE_JOB(lj);
    constexpr int NU = 64 * CW;
    constexpr int NTL = PDBEDA_LABELS_NT_THREADS, RPW = 64 / (NTL / 64);   // rows of the tile per wave
    __shared__ int32_t s_lab[CCAP];
    // per 32-bit half of a mask word (a lane's 4 voxels live in one half): the mask halves of both signs side by side, and
    // per sign the half's run starts beside the tile-local id of the run before its first start -- what a lane needs of a
    // word, as two 8-byte reads, with nothing left to compute per row that does not depend on the lane's own bits.
    // FUSED: 37.6 KiB in all -- four workgroups a CU (1 024 tiles: one round) leave nothing to spare.  The run -> component
    // bytes take the place of the rank prefix table once every rank has been taken (behind a barrier); the inbox accumulators
    // (20 KiB + 1 KiB) stay until the roots have written their rows, which waves 4-7 do while waves 0-3 already store labels.
    __shared__ __attribute__((aligned(16))) uint2 s_mh[2 * 256];
    __shared__ __attribute__((aligned(16))) uint2 s_sb[2][2 * 256];
    __shared__ __attribute__((aligned(16))) unsigned char s_pc[(LCAP + 8) > 4 * KEY_GROUPS ? (LCAP + 8) : 4 * KEY_GROUPS];
    uint8_t *s_comp8 = s_pc;                                   // [LCAP + 4]
    uint32_t *s_pre = reinterpret_cast<uint32_t *>(s_pc);      // [KEY_GROUPS] (fused)
    __shared__ unsigned long long s_acc[FUSED ? 10 : 1][FUSED ? CCAP : 1];   // the seven FixSums fields, sum c / r / s
    __shared__ uint32_t s_accn[FUSED ? CCAP : 1];
    __shared__ uint32_t s_wave[NTL / 64];
    __shared__ uint32_t s_vol0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const int uc = td.uc, ur = td.ur, us = td.us, row_words = td.row_words;
    // (a 3-D grid: the tile coordinates come with the workgroup -- two integer divisions here were ~50 scalar instructions in
    //  every wave, and the scalar unit of a CU serves all 32 of them.  Workgroups are dealt x fastest: tile order, as before)
    const int ct = (int)blockIdx.x, rt = (int)blockIdx.y, st = (int)blockIdx.z;
    const uint32_t bid = ((uint32_t)st * (uint32_t)td.rtiles + (uint32_t)rt) * (uint32_t)td.ctiles + (uint32_t)ct;
    /*@L0*/
    const int w0 = ct * CW, r0 = rt * TILE_R, s0 = st * TILE_S;
    const int64_t plane_words = (int64_t)row_words * ur * us;
    const uint32_t rb = (uint32_t)bid * (uint32_t)(NU * 32), cb = (uint32_t)bid * CCAP;
    // every load of the prologue is issued before anything depends on one: the tile's first 2048 run -> component ids and its
    // label table do not wait for tile_mode / tile_runs (ids beyond the tile's run count hold stale bytes: stored, never used)
    // (round 5: TWO unconditional loads per thread -- 1 024 runs; a tile at 1.5 sigma has ~790, at most ~990 -- where four, 2 048
    //  runs, read 8 KiB a tile of which 3 were ever used; tiles with more runs fetch the rest once tile_runs is known)
    static_assert(NTL == 512 && LCAP == 4096, "two unconditional comp loads per thread cover the first 1024 runs");
    constexpr int CPRE = 2;
    uint32_t c_pre[CPRE];
#pragma unroll
    for (int k = 0; k < CPRE; ++k) c_pre[k] = lj.comp_of_run[rb + tid + NTL * k];
    // (tile-local: a byte each -- packed at once, one register through the prologue)
    const uint32_t c_pack = ((c_pre[0] - cb) & 0xffu) | (((c_pre[1] - cb) & 0xffu) << 8);
    // non-fused: the root of a component carries its label (k_emit_tiles).  k_emit_tiles numbers the blobs of volume 1 by their
    // rank in the whole table: their labels are -1 - rank, and the blobs of volume 0 come off here (one scalar load beside the others)
    int32_t lab_pre = 0;
    int32_t vol0 = 0;
    if (!FUSED) {
        lab_pre = tid < CCAP ? lj.label_of_comp[lj.parent[cb + tid]] : 0;
        vol0 = lj.n_vols > 1 ? (int32_t)lj.ctr->n_blobs_vol0 : 0;
    }
    // fused: what kpar[] holds for my component -- its root's first key and id
    kp_t kp = KP_UNUSED;
    // first key of volume 1 = the keys of a plane (from the kernel arguments: loading it from the volume descriptor was a
    // dependent trip in front of every other load of the prologue -- 2 us, stamps)
    const int64_t key_base1 = (FUSED && td.n_planes > 1) ? (int64_t)uc * ur * us : INT64_MAX;
    uint32_t n_in = 0;
    // (waves 4-7, thread j = tid - 256: the record of component j of the tile if it is a root, loaded beside the inbox)
    const uint32_t jd = cb + (uint32_t)(tid & 255);
    uint32_t rec_n = 0;
    FixSums rec_f = fix_zero();
    long long rec_c = 0, rec_r = 0, rec_s = 0;
    if (FUSED) {
        kp = kuf_load(lj.kpar, jd);
        n_in = lj.inbox_count[(size_t)bid * INBOX_STRIDE];
    }
    auto own_list = [&](int32_t lab) { return lab < 0 ? lab + vol0 : lab; };
    const bool unit = lj.tile_mode[(int64_t)bid] != 0;
    const uint32_t n_runs_raw = lj.tile_runs[bid];   // (beside the mode, not behind it)
    const uint32_t n_runs = unit ? 0u : n_runs_raw;
    const bool fast = !unit && n_runs <= (uint32_t)LCAP;   // block-uniform
    const int pwl = tid % CW, prowl = (tid / CW) & 63;
    const int pr = r0 + (prowl & 7), ps = s0 + (prowl >> 3);
    const bool pvalid = tid < NU && pr < ur && ps < us && w0 + pwl < row_words;
    const int64_t pw = ((int64_t)ps * ur + pr) * row_words + (w0 + pwl);
    uint64_t m[2] = {0ull, 0ull};
    uint32_t base[2] = {0u, 0u};
    if (tid < 256) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool has = pvalid && p < td.n_planes;
            m[p] = has ? lj.mask[pw + p * plane_words] : 0ull;
            base[p] = has ? lj.run_base[pw + p * plane_words] : 0u;
        }
    }
    uint32_t total_blobs = 0, blobs_vol0 = 0;
    bool my_root = false;
    auto write_mask_tables = [&]() {
        if (tid < 256) {
            s_mh[2 * tid] = make_uint2((uint32_t)m[0], (uint32_t)m[1]);
            s_mh[2 * tid + 1] = make_uint2((uint32_t)(m[0] >> 32), (uint32_t)(m[1] >> 32));
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint64_t starts = run_starts(m[p]);
                const uint32_t slo = (uint32_t)starts, shi = (uint32_t)(starts >> 32), before = base[p] - rb - 1u;
                s_sb[p][2 * tid] = make_uint2(slo, before);
                s_sb[p][2 * tid + 1] = make_uint2(shi, before + (uint32_t)__popc(slo));
            }
        }
    };
    // label of the blob whose root is named by value v: signed by the root's plane, numbered by the rank of its first key
    auto label_of_value = [&](kp_t v) -> int32_t {
        const uint32_t key32 = (uint32_t)(v >> 32);
        const int plane = (int)(key32 >> 31);
        const unsigned long long key = (unsigned long long)(plane ? key_base1 : 0) + (key32 & 0x7fffffffu);
        const int32_t rank = (int32_t)rank_of_key(lj, s_pre, key);
        return lj.vol_sign[plane] > 0 ? 1 + rank : -1 - rank;
    };
    if (FUSED) {
        // (unconditionally: waiting for the inbox count here would hold back the table's loads; its barriers publish the zeros)
        static_assert(10 * CCAP == 5 * NTL, "five accumulator words per thread");
#pragma unroll
        for (int k = 0; k < 5; ++k) (&s_acc[0][0])[tid + NTL * k] = 0ull;
        if (tid < CCAP) s_accn[tid] = 0u;
        /*@L7*/
        total_blobs = rank_table_lds<NTL>(lj, s_pre, s_wave);   // (two barriers inside)
        /*@L1*/
        n_in = min(n_in, (uint32_t)INBOX_CAP);
        write_mask_tables();   // (the masks are back by now; their registers are free for the second trip)
        // ---- second trip: waves 0-3 fetch the tile's inbox (an entry per thread) and rank the roots of their components;
        //      waves 4-7 fetch the records of the tile's own roots.  Everything is issued before anything is consumed. ----
        static_assert(INBOX_CAP <= 256, "an inbox entry per thread of the lower half");
        const bool used = kp != KP_UNUSED;
        // an inbox entry is eleven 8-byte words (local | n, the seven FixSums fields, sum c / r / s): thread t of the lower half
        // takes words 0-4 of entry t, thread t + 256 words 5-10 (and word 0 again, for the slot).  The two halves of the
        // workgroup run DIFFERENT code from here to the barrier (whole waves: no divergence), and what one half holds is
        // defined inside its own region -- a value loaded in front of the other half's code would be live across it for the
        // register allocator (22 registers of entry + 24 of record beside the rank loads spilled to scratch).
        static_assert(sizeof(InboxEntry) == 88, "eleven words");
        const bool have = (uint32_t)(tid & 255) < n_in;
        const unsigned long long *ew = reinterpret_cast<const unsigned long long *>(lj.inbox + (size_t)bid * INBOX_CAP + (tid & 255));
        my_root = tid >= 256 && used && kp_id(kp) == jd;   // a root of this tile: its blob's table row is mine
        if (tid < 256) {
            unsigned long long e_head = 0ull, e_w[4] = {0ull, 0ull, 0ull, 0ull};
            if (have) {
                e_head = ew[0];
#pragma unroll
                for (int k = 0; k < 4; ++k) e_w[k] = ew[1 + k];
            }
            // thread 255 also ranks the first key of volume 1 (= the blobs of volume 0): in the slot of its own component if
            // that id is unused, as it nearly always is -- a second pass otherwise
            const bool vol0_lane = tid == 255 && key_base1 != INT64_MAX;
            int32_t raw = 0;
            if (used || vol0_lane) {
                const uint32_t key32 = (uint32_t)(kp >> 32);
                const int plane = used ? (int)(key32 >> 31) : 1;
                const unsigned long long key = (unsigned long long)(plane ? key_base1 : 0) + (used ? (key32 & 0x7fffffffu) : 0u);
                const int32_t rank = (int32_t)rank_of_key(lj, s_pre, key);
                if (used) raw = lj.vol_sign[plane] > 0 ? 1 + rank : -1 - rank;
                else s_vol0 = (uint32_t)rank;
            }
            if (vol0_lane && used) s_vol0 = rank_of_key(lj, s_pre, (unsigned long long)key_base1);   // (a tile with all 256 ids in use)
            if (tid == 255 && key_base1 == INT64_MAX) s_vol0 = total_blobs;
            s_lab[tid] = raw;
            if (have) {   // (the accumulators were cleared in front of the table's barriers)
                const uint32_t l = (uint32_t)e_head;
                atomicAdd(&s_accn[l], (uint32_t)(e_head >> 32));
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&s_acc[k][l], e_w[k]);
            }
        } else {
            unsigned long long e_head = 0ull, e_w[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
            if (have) {
                e_head = ew[0];
#pragma unroll
                for (int k = 0; k < 6; ++k) e_w[k] = ew[5 + k];
            }
            if (my_root) {
                rec_n = lj.r_n[jd];
                rec_f = fix_load(lj, jd);
                rec_c = lj.r_c[jd]; rec_r = lj.r_r[jd]; rec_s = lj.r_s[jd];
            }
            if (have) {
                const uint32_t l = (uint32_t)e_head;
#pragma unroll
                for (int k = 0; k < 6; ++k) atomicAdd(&s_acc[4 + k][l], e_w[k]);
            }
        }
        /*@L2*/
        __syncthreads();
        /*@L3*/
        blobs_vol0 = s_vol0;                                           // (one volume: every blob, as k_emit_tiles publishes it)
        vol0 = key_base1 != INT64_MAX ? (int32_t)blobs_vol0 : 0;       // what comes off the labels of volume 1
    }
    {
        if (!FUSED) write_mask_tables();
        if (tid < CCAP) s_lab[tid] = own_list(FUSED ? s_lab[tid] : lab_pre);
        // (fused: every rank of this tile has been taken -- the bytes take the place of the prefix table.  A unit tile takes
        //  ranks in its rows: it keeps the table, and does not read the bytes)
        if (!FUSED || fast) {
#pragma unroll
            for (int k = 0; k < CPRE; ++k) s_comp8[tid + NTL * k] = (uint8_t)(c_pack >> (8 * k));
        }
        if (fast)
            for (uint32_t i = tid + CPRE * NTL; i < n_runs; i += NTL) s_comp8[i] = (uint8_t)(lj.comp_of_run[rb + i] - cb);
    }
    __syncthreads();
    /*@L4*/
    if (FUSED) {
        // waves 4-7: the blob table rows of the tile's own roots -- record + inbox sums, rank from the label -- while waves 0-3
        // are already storing label rows
        if (my_root) {
            const int l = tid - 256;
            if (n_in != 0u) {
                rec_n += s_accn[l];
                rec_f.rho += (long long)s_acc[0][l];
                rec_f.c_lo += (long long)s_acc[1][l]; rec_f.c_hi += (long long)s_acc[2][l];
                rec_f.r_lo += (long long)s_acc[3][l]; rec_f.r_hi += (long long)s_acc[4][l];
                rec_f.s_lo += (long long)s_acc[5][l]; rec_f.s_hi += (long long)s_acc[6][l];
                rec_c += (long long)s_acc[7][l]; rec_r += (long long)s_acc[8][l]; rec_s += (long long)s_acc[9][l];
            }
            const uint32_t key32 = (uint32_t)(kp >> 32);
            const unsigned long long key = (unsigned long long)((key32 >> 31) ? key_base1 : 0) + (key32 & 0x7fffffffu);
            const int32_t lab = s_lab[l];   // (final: a blob of volume 1 carries -1 - (rank - vol0))
            emit_row_ranked(lj, *gp, true, key_base1, jd, rec_n, key, rec_f, rec_c, rec_r, rec_s, (uint32_t)(lab > 0 ? lab - 1 : vol0 - 1 - lab));
        }
        if (bid == 0 && tid == 0) {   // the table's totals, for the host
            lj.ctr->n_blobs = total_blobs;
            lj.ctr->n_blobs_vol0 = blobs_vol0;
            if (total_blobs > lj.blob_cap) atomicOr(&lj.ctr->overflow, 2u);
        }
        /*@L5*/
    }
    // wave wv writes RPW rows of the tile; a lane owns 4 consecutive voxels of a 256-voxel row.  A voxel carries one sign, and
    // nearly every 4-voxel group one sign or none: the lane works on the plane that has voxels in its group (a second pass,
    // taken only by a wave that has a group with both signs, adds the other), and a group of four holds at most two runs --
    // two table look-ups, not one per voxel.  The kernel is bound by the instructions it issues (8 waves a SIMD, ~0.7 us a
    // row), not by its 64 MiB of stores: storing only the groups that hold a label into a volume zeroed under k_tile_label
    // changed nothing (r03), so the row loop is written for instruction count -- tables per half word (above), the in-LDS and
    // inside-the-grid cases compiled apart.
    const int hsel = lane >> 3, sh = (lane & 7) * 4;   // which half word of the row; where my 4 bits sit in it
    const uint32_t lowm = (1u << sh) - 1u;
    const int c = w0 * 64 + lane * 4;
    const bool inside = (r0 + TILE_R <= ur) && (s0 + TILE_S <= us) && ((w0 + CW) * 64 <= uc) && (uc & 3) == 0;   // block-uniform
    // the label of a run by its global id (unit tiles): through its component to the root; fused, the root's key is ranked here
    const bool ids_ran_out = unit && lj.ctr->overflow != 0u;   // (block-uniform; read by unit tiles only: rare)
    auto label_of_run = [&](uint32_t run) -> int32_t {
        if (ids_ran_out) return 0;                     // (void job: stay inside the arena, the host runs it again)
        const uint32_t comp = lj.comp_of_run[run];
        if (FUSED) return own_list(label_of_value(kuf_load(lj.kpar, comp)));
        return own_list(lj.label_of_comp[lj.parent[comp]]);
    };
    auto rows = [&](auto fast_tag, auto inside_tag) {
        constexpr bool FAST = decltype(fast_tag)::value, INSIDE = decltype(inside_tag)::value;
        auto group = [&](int p, unsigned nib, int hidx, int32_t (&o)[4]) {
            const uint2 sb = s_sb[p][hidx];
            const unsigned snib = (sb.x >> sh) & 0xfu;
            const unsigned upto_first = (2u << __builtin_ctz(nib)) - 1u;     // the group's bits up to its first voxel
            const uint32_t run = sb.y + (uint32_t)__popc(sb.x & lowm) + (uint32_t)__popc(snib & upto_first);   // (tile-local; a voxel that continues a run: the last start before it)
            const unsigned s2 = snib & ~upto_first;                          // a second run starts inside the group
            const unsigned from2 = 0u - (s2 & (0u - s2));                    // all bits from that start on (0 without one)
            const unsigned abits = nib & ~from2, bbits = nib & from2;
            int32_t la, lb;
            if (FAST) { la = s_lab[s_comp8[run]]; lb = s_lab[s_comp8[run + 1u]]; }   // (lb: read, used only behind s2 -- the table has LCAP + 4 bytes)
            else {
                la = label_of_run(run + rb);
                lb = s2 ? label_of_run(run + rb + 1u) : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] |= (__builtin_amdgcn_sbfe((int)abits, k, 1) & la) | (__builtin_amdgcn_sbfe((int)bbits, k, 1) & lb);
        };
        constexpr int ROW_UNROLL = FAST ? RPW : 1;
#pragma unroll ROW_UNROLL   // (the look-ups of a unit tile go through memory and, fused, through rank_of_key: rare, kept small)
        for (int rr = 0; rr < RPW; ++rr) {
            const int rowl = wvs * RPW + rr;
            const int r = r0 + (rowl & 7), s = s0 + (rowl >> 3);
            if (!INSIDE && (r >= ur || s >= us)) continue;   // wave-uniform
            if (hsel >= 2 * CW || (!INSIDE && c >= uc)) continue;
            const int hidx = rowl * (2 * CW) + hsel;
            const uint2 mh = s_mh[hidx];
            const unsigned nib0 = (mh.x >> sh) & 0xfu, nib1 = (mh.y >> sh) & 0xfu;
            int32_t out[4] = {0, 0, 0, 0};
            {
                const bool second = nib0 == 0u;
                const unsigned nib = second ? nib1 : nib0;
                if (nib) group(second ? 1 : 0, nib, hidx, out);
            }
            if (__ballot(nib0 != 0u && nib1 != 0u)) {   // wave-uniform, rare
                if (nib0 != 0u && nib1 != 0u) group(1, nib1, hidx, out);
            }
            int32_t *dst = labels + ((int64_t)s * ur + r) * uc + c;
            if (INSIDE || (c + 3 < uc && ((uc & 3) == 0))) {
                // streaming store: the labels are not read again by this job, and keeping them out of L2 measured 2.4 us faster
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i o4 = {out[0], out[1], out[2], out[3]};
                __builtin_nontemporal_store(o4, reinterpret_cast<v4i *>(dst));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c + q < uc) dst[q] = out[q];
            }
        }
    };
    if (fast) { if (inside) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
    else { if (inside) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
    /*@L6*/
    if (FUSED) {   // unit components (every run of a tile that overflowed LDS its own component): none on ordinary maps
        const bool any_unit = *lj.unit_flag == lj.epoch;   // block-uniform
        if (any_unit) {
            __syncthreads();   // (everybody is done with the run -> component bytes: the prefix table comes back)
            rank_table_lds<NTL>(lj, s_pre, s_wave);
            const uint32_t n_comp = n_components(lj), first = (uint32_t)lj.n_tiles * (uint32_t)TILE_COMPS;
            if (n_comp > first && lj.ctr->overflow == 0u) emit_ids(lj, *gp, s_pre, true, key_base1, first, n_comp, bid, (uint32_t)lj.n_tiles);
        }
    }
}

// Decode the signed volume for one list: -1 background / other sign, else 0-based blob index.
__global__ void k_labels_decode(const int32_t *__restrict__ signed_labels, int64_t n, int sign, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t v = signed_labels[i];
        out[i] = sign > 0 ? (v > 0 ? v - 1 : -1) : (v < 0 ? -v - 1 : -1);
    }
}

}  // namespace pdbeda
