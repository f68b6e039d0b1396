/*
 * mpc_tree.hip -- the reference's own candidate set: the limited-branching tree of controller.cpp:80-189.
 *
 * The reference walks the tree depth-first, once per deepening iteration, with an explicit stack.  Here the
 * tree is expanded level-synchronously.  The children of all nodes of a level are laid out in parent order and,
 * within a parent, throttle-outer / rudder-inner (controller.cpp:174-182), so a node's index within its level
 * IS the ordinal at which the reference's DFS of that depth visits it.  Iteration `it` of the reference
 * (controller.cpp:80) selects, among the leaves at depth it+1, the first one with the strictly smallest
 * LAST-step score (:158-166): that is the (score, index) argmin of level it+1 here, so one expansion yields
 * every deepening iteration's result.  The deadline of :101-103 is polled between levels; a level that does
 * not complete before end_time is discarded exactly as the reference discards an unfinished iteration.
 *
 * Map of this file:
 *   control_pick / control_count     the control sets of :121-146, per child, in registers
 *   scan_*_kernel                    exclusive scan of the child counts -> offsets (= DFS ordinals) and bpar[] (a tile's first parent)
 *   expand_kernel / expand_fast_kernel<STORE>   one lane per child: parent from an LDS window of offsets, simulate + score
 *                                    (the reference's arithmetic / the streamlined model), SoA store of a stored level, argmin
 *   tree_finish_body                 fast expansion: the near-tie leaves once more along their paths with the strict model
 *   tree_post_kernel                 levels 1-4 without host round trips (reduction + the next level's count and scan)
 *   level_reduce / leaf_record / trace kernels   strict expansion: a level's winner, its record, the winner's trajectory
 *   node_margin, tile_select_kernel, tree_bound_*   the FINAL level as a search: reach bounds, the short list of tiles, the bound
 *   tree_greedy_seed_kernel, ghost_select_kernel    the level above it: ghosts
 *   solve_tree_once                  one search (or one shard of one); mpc_hip_solve_tree_deep: further levels in pieces
 * DESIGN.md sections 6 and 7 tell the story and the measurements.
 */
#include <math.h>
#include <string.h>

#include <algorithm>

#include "mpc_internal.h"

using namespace mpc;

/* ---- control sets, controller.cpp:121-146 --------------------------------------------------------------- */
/* the j-th control pair of that set (throttle outer, rudder inner, controller.cpp:174-182) without building the
 * arrays: indexing cs.r[] / cs.t[] with run-time subscripts puts them in scratch memory, and the expansion then
 * spends more time on scratch traffic than on the model */
__device__ inline void control_pick(const DevProblem &P, double last_r, double last_t, uint32_t j, double &r, double &t)
{
    const double tol = P.tol;
    const double low_r = last_r - P.gr, high_r = last_r + P.gr;
    const bool r3 = P.minr + tol < low_r && fabs(P.midr - low_r) > tol;
    const bool r4 = fabs(P.midr - high_r) > tol && P.maxr > high_r + tol;
    const bool r5 = fabs(P.minr - last_r) > tol && fabs(P.midr - last_r) > tol && fabs(P.maxr - last_r) > tol;
    const double low_t = last_t - P.gt, high_t = last_t + P.gt;
    const bool t2 = P.mint + tol < low_t;
    const bool t3 = P.maxt > high_t + tol;
    const bool t4 = fabs(P.mint - last_t) > tol && fabs(P.maxt - last_t) > tol;
    const uint32_t nr = 3u + r3 + r4 + r5;
    /* j / nr without the ~20-instruction integer division: nr is 3..6 and j < 30, where floor(j * ceil(2^16 / nr) / 2^16)
     * is exact (the excess j * (ceil - 2^16 / nr) / 2^16 < 30 / 65536 never reaches the next multiple of 1 / nr) */
    const uint32_t inv = nr == 3u ? 21846u : (nr == 4u ? 16384u : (nr == 5u ? 13108u : 10923u));
    const uint32_t ti = (j * inv) >> 16, ri = j - ti * nr;
    r = ri == 0 ? P.minr : (ri == 1 ? P.midr : P.maxr);
    int k = (int)ri - 3;                                   /* k-th of the optional rudders that are present */
    if (r3) { if (k == 0) r = low_r; k--; }
    if (r4) { if (k == 0) r = high_r; k--; }
    if (r5) { if (k == 0) r = last_r; }
    t = ti == 0 ? P.mint : P.maxt;
    k = (int)ti - 2;
    if (t2) { if (k == 0) t = low_t; k--; }
    if (t3) { if (k == 0) t = high_t; k--; }
    if (t4) { if (k == 0) t = last_t; }
}

/* only the counts (same predicates) */
__device__ inline uint32_t control_count(const DevProblem &P, double last_r, double last_t)
{
    const double tol = P.tol;
    int nr = 3, nt = 2;
    const double low_r = last_r - P.gr, high_r = last_r + P.gr;
    nr += (P.minr + tol < low_r && fabs(P.midr - low_r) > tol);
    nr += (fabs(P.midr - high_r) > tol && P.maxr > high_r + tol);
    nr += (fabs(P.minr - last_r) > tol && fabs(P.midr - last_r) > tol && fabs(P.maxr - last_r) > tol);
    const double low_t = last_t - P.gt, high_t = last_t + P.gt;
    nt += (P.mint + tol < low_t);
    nt += (P.maxt > high_t + tol);
    nt += (fabs(P.mint - last_t) > tol && fabs(P.maxt - last_t) > tol);
    return (uint32_t)(nr * nt);
}

static LevelPtrs ptrs(const TreeLevelBuf &b)
{
    const size_t f = b.first;       /* a sliced level looks like a whole one to the kernels */
    return LevelPtrs{b.x + f, b.y + f, b.h + f, b.v + f, b.rpm + f, b.cmg + f, b.sog + f, b.last_r + f, b.last_t + f,
                     b.cum + f, b.last_score + f, b.parent + f, b.count + f, b.offset + f, b.margin + f};
}

/* Node records.  strict: x y h v rpm cmg sog.  fast (MPC_PREC_FAST): x, y through the water, h = sin(heading),
 * cmg = cos(heading), v, rpm, sog = the last substep's v*dt -- the FastVeh of mpc_model.h in the same arrays. */
/* meta (optional; the blind shallow levels, below): also the root's child count and offsets, so that level 1 can be
 * expanded without a scan launch: meta[l * TREE_META + 0] = nodes of level l */
#define TREE_META 8              /* uint64 per level: nodes, winner key, winner ordinal, cum bits, last bits, overflow, clock, libm tie */
__global__ void root_kernel(const DevProblem *__restrict__ probs, LevelPtrs L, int fast, uint32_t *__restrict__ bpar,
                            uint64_t *__restrict__ meta)
{
    if (threadIdx.x || blockIdx.x) return;
    const DevProblem &P = probs[0];
    L.x[0] = P.sx; L.y[0] = P.sy; L.h[0] = fast ? sin(P.sh) : P.sh; L.v[0] = P.sv; L.rpm[0] = P.srpm;
    L.cmg[0] = fast ? cos(P.sh) : 0; L.sog[0] = 0; L.last_r[0] = P.last_r; L.last_t[0] = P.last_t; L.cum[0] = 0; L.last_score[0] = 0;
    L.parent[0] = 0; L.margin[0] = 0.0f;
    if (meta) {
        const uint32_t c = control_count(P, P.last_r, P.last_t);
        L.count[0] = c; L.offset[0] = 0; L.offset[1] = c; bpar[0] = 0;
        meta[0] = 1; meta[TREE_META] = c;
    }
}

__global__ __launch_bounds__(256) void count_kernel(const DevProblem *__restrict__ probs, LevelPtrs L, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) L.count[i] = control_count(probs[0], L.last_r[i], L.last_t[i]);
}

/* ---- exclusive scan of uint32 counts: per-block sums, scan of the sums, per-block rescan ----------------- */
#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *sh, uint32_t &total)
{
    /* 256 threads = four wavefronts: a shuffle scan inside each (six steps, no LDS), the four wave totals through LDS --
     * two barriers instead of the sixteen of a Hillis-Steele scan in LDS */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(x, d, 64);
        if (lane >= d) x += up;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    const uint32_t t0 = sh[0], t1 = sh[1], t2 = sh[2], t3 = sh[3];
    total = t0 + t1 + t2 + t3;
    const uint32_t before = (wave > 0 ? t0 : 0) + (wave > 1 ? t1 : 0) + (wave > 2 ? t2 : 0);
    __syncthreads();                                  /* sh may be rewritten by the caller's next scan */
    return before + x - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_sums_kernel(const uint32_t *__restrict__ in, uint32_t n,
                                                               uint32_t *__restrict__ sums)
{
    __shared__ uint32_t sh[SCAN_BLOCK];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    if (base + SCAN_ITEMS <= n && ((uintptr_t)(in + base) & 15u) == 0) {  /* eight counts as two 16-byte loads (a sliced level may start anywhere) */
        const uint4 lo = reinterpret_cast<const uint4 *>(in + base)[0], hi = reinterpret_cast<const uint4 *>(in + base)[1];
        s = lo.x + lo.y + lo.z + lo.w + hi.x + hi.y + hi.z + hi.w;
    } else {
        for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) s += in[base + j];
    }
    uint32_t total;
    block_exclusive_scan(s, sh, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

/* one block: exclusive scan of the tile sums in place; sums[nt] = grand total (as two words: lo, overflow flag).
 * SCAN_ITEMS consecutive sums per thread and trip (a 1e8-node level has 5e4 tile sums: 25 trips, not 200). */
__global__ __launch_bounds__(SCAN_BLOCK) void scan_top_kernel(uint32_t *__restrict__ sums, uint32_t nt)
{
    __shared__ uint32_t sh[SCAN_BLOCK];
    unsigned long long carry = 0;                          /* the same in every thread */
    for (uint32_t base = 0; base < nt; base += SCAN_TILE) {
        const uint32_t i0 = base + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) { v[j] = i0 + j < nt ? sums[i0 + j] : 0; s += v[j]; }
        uint32_t total;
        unsigned long long run = carry + block_exclusive_scan(s, sh, total);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            if (i0 + j < nt) sums[i0 + j] = (uint32_t)run;
            run += v[j];
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        sums[nt] = (uint32_t)(carry & 0xFFFFFFFFull);
        sums[nt + 1] = (uint32_t)(carry >> 32);          /* non-zero: the level does not fit 32-bit indices */
    }
}

/* also records, for every 256-child boundary q*256 of the NEXT level, the parent whose child range contains it
 * (bpar[q]): a node has fewer than 256 children, so each node covers at most one boundary and the expansion
 * workgroups need no search for their first parent */
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const uint32_t *__restrict__ in, uint32_t n,
                                                                const uint32_t *__restrict__ sums,
                                                                uint32_t *__restrict__ out, uint32_t *__restrict__ bpar)
{
    __shared__ uint32_t sh[SCAN_BLOCK];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
    const bool whole = base + SCAN_ITEMS <= n && (((uintptr_t)(in + base) | (uintptr_t)(out + base)) & 15u) == 0;   /* (a sliced level may start anywhere) */
    if (whole) {                                                       /* two 16-byte loads, two 16-byte stores */
        const uint4 lo = reinterpret_cast<const uint4 *>(in + base)[0], hi = reinterpret_cast<const uint4 *>(in + base)[1];
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) s += v[j];
    } else {
        for (int j = 0; j < SCAN_ITEMS; j++) { v[j] = (base + j < n) ? in[base + j] : 0; s += v[j]; }
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(s, sh, total) + sums[blockIdx.x];
    uint32_t o[SCAN_ITEMS];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        o[j] = run;
        if (base + j < n) {
            const uint64_t q = ((uint64_t)run + 255) >> 8;
            if ((q << 8) < (uint64_t)run + v[j]) bpar[q] = base + j;
        }
        run += v[j];
    }
    if (whole) {
        reinterpret_cast<uint4 *>(out + base)[0] = make_uint4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<uint4 *>(out + base)[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
        for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) out[base + j] = o[j];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) out[n] = sums[gridDim.x];
}

/* ---- expansion: one lane per child ------------------------------------------------------------------------ */
struct Child { Veh s; double r, t, score, cum; uint32_t parent; };

__device__ inline void tree_bound_update(uint64_t *bound, uint64_t k, const DevProblem &P, double rel);
__device__ inline double tree_bound_read(const uint64_t *bound);
__device__ inline uint64_t tree_bound_key(const uint64_t *bound);
__device__ inline void tree_bound_offer(uint64_t *bound, uint64_t seen, uint64_t kc, const DevProblem &P, double rel);

/* What a stored node keeps for the expansion of its children, should that be the search's final level: a lower bound on
 * the distance between the NEXT level's checkpoint and wherever a child of the node can end.  (gx, gy) is that checkpoint
 * in the frame the node's position is held in -- moved back by the current's drift over the level where the frame is the
 * ground's; (ux, uy) = (sin, cos) of the node's heading, any length.
 *   Through the water a child ends at p + sum v' dt (sin h, cos h) with 0 <= v', sum v' dt <= s = v0 T + a T (T + dt) / 2 and
 *   |h - h0| <= D = the most the rudder can turn the vessel in a level (DevProblem.reach_sd / _cd, mpc_pack_problem): a positive
 *   combination of unit vectors within D of the heading, so the end point lies in the circular sector of radius s and
 *   half-angle D about the heading (D < pi/2; a disc otherwise).  The margin is the checkpoint's distance to that sector.
 * Rounded down to a float; 0 = no statement: a NaN state, a node moving astern, a hull without a reach bound, a
 * checkpoint within reach. */
/* K = 2: the same over TWO levels -- (gx, gy) the checkpoint after the next one -- for the level above a final level
 * (ghost_select_kernel): speed bound v0 + 2 a T, path v0 2T + a 2T (2T + dt) / 2, turn 2 D. */
template <int K = 1>
__device__ inline float node_margin(const DevProblem &P, double gx, double gy, double px, double py, double v0, double rpm0,
                                    double ux, double uy)
{
    const double V = fma((double)K, P.reach_dv, v0);
    const bool bounded = v0 >= 0.0 && V <= P.reach_vcap && fabs(rpm0) <= P.reach_rpm;
    const double ex = gx - px, ey = gy - py;
    const double d = sqrt(fma(ex, ex, ey * ey));
    const double s = fma(v0, (double)K * P.reach_t, K == 1 ? P.reach_acc : P.reach_acc2) + 1.0e-6;
    /* the checkpoint in the heading's frame: along = d cos(angle to the heading), side = d |sin| of it */
    const double iu = rsqrt(fma(ux, ux, uy * uy));
    const double along = fma(ex, ux, ey * uy) * iu, side = fabs(fma(ex, uy, -(ey * ux))) * iu;
    const double c1 = P.reach_cd, s1 = P.reach_sd;
    const double cD = K == 1 ? c1 : fma(c1, c1, -(s1 * s1)), sD = K == 1 ? s1 : 2.0 * s1 * c1;
    double m = d - s;                                                    /* the disc; inside the sector's angle: its arc */
    if (c1 >= 0.0 && cD >= 0.0 && V <= P.reach_sqrtps && along < cD * d) {   /* outside the angle: the sector's nearer edge decides */
        const double a = fma(along, cD, side * sD), p = fmax(fma(side, cD, -(along * sD)), 0.0);   /* along / across that edge */
        m = a <= 0.0 ? d : (a <= s ? p : sqrt(fma(a - s, a - s, p * p)));
    }
    m = m * (1.0 - 1.0e-9) - 1.0e-6;
    return bounded && m > 0.0 ? __double2float_rd(m) : 0.0f;             /* (NaN anywhere: 0) */
}

__device__ inline float strict_node_margin(const DevProblem &P, const double *cp_next, const Veh &s)
{
    double sh, ch;
#if defined(__HIP_DEVICE_COMPILE__)
    heading_sincos(s.h, sh, ch);                                        /* (headings are in [0, 2 pi): estimate()'s fmod) */
#else
    sincos(s.h, &sh, &ch);
#endif
    return node_margin(P, cp_next[0] - P.reach_cx, cp_next[1] - P.reach_cy, s.x, s.y, s.v, s.rpm, sh, ch);
}

#define MARGIN_GHOST __builtin_inff()

/* the final level's tests: none of these nodes' children can end within dmax (NaN: no bound yet) of the checkpoint */
__device__ inline bool margin_beyond(float margin, double dmax) { return margin > 0.0f && ((double)margin >= dmax || margin == MARGIN_GHOST); }

/* A GHOST is a node of the level above a final level that was given its place in the level (its child count, hence its
 * children's ordinals) but never simulated: neither it nor any child of it can win (ghost_select_kernel).  Its margin says
 * so -- +inf, beyond any bound, even none -- and nothing else of its record is ever written or may be read: whoever makes
 * children from parents reports +inf for a ghost's. */
__device__ inline bool is_ghost(const LevelPtrs &A, uint32_t par) { return A.margin[par] == MARGIN_GHOST; }

/* one past a tile's last child, without wrapping for the last tile of a level of nearly 2^32 children */
__device__ inline uint32_t tile_end_of(uint32_t tile) { return tile + 1 < (1u << 24) ? (tile + 1) * 256 : 0xFFFFFFFFu; }

/* wave 0 of a workgroup, all 64 lanes: the smallest margin among the parents of one tile (lane i holds parent par0 + i's
 * offset; parents whose children begin at or after tile_end, and lanes past the level, do not belong to the tile) */
__device__ inline float tile_margin(const LevelPtrs &A, uint32_t par, uint32_t n_parent, uint32_t off, uint32_t tile_end)
{
    float m = (par < n_parent && off < tile_end) ? A.margin[par] : __builtin_inff();
    for (int d = 32; d >= 1; d >>= 1) m = fminf(m, __shfl_xor(m, d, 64));
    return m;
}

/* child c of the level whose parents are A: j-th control pair (throttle outer, rudder inner) of parent `par`.
 * PRUNE / dmax: as make_child_fast below -- a wave all of whose leaves are too far from the checkpoint to be the level's
 * winner (or its runner-up within the libm_tie window) skips compareStates and reports +inf. */
template <bool PRUNE = false>
__device__ inline Child make_child(const DevProblem &P, const double *cpL, const double *subL, const LevelPtrs &A,
                                   uint32_t par, uint32_t j, int level, double dmax = 0.0)
{
    double r, t;
    control_pick(P, A.last_r[par], A.last_t[par], j, r, t);
    Child ch;
    ch.s = Veh{A.x[par], A.y[par], A.h[par], A.v[par], A.rpm[par], 0.0, 0.0};
    const double *cp = cpL + level * CP_STRIDE;
    if (PRUNE && __builtin_amdgcn_ballot_w64(!margin_beyond(A.margin[par], dmax)) == 0) {   /* every parent of this wave's children */
        ch.score = ch.cum = __longlong_as_double((long long)KEY_INF);                 /* not simulated at all */
        ch.r = r; ch.t = t; ch.parent = par;
        return ch;
    }
    const Model M = P.model;                                                          /* wave-uniform: scalar registers */
    strict_level(ch.s, r, t, subL, P.nsub, P.cx, P.cy, M);                              /* controller.cpp:149-150 */
    const Weights w{P.wd, P.wh, P.ws};
    bool skip = false;
    if (PRUNE) {
        const double dx = cp[0] - ch.s.x, dy = cp[1] - ch.s.y;
        const bool far = dx * dx + dy * dy >= dmax * dmax;
        skip = __builtin_amdgcn_ballot_w64(!far) == 0;
    }
    ch.score = skip || is_ghost(A, par) ? __longlong_as_double((long long)KEY_INF) : strict_score(cp, ch.s, w);   /* :155 */
    ch.cum = A.cum[par] + ch.score;                                                   /* :156 */
    ch.r = r; ch.t = t; ch.parent = par;                                              /* :152-153 (unclamped) */
    return ch;
}

__device__ inline void store_child(const LevelPtrs &B, uint32_t at, const Child &ch)
{
    B.x[at] = ch.s.x; B.y[at] = ch.s.y; B.h[at] = ch.s.h; B.v[at] = ch.s.v; B.rpm[at] = ch.s.rpm;
    B.cmg[at] = ch.s.cmg; B.sog[at] = ch.s.sog;
    B.last_r[at] = ch.r; B.last_t[at] = ch.t; B.cum[at] = ch.cum; B.last_score[at] = ch.score; B.parent[at] = ch.parent;
}

/* parent of child c = the last node whose offset is <= c (global binary search) */
__device__ inline uint32_t find_parent(const uint32_t *offset, uint32_t n_parent, uint32_t c)
{
    uint32_t lo = 0, hi = n_parent;                     /* invariant: offset[lo] <= c < offset[hi] */
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (offset[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

/* the same when the parent of the first child of c's tile is known (bpar[c / 256], scan_apply_kernel): c's parent is one
 * of the next PARENT_WINDOW nodes -- six dependent loads instead of ~25 */
__device__ inline uint32_t find_parent_from(const uint32_t *offset, uint32_t par0, uint32_t n_parent, uint32_t c)
{
    uint32_t lo = par0, hi = min(par0 + 64u, n_parent);   /* offset[lo] <= c < offset[hi] (offset[n_parent] = the level's size) */
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (offset[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

/* A node has at least 3 x 2 children (controller.cpp:121-146), so the 256 children of a workgroup belong to at most
 * 44 consecutive parents: the first one comes from bpar[] (written by the scan), the workgroup stages a window of
 * offsets in LDS, and every lane finds its parent there -- no search in global memory (a binary search over a
 * 2e7-node level is ~25 dependent loads, ten times the workgroup's arithmetic). */
#define PARENT_WINDOW 64

/* STORE = false: the last level of a search -- its nodes will never be parents, so nothing but the workgroup
 * winners is written (100 B per child saved: 30 GB on a 3e8-leaf level) and no level buffer is needed. */
/* A workgroup expands tpb <= EXPAND_TILES consecutive tiles of 256 children: the problem is staged once per
 * workgroup (a 256-child workgroup spent more time in its prologue's memory latencies than on the model: 12.9 ->
 * 8.5 ms on a 3e8-leaf level), and the next tile's offset window is fetched while the current tile computes.  The
 * host picks tpb = 1 while a level has too few tiles to fill the chip. */
#define EXPAND_TILES 8

__device__ inline uint32_t window_parent(const uint32_t *win, uint32_t par0, uint32_t n_parent, uint32_t c, uint32_t &first)
{
    /* win[0] <= c < win[hi]; the window has PARENT_WINDOW = 64 entries, so SIX halvings always suffice: a fixed trip count
     * (no data-dependent loop: hipcc turns `while (hi - lo > 1)` into an exec-masked loop with a branch and a full LDS
     * wait per trip), every step one LDS read and two selects; steps past convergence re-read win[lo] and change nothing */
    uint32_t lo = 0, hi = min((uint32_t)PARENT_WINDOW - 1, n_parent - par0);
#pragma unroll
    for (int step = 0; step < 6; step++) {
        const uint32_t mid = (lo + hi) >> 1;              /* == lo once hi - lo <= 1 */
        const bool le = win[mid] <= c;
        lo = le ? mid : lo;
        hi = le ? hi : max(mid, lo + 1);                  /* never collapses the interval below one entry */
    }
    first = win[lo];
    return par0 + lo;
}

/* meta_parent != nullptr (the blind shallow levels): n_parent / n_child are upper bounds the grid was sized for; the
 * real counts are read from the device (meta_parent[0], meta_parent[TREE_META]) and surplus workgroups leave. */
template <bool STORE>
__global__ __launch_bounds__(256) void expand_kernel(const DevProblem *__restrict__ probs, LevelPtrs A, uint32_t n_parent,
                                                     const uint32_t *__restrict__ bpar, LevelPtrs B, uint32_t n_child, int level,
                                                     uint32_t tpb, uint64_t *__restrict__ pkey, uint64_t *__restrict__ pord,
                                                     uint64_t *__restrict__ pkey2, const uint64_t *__restrict__ meta_parent,
                                                     uint64_t *bound, const uint32_t *__restrict__ tile_list, uint32_t n_list)
{
    __shared__ __attribute__((aligned(16))) double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    __shared__ uint64_t red[10];
    __shared__ uint32_t win[2][PARENT_WINDOW];
    __shared__ double s_dmax[2];
    __shared__ float s_margin[2];
    const DevProblem &P = probs[0];
    double *cpL = lds, *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    if (meta_parent) { n_parent = (uint32_t)meta_parent[0]; n_child = (uint32_t)meta_parent[TREE_META]; }
    /* tile_list: the workgroup's tiles are entries of the list -- the final level's short list (tile_select_kernel), or the
     * tiles of the level above it that are not ghosts' (ghost_select_kernel) */
    const uint32_t *list = tile_list;
    const uint32_t n_tiles = list ? n_list : (n_child + 255) / 256;
    const uint32_t slot0 = blockIdx.x * tpb;
    if (slot0 >= n_tiles) {                                          /* uniform: a workgroup beyond the level */
        if (threadIdx.x == 0) { pkey[blockIdx.x] = ~0ull; pord[blockIdx.x] = ~0ull; pkey2[blockIdx.x] = ~0ull; }
        return;
    }
    const uint32_t tile0 = list ? list[slot0] : slot0;
    uint32_t par0 = bpar[tile0];                                     /* the parent of the tile's first child (scan_apply_kernel) */
    if (threadIdx.x < PARENT_WINDOW) {
        const uint32_t p = min(par0 + threadIdx.x, n_parent), off = A.offset[p];
        win[0][threadIdx.x] = off;
        if (!STORE) { const float m = tile_margin(A, p, n_parent, off, tile_end_of(tile0)); if (threadIdx.x == 0) s_margin[0] = m; }
    }
    if (!STORE && threadIdx.x == 0) s_dmax[0] = tree_bound_read(bound);
    stage_problem(P, cpL, subL);                                     /* ends with a barrier */
    uint64_t k = ~0ull, o = ~0ull, second = ~0ull;                   /* this lane's best child and the key of its next best */
    uint32_t tile = tile0;
    for (uint32_t i = 0; i < tpb && slot0 + i < n_tiles; i++) {
        const bool more = i + 1 < tpb && slot0 + i + 1 < n_tiles;
        uint32_t tile_next = 0, par_next = 0, p_next = 0, off_next = 0;
        if (more) {                                                   /* in flight during this tile's arithmetic */
            tile_next = list ? list[slot0 + i + 1] : tile + 1;
            par_next = bpar[tile_next];
            p_next = min(par_next + threadIdx.x, n_parent);
            if (threadIdx.x < PARENT_WINDOW) off_next = A.offset[p_next];
        }
        const uint32_t c = tile * 256 + threadIdx.x;
        /* the unstored (final) level: the bound as thread 0 read it before the last barrier -- one value for the whole
         * workgroup, so that leaving a tile out is a uniform decision: no parent of the tile has a child within reach */
        const double dmax = STORE ? 0.0 : s_dmax[i & 1];
        const bool skip_tile = !STORE && margin_beyond(s_margin[i & 1], dmax);
        if (!skip_tile && c < n_child) {
            uint32_t first;
            const uint32_t par = window_parent(win[i & 1], par0, n_parent, c, first);
            const Child ch = make_child<!STORE>(P, cpL, subL, A, par, c - first, level, dmax);
            if (STORE) {
                store_child(B, c, ch);
                B.count[c] = control_count(P, ch.r, ch.t);            /* its children-to-be, for the next level's scan */
                const double *cn = cpL + (level + 1) * CP_STRIDE;     /* (rows past the problem's depth are never read back) */
                B.margin[c] = level + 1 < P.depth ? strict_node_margin(P, cn, ch.s) : 0.0f;
            }
            const uint64_t kc = key_of(ch.score);
            if (kc < k) { second = k; k = kc; o = c; }                /* (a lane's children come in ascending order: ties keep the first) */
            else if (kc < second) second = kc;
        }
        if (more) {
            if (threadIdx.x < PARENT_WINDOW) {
                win[(i + 1) & 1][threadIdx.x] = off_next;
                if (!STORE) { const float m = tile_margin(A, p_next, n_parent, off_next, tile_end_of(tile_next)); if (threadIdx.x == 0) s_margin[(i + 1) & 1] = m; }
            }
            if (!STORE && threadIdx.x == 0) s_dmax[(i + 1) & 1] = tree_bound_read(bound);
            par0 = par_next; tile = tile_next;
            __syncthreads();
        }
    }
    const uint64_t my_k = k, my_o = o;
    block_argmin(k, o, red);
    if (threadIdx.x == 0) {
        pkey[blockIdx.x] = k; pord[blockIdx.x] = o; red[8] = o;
        /* strict expansion: the window that must stay unpruned is libm_tie's (1e-12); 1e-9 leaves six orders of margin.
         * (Once per workgroup: offering after every tile, as the fast expansion does, made this kernel 27 % slower.) */
        if (!STORE) tree_bound_update(bound, k, P, 1.0e-9);
    }
    __syncthreads();
    /* the workgroup's runner-up key (mpc_tree_level.libm_tie): every lane offers its best, the winner's lane its next best */
    uint64_t ru = (my_o == red[8]) ? second : my_k, zero = 0;
    block_argmin(ru, zero, red);
    if (threadIdx.x == 0) pkey2[blockIdx.x] = ru;
}

/* the level's runner-up key from the workgroups' (key, ordinal, runner-up key): a workgroup other than the winner's offers
 * its best, the winner's its runner-up; every thread of the block must call, valid in thread 0 */
__device__ inline uint64_t level_runner_up(const uint64_t *__restrict__ pkey, const uint64_t *__restrict__ pord,
                                           const uint64_t *__restrict__ pkey2, uint32_t nblk, uint64_t winner, uint64_t *red)
{
    uint64_t ru = ~0ull, zero = 0;
    for (uint32_t j0 = threadIdx.x; j0 < nblk; j0 += 8 * blockDim.x) {      /* eight loads in flight per lane, as in tree_finish_body */
        uint64_t kk[8], oo[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = j0 + u * blockDim.x;
            kk[u] = j < nblk ? pkey[j] : ~0ull; oo[u] = j < nblk ? pord[j] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = j0 + u * blockDim.x;
            if (j < nblk) ru = min(ru, oo[u] == winner ? pkey2[j] : kk[u]);
        }
    }
    block_argmin(ru, zero, red);
    return ru;
}

/* the winner of a level that was expanded without stores: evaluated once more (same code, same bits) into slot 0 of
 * the level's one-node buffer, where level_reduce_kernel and trace_kernel find it */
__global__ __launch_bounds__(64) void leaf_record_kernel(const DevProblem *__restrict__ probs, LevelPtrs A, uint32_t n_parent,
                                                         LevelPtrs B, int level, const uint64_t *__restrict__ winner)
{
    __shared__ __attribute__((aligned(16))) double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    const DevProblem &P = probs[0];
    stage_problem(P, lds, lds + MPC_MAX_DEPTH * CP_STRIDE);
    if (threadIdx.x != 0 || winner[1] == ~0ull) return;
    const uint32_t c = (uint32_t)winner[1];
    const uint32_t par = find_parent(A.offset, n_parent, c);
    store_child(B, 0, make_child(P, lds, lds + MPC_MAX_DEPTH * CP_STRIDE, A, par, c - A.offset[par], level));
}

/* ---- fast-model expansion (MPC_PREC_FAST) -------------------------------------------------------------------
 * The same tree (the control sets depend on controls only, never on states), expanded with the streamlined model
 * of mpc_model.h (27 instead of ~150 FP64 instructions per substep).  What makes the result the reference's:
 * every leaf whose fast last-step score is within MPC_REFINE_REL of the level's minimum is re-evaluated along its
 * whole path from the root with the strict model, the strict (score, ordinal) decides, and the returned trajectory
 * and scores are the strict path's -- bit for bit what the strict expansion returns for the same leaf. */
struct FastChild { FastVeh s; double r, t, score; uint32_t parent; };

/* dmax (PRUNE): a leaf whose distance to its checkpoint is at least dmax cannot be the level's winner nor lie
 * inside its refine window, whatever its heading and speed terms are (all three terms of controller.cpp:240 are >= 0, so
 * score >= w_d exp(distance)): when that holds for EVERY leaf of a wave, the wave skips the cost -- atan2, two square roots,
 * exp: ~100 of ~600 instructions per leaf -- and reports +inf.  The bound only ever comes from scores already computed
 * (tree_bound_update below), so which leaves are skipped depends on timing, but no result does: the level's minimum, its
 * first holder and its near-tie set are the same with any bound, including none. */
template <bool PRUNE = false>
__device__ inline FastChild make_child_fast(const DevProblem &P, const double *cpL, const double *subL, const LevelPtrs &A,
                                            uint32_t par, uint32_t j, int level, double dmax = 0.0)
{
    FastChild ch;
    control_pick(P, A.last_r[par], A.last_t[par], j, ch.r, ch.t);
    ch.s.x = A.x[par]; ch.s.y = A.y[par]; ch.s.sh = A.h[par]; ch.s.ch = A.cmg[par]; ch.s.v = A.v[par]; ch.s.rpm = A.rpm[par];
    ch.s.delta = 0;
    ch.parent = par;
    const double *cp = cpL + level * CP_STRIDE;
    if (PRUNE && __builtin_amdgcn_ballot_w64(!margin_beyond(A.margin[par], dmax)) == 0) {   /* every parent of this wave's children */
        ch.score = __longlong_as_double((long long)KEY_INF);                       /* not simulated at all */
        return ch;
    }
    const Model M = P.model;
    if (P.head_uniform) fast_level_head_uniform<true>(ch.s, ch.r, ch.t, subL, P.nsub, M, P.head_uniform == 2);      /* the reference's own 1-s level */
    else fast_level<SUB_STRIDE, true>(ch.s, ch.r, ch.t, subL, P.nsub, M);
    const Weights w{P.wd, P.wh, P.ws};
    bool skip = false;
    if (PRUNE) {
        const double ex = cp[6] - ch.s.x, ey = cp[7] - ch.s.y;
        const bool far = fma(ex, ex, ey * ey) >= dmax * dmax;          /* false for a NaN state and for "no bound yet" (NaN) */
        skip = __builtin_amdgcn_ballot_w64(!far) == 0;                 /* wave-uniform: every active lane is far */
    }
    ch.score = skip || is_ghost(A, par) ? __longlong_as_double((long long)KEY_INF) : fast_score<true>(cp, ch.s, w, subL + (P.nsub - 1) * SUB_STRIDE);
    return ch;
}

/* The pruning bound of an unstored (final) level: bound[0] = the smallest score key any finished workgroup has reported,
 * bound[1] = bits of dmax = log(limit / w_d) with limit = twice the refine window above that score (strict
 * expansion: the libm_tie window, which is far narrower) -- both only ever decrease (atomicMin on the bit patterns of
 * non-negative doubles).  Thread 0 of a workgroup, once, after its argmin. */
__device__ inline void tree_bound_update(uint64_t *bound, uint64_t k, const DevProblem &P, double rel)
{
    if (k >= KEY_INF) return;
    const uint64_t old = atomicMin((unsigned long long *)bound, (unsigned long long)k);
    if (k >= old) return;
    const double limit = mpc_refine_limit(__longlong_as_double((long long)k), rel, P.wd, P.wh, P.ws);
    const double dmax = log(limit / P.wd);                             /* w_d == 0: +inf, nothing is ever far */
    if (!(dmax >= 0.0)) return;                                        /* (a score below w_d cannot happen: exp >= 1) */
    atomicMin((unsigned long long *)(bound + 1), (unsigned long long)__double_as_longlong(dmax));
}

/* After a tile of the unstored level: a wave one of whose scores beats the best key it saw before the tile (seen) reports
 * its minimum -- rare once the first tiles of a level are through, so the level's bound is there after the first tile
 * of the first workgroups instead of after their last.  Every lane of the wave must call (kc = all ones for a lane
 * without a child). */
__device__ inline void tree_bound_offer(uint64_t *bound, uint64_t seen, uint64_t kc, const DevProblem &P, double rel)
{
    if (__builtin_amdgcn_ballot_w64(kc < seen) == 0) return;          /* wave-uniform */
    uint64_t m = kc;
    for (int d = 32; d >= 1; d >>= 1) m = min(m, (uint64_t)__shfl_xor((unsigned long long)m, d, 64));
    if ((threadIdx.x & 63) == 0) tree_bound_update(bound, m, P, rel);
}

__device__ inline uint64_t tree_bound_key(const uint64_t *bound)
{
    return __hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* the current bound, read past the caches (another workgroup's atomic wrote it) */
__device__ inline double tree_bound_read(const uint64_t *bound)
{
    return __longlong_as_double((long long)__hip_atomic_load(bound + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <bool STORE>
__global__ __launch_bounds__(256) void expand_fast_kernel(const DevProblem *__restrict__ probs, LevelPtrs A, uint32_t n_parent,
                                                          const uint32_t *__restrict__ bpar, LevelPtrs B, uint32_t n_child, int level,
                                                          double refine_rel, uint32_t tpb,
                                                          uint64_t *__restrict__ pkey, uint64_t *__restrict__ pord,
                                                          uint32_t *__restrict__ pcnt, const uint64_t *__restrict__ meta_parent,
                                                          uint64_t *bound, const uint32_t *__restrict__ tile_list, uint32_t n_list)
{
    __shared__ __attribute__((aligned(16))) double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    __shared__ uint64_t red[10];
    __shared__ uint32_t win[2][PARENT_WINDOW];
    __shared__ double s_dmax[2];
    __shared__ float s_margin[2];
    const DevProblem &P = probs[0];
    double *cpL = lds, *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    if (meta_parent) { n_parent = (uint32_t)meta_parent[0]; n_child = (uint32_t)meta_parent[TREE_META]; }
    /* tile_list: the workgroup's tiles are entries of the list -- the final level's short list (tile_select_kernel), or the
     * tiles of the level above it that are not ghosts' (ghost_select_kernel) */
    const uint32_t *list = tile_list;
    const uint32_t n_tiles = list ? n_list : (n_child + 255) / 256;
    const uint32_t slot0 = blockIdx.x * tpb;
    if (slot0 >= n_tiles) {
        if (threadIdx.x == 0) { pkey[blockIdx.x] = ~0ull; pord[blockIdx.x] = ~0ull; pcnt[blockIdx.x] = 0; }
        return;
    }
    const uint32_t tile0 = list ? list[slot0] : slot0;
    uint32_t par0 = bpar[tile0];
    if (threadIdx.x < PARENT_WINDOW) {
        const uint32_t p = min(par0 + threadIdx.x, n_parent), off = A.offset[p];
        win[0][threadIdx.x] = off;
        if (!STORE) { const float m = tile_margin(A, p, n_parent, off, tile_end_of(tile0)); if (threadIdx.x == 0) s_margin[0] = m; }
    }
    if (!STORE && threadIdx.x == 0) s_dmax[0] = tree_bound_read(bound);
    stage_problem(P, cpL, subL);
    uint64_t k = ~0ull, o = ~0ull;
    double keys[EXPAND_TILES];                                         /* this lane's scores, for the near-tie count */
    uint32_t tile = tile0;
#pragma unroll
    for (uint32_t i = 0; i < EXPAND_TILES; i++) {
        keys[i] = __longlong_as_double((long long)KEY_INF);
        if (i < tpb && slot0 + i < n_tiles) {                          /* uniform */
            const bool more = i + 1 < tpb && slot0 + i + 1 < n_tiles;
            uint32_t tile_next = 0, par_next = 0, p_next = 0, off_next = 0;
            if (more) {
                tile_next = list ? list[slot0 + i + 1] : tile + 1;
                par_next = bpar[tile_next];
                p_next = min(par_next + threadIdx.x, n_parent);
                if (threadIdx.x < PARENT_WINDOW) off_next = A.offset[p_next];
            }
            const uint32_t c = tile * 256 + threadIdx.x;
            /* the unstored (final) level: the bound as thread 0 read it before the last barrier -- one value for the whole
             * workgroup, so that leaving a tile out is a uniform decision: no parent of the tile has a child within reach */
            const double dmax = STORE ? 0.0 : s_dmax[i & 1];
            const bool skip_tile = !STORE && margin_beyond(s_margin[i & 1], dmax);
            if (!skip_tile) {
                const uint64_t seen = STORE ? 0 : tree_bound_key(bound);
                uint64_t kc = ~0ull;
                if (c < n_child) {
                    uint32_t first;
                    const uint32_t par = window_parent(win[i & 1], par0, n_parent, c, first);
                    const FastChild ch = make_child_fast<!STORE>(P, cpL, subL, A, par, c - first, level, dmax);
                    if (STORE) {
                        B.x[c] = ch.s.x; B.y[c] = ch.s.y; B.h[c] = ch.s.sh; B.cmg[c] = ch.s.ch; B.v[c] = ch.s.v; B.rpm[c] = ch.s.rpm;
                        /* (sog, cum, last_score stay unwritten: fast nodes are only ever parents -- what is returned comes
                         * from the strict model along the winner's path -- and a level is bound by these stores) */
                        B.last_r[c] = ch.r; B.last_t[c] = ch.t; B.parent[c] = ch.parent;
                        B.count[c] = control_count(P, ch.r, ch.t);        /* its children-to-be, for the next level's scan */
                        const double *cn = cpL + (level + 1) * CP_STRIDE;   /* (rows past the problem's depth are never read back) */
                        B.margin[c] = level + 1 < P.depth ? node_margin(P, cn[6], cn[7], ch.s.x, ch.s.y, ch.s.v, ch.s.rpm, ch.s.sh, ch.s.ch) : 0.0f;
                    }
                    kc = key_of(ch.score);
                    keys[i] = __longlong_as_double((long long)kc);
                    lex_min(k, o, kc, (uint64_t)c);
                }
                if (!STORE) tree_bound_offer(bound, seen, kc, P, 2.0 * refine_rel);
            }
            if (more) {
                if (threadIdx.x < PARENT_WINDOW) {
                    win[(i + 1) & 1][threadIdx.x] = off_next;
                    if (!STORE) { const float m = tile_margin(A, p_next, n_parent, off_next, tile_end_of(tile_next)); if (threadIdx.x == 0) s_margin[(i + 1) & 1] = m; }
                }
                if (!STORE && threadIdx.x == 0) s_dmax[(i + 1) & 1] = tree_bound_read(bound);
                par0 = par_next; tile = tile_next;
                __syncthreads();
            }
        }
    }
    block_argmin(k, o, red);
    /* how many of this workgroup's children lie within refine_rel of ITS minimum (a superset test for the level's) */
    uint32_t *cnt = (uint32_t *)(red + 9);
    if (threadIdx.x == 0) { red[8] = k; *cnt = 0; }
    __syncthreads();
    const double m = __longlong_as_double((long long)red[8]);
    const double limit = mpc_refine_limit(m, refine_rel, P.wd, P.wh, P.ws);
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t i = 0; i < EXPAND_TILES; i++) mine += keys[i] <= limit && keys[i] < __longlong_as_double((long long)KEY_INF);
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        pkey[blockIdx.x] = k; pord[blockIdx.x] = o; pcnt[blockIdx.x] = *cnt;
    }
}

/* controls along the path root -> child c of the level whose parents are levels[L] (n_parent of them) */
struct PathCtl { double r[MPC_MAX_DEPTH], t[MPC_MAX_DEPTH]; };

__device__ inline void path_controls(const DevProblem &P, const LevelPtrs *levels, int L, uint32_t n_parent, uint32_t c,
                                     PathCtl &pc, const uint32_t *bpar = nullptr)
{
    const LevelPtrs A = levels[L];
    uint32_t node = bpar ? find_parent_from(A.offset, bpar[c >> 8], n_parent, c) : find_parent(A.offset, n_parent, c);
    control_pick(P, A.last_r[node], A.last_t[node], c - A.offset[node], pc.r[L], pc.t[L]);
    for (int l = L; l >= 1; l--) {
        const LevelPtrs N = levels[l];
        pc.r[l - 1] = N.last_r[node]; pc.t[l - 1] = N.last_t[node];
        node = N.parent[node];
    }
}

/* the reference's arithmetic along one path: controller.cpp:149-156 for levels 0..L */
__device__ inline void strict_path(const DevProblem &P, const double *cpL, const double *subL, const PathCtl &pc, int L,
                                   double &last, double &cum, mpc_level_state *traj)
{
    Veh s{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0};
    const Weights w{P.wd, P.wh, P.ws};
    const Model M = P.model;
    last = 0; cum = 0;
    if (traj) traj[0] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
    for (int l = 0; l <= L; l++) {
        strict_level(s, pc.r[l], pc.t[l], subL, P.nsub, P.cx, P.cy, M);
        last = strict_score(cpL + l * CP_STRIDE, s, w);
        cum += last;
        if (traj) traj[l + 1] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, s.cmg, s.sog, pc.r[l], pc.t[l], cum};
    }
}

/* one workgroup per level: fast minimum -> near-tie list -> strict decision.  out[0..3] = strict key bits, leaf
 * ordinal, strict cumulative and last-step score bits; out[4] = 1 if more than LIST_CAP leaves tie (the caller then
 * repeats the search with the strict expansion). */
struct TreeFinishLds {
    double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    uint64_t s_list[LIST_CAP];
    uint64_t red[32];                  /* block_argmin: two words per wave, up to 16 waves */
    uint64_t s_win[2];
    uint32_t s_blocks[256];
    uint32_t s_n, s_nb;
};

__device__ inline void tree_finish_body(const DevProblem *__restrict__ probs, const LevelPtrs *__restrict__ levels,
                                        int L, uint32_t n_parent, uint32_t n_child, double refine_rel, uint32_t tpb,
                                        const uint64_t *__restrict__ pkey, const uint64_t *__restrict__ pord,
                                        const uint32_t *__restrict__ pcnt, uint32_t nblk, const uint32_t *__restrict__ bpar,
                                        uint64_t *__restrict__ out, TreeFinishLds &S,
                                        const uint32_t *__restrict__ tile_list = nullptr, uint32_t n_list = 0)
{
    double *lds = S.lds;
    uint64_t *s_list = S.s_list, *red = S.red, *s_win = S.s_win;
    uint32_t *s_blocks = S.s_blocks;
    uint32_t &s_n = S.s_n, &s_nb = S.s_nb;
    const DevProblem &P = probs[0];
    double *cpL = lds, *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    if (threadIdx.x == 0) { s_n = 0; s_nb = 0; }
    stage_problem(P, cpL, subL);
    /* (one workgroup walks all the expansion's workgroups: eight independent loads in flight per lane and trip -- a plain
     * loop is one memory latency per trip, 0.3 ms of them on a 1.5e5-workgroup level) */
    uint64_t k = ~0ull, o = ~0ull;
    for (uint32_t j0 = threadIdx.x; j0 < nblk; j0 += 8 * blockDim.x) {
        uint64_t kk[8], oo[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = j0 + u * blockDim.x;
            kk[u] = j < nblk ? pkey[j] : ~0ull; oo[u] = j < nblk ? pord[j] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) lex_min(k, o, kk[u], oo[u]);
    }
    block_argmin(k, o, red);
    if (threadIdx.x == 0) { s_win[0] = k; s_win[1] = o; }
    __syncthreads();
    k = s_win[0]; o = s_win[1];
    if (k >= KEY_INF) {                                      /* no finite score: nothing selected */
        if (threadIdx.x == 0) { out[0] = KEY_INF; out[1] = ~0ull; out[2] = KEY_INF; out[3] = KEY_INF; out[4] = 0; out[5] = 0; }
        return;
    }
    const double m = __longlong_as_double((long long)k);
    const double limit = mpc_refine_limit(m, refine_rel, P.wd, P.wh, P.ws);
    uint32_t overflow = 0;
    for (uint32_t j0 = threadIdx.x; j0 < nblk; j0 += 8 * blockDim.x) {
        uint64_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t j = j0 + u * blockDim.x; kk[u] = j < nblk ? pkey[j] : ~0ull; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = j0 + u * blockDim.x;
            if (!(__longlong_as_double((long long)kk[u]) <= limit)) continue;     /* (all ones -- a workgroup none of whose tiles was expanded, or no workgroup -- reads as NaN) */
            if (pcnt[j] <= 1) {
                const uint32_t at = atomicAdd(&s_n, 1u);
                if (at < LIST_CAP) s_list[at] = pord[j];
            } else {                                             /* several near-ties in one expansion workgroup */
                const uint32_t at = atomicAdd(&s_nb, 1u);
                if (at < 256) s_blocks[at] = j;
            }
        }
    }
    __syncthreads();
    if (s_nb > 256) overflow = 1;
    /* those workgroups' children once more, a lane per child and 256 lanes per tile, to see which of them are near */
    const LevelPtrs A = levels[L];
    for (uint32_t b = threadIdx.x >> 8; b < min(s_nb, 256u) * tpb; b += blockDim.x >> 8) {
        const uint32_t slot = s_blocks[b / tpb] * tpb + b % tpb;      /* as the expansion's workgroup numbered its tiles */
        const uint32_t tile = tile_list ? tile_list[min(slot, n_list - 1)] : slot;
        const uint32_t c = tile * 256 + (threadIdx.x & 255);
        if ((!tile_list || slot < n_list) && c < n_child) {
            const uint32_t par = find_parent_from(A.offset, bpar[tile], n_parent, c);
            const FastChild ch = make_child_fast(P, cpL, subL, A, par, c - A.offset[par], L);
            if (ch.score <= limit) {
                const uint32_t at = atomicAdd(&s_n, 1u);
                if (at < LIST_CAP) s_list[at] = c;
            }
        }
    }
    __syncthreads();
    uint32_t n = s_n;
    if (n > LIST_CAP) { overflow = 1; n = LIST_CAP; }
    /* the strict model along each listed leaf's path decides, lowest ordinal among equal scores */
    uint64_t k2 = ~0ull, o2 = ~0ull, second = ~0ull;
    double my_last = 0, my_cum = 0;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        PathCtl pc;
        path_controls(P, levels, L, n_parent, (uint32_t)s_list[e], pc, bpar);
        double last, cum;
        strict_path(P, cpL, subL, pc, L, last, cum, nullptr);
        const uint64_t ke = key_of(last);
        if (ke < k2 || (ke == k2 && s_list[e] < o2)) { second = k2; k2 = ke; o2 = s_list[e]; my_last = last; my_cum = cum; }
        else if (ke < second) second = ke;
    }
    const uint64_t mine_k = k2, mine_o = o2;
    block_argmin(k2, o2, red);
    if (threadIdx.x == 0) { s_win[0] = k2; s_win[1] = o2; }
    __syncthreads();
    const uint64_t wk = s_win[0], wo = s_win[1];
    /* the strict runner-up (mpc_tree_level.libm_tie): a leaf outside the near-tie list is > MPC_REFINE_REL away, never
     * within an ulp or two */
    uint64_t ru = (mine_o == wo) ? second : mine_k, zero = 0;
    block_argmin(ru, zero, red);
    if (threadIdx.x == 0) out[5] = keys_tie(wk, ru, P.wd, P.wh, P.ws);
    if (mine_k == wk && mine_o == wo && mine_o != ~0ull) {      /* exactly one lane holds the winner */
        out[0] = mine_k; out[1] = mine_o;
        out[2] = (uint64_t)__double_as_longlong(my_cum); out[3] = (uint64_t)__double_as_longlong(my_last);
        out[4] = overflow;
    }
}

#define TREE_FINISH_THREADS 512
__global__ __launch_bounds__(TREE_FINISH_THREADS) void tree_finish_kernel(const DevProblem *__restrict__ probs, const LevelPtrs *__restrict__ levels,
                                                          int L, uint32_t n_parent, uint32_t n_child, double refine_rel, uint32_t tpb,
                                                          const uint64_t *__restrict__ pkey, const uint64_t *__restrict__ pord,
                                                          const uint32_t *__restrict__ pcnt, uint32_t nblk,
                                                          const uint32_t *__restrict__ bpar, uint64_t *__restrict__ out,
                                                          const uint32_t *__restrict__ tile_list, uint32_t n_list)
{
    __shared__ __attribute__((aligned(16))) TreeFinishLds S;
    tree_finish_body(probs, levels, L, n_parent, n_child, refine_rel, tpb, pkey, pord, pcnt, nblk, bpar, out, S, tile_list, n_list);
}

/* ---- the blind shallow levels --------------------------------------------------------------------------------
 * While a level cannot have more than BLIND_MAX_NODES nodes whatever the problem (30^l: levels 1-4), the host does not
 * need its size to launch it: buffers and grids are sized for the bound, the kernels read the real counts from the
 * device (meta[]), surplus workgroups leave, and ONE read-back after the last blind level returns every level's
 * size and winner.  Per level two launches -- expand, and this one, which is the level's reduction (strict: + the
 * record of an unstored level's winner and its scores; fast: tree_finish_body) followed by the count + exclusive scan
 * of the new level for the next expansion -- instead of seven and two host round trips. */
#define BLIND_MAX_NODES (1u << 20)

template <bool FAST>
__global__ __launch_bounds__(256) void tree_post_kernel(const DevProblem *__restrict__ probs, const LevelPtrs *__restrict__ levels,
                                                        int L, int stored, double refine_rel, uint32_t tpb,
                                                        const uint64_t *__restrict__ pkey, const uint64_t *__restrict__ pord,
                                                        const uint32_t *__restrict__ pcnt, const uint64_t *__restrict__ pkey2,
                                                        uint64_t *__restrict__ out,
                                                        uint64_t *__restrict__ meta, int do_scan, uint32_t *__restrict__ bpar)
{
    __shared__ __attribute__((aligned(16))) TreeFinishLds S;
    __shared__ uint32_t sh[SCAN_BLOCK];
    const DevProblem &P = probs[0];
    const uint32_t n_parent = (uint32_t)meta[L * TREE_META], n_child = (uint32_t)meta[(L + 1) * TREE_META];
    const uint32_t n_tiles = (n_child + 255) / 256, nblk = (n_tiles + tpb - 1) / tpb;
    const LevelPtrs A = levels[L], B = levels[L + 1];
    if (FAST) {
        tree_finish_body(probs, levels, L, n_parent, n_child, refine_rel, tpb, pkey, pord, pcnt, nblk, bpar, out, S);
    } else {
        stage_problem(P, S.lds, S.lds + MPC_MAX_DEPTH * CP_STRIDE);
        uint64_t k = ~0ull, o = ~0ull;
        for (uint32_t j = threadIdx.x; j < nblk; j += blockDim.x) lex_min(k, o, pkey[j], pord[j]);
        block_argmin(k, o, S.red);
        if (threadIdx.x == 0) { S.s_win[0] = k; S.s_win[1] = o; }
        __syncthreads();
        const uint64_t ru = level_runner_up(pkey, pord, pkey2, nblk, S.s_win[1], S.red);
        if (threadIdx.x == 0) {
            out[5] = keys_tie(k, ru, P.wd, P.wh, P.ws);
            uint64_t cum = KEY_INF, last = KEY_INF;
            if (o != ~0ull) {
                uint32_t at = (uint32_t)o;
                if (!stored) {                                   /* leaf_record_kernel: the winner's record into slot 0 */
                    const uint32_t par = find_parent(A.offset, n_parent, (uint32_t)o);
                    store_child(B, 0, make_child(P, S.lds, S.lds + MPC_MAX_DEPTH * CP_STRIDE, A, par, (uint32_t)o - A.offset[par], L));
                    __threadfence();
                    at = 0;
                }
                cum = (uint64_t)__double_as_longlong(B.cum[at]); last = (uint64_t)__double_as_longlong(B.last_score[at]);
            }
            out[0] = k; out[1] = o; out[2] = cum; out[3] = last; out[4] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t *m = meta + (L + 1) * TREE_META;
        m[1] = out[0]; m[2] = out[1]; m[3] = out[2]; m[4] = out[3]; m[5] = FAST ? out[4] : 0;
        m[6] = (uint64_t)wall_clock64(); m[7] = out[5];
    }
    if (!do_scan) return;
    /* the new level as parents: children per node (controller.cpp:121-146) and their exclusive scan, one workgroup */
    const uint32_t n = n_child, ipt = (n + SCAN_BLOCK - 1) / SCAN_BLOCK, lo = threadIdx.x * ipt;
    uint32_t sum = 0;
    for (uint32_t j = lo; j < min(lo + ipt, n); j++) {
        const uint32_t c = control_count(P, B.last_r[j], B.last_t[j]);
        B.count[j] = c; sum += c;
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, sh, total);
    for (uint32_t j = lo; j < min(lo + ipt, n); j++) {
        const uint32_t c = B.count[j];
        B.offset[j] = run;
        const uint64_t q = ((uint64_t)run + 255) >> 8;
        if ((q << 8) < (uint64_t)run + c) bpar[q] = j;
        run += c;
    }
    if (threadIdx.x == 0) { B.offset[n] = total; meta[(L + 2) * TREE_META] = total; }
}

/* the winner's trajectory: the strict model along its path (fast mode's replacement for the trace kernels) */
__global__ __launch_bounds__(64) void path_trace_kernel(const DevProblem *__restrict__ probs, const LevelPtrs *__restrict__ levels,
                                                        int L, uint32_t n_parent, uint32_t leaf, DevResult *__restrict__ results)
{
    __shared__ __attribute__((aligned(16))) double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    const DevProblem &P = probs[0];
    stage_problem(P, lds, lds + MPC_MAX_DEPTH * CP_STRIDE);
    if (threadIdx.x != 0) return;
    DevResult &R = results[0];
    PathCtl pc;
    path_controls(P, levels, L, n_parent, leaf, pc);
    double last, cum;
    strict_path(P, lds, lds + MPC_MAX_DEPTH * CP_STRIDE, pc, L, last, cum, R.traj);
    R.best_last = last; R.best_cum = cum; R.key_bits = key_of(last); R.ordinal = leaf; R.refined = 0; R.overflow = 0;
}

/* stage 1: (key, ordinal) of the level; stage 2 (a second launch, after leaf_record_kernel where the level was not
 * stored): the winner's cumulative and last-step score from its record (slot `o`, or slot 0 of a one-node buffer) */
__global__ __launch_bounds__(TREE_FINISH_THREADS) void level_reduce_kernel(const DevProblem *__restrict__ probs, const uint64_t *__restrict__ pkey,
                                                           const uint64_t *__restrict__ pord, const uint64_t *__restrict__ pkey2,
                                                           uint32_t nblk, uint64_t *__restrict__ out)
{
    __shared__ uint64_t red[32];
    __shared__ uint64_t s_o;
    uint64_t k = ~0ull, o = ~0ull;
    for (uint32_t j0 = threadIdx.x; j0 < nblk; j0 += 8 * blockDim.x) {
        uint64_t kk[8], oo[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = j0 + u * blockDim.x;
            kk[u] = j < nblk ? pkey[j] : ~0ull; oo[u] = j < nblk ? pord[j] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) lex_min(k, o, kk[u], oo[u]);
    }
    block_argmin(k, o, red);
    if (threadIdx.x == 0) { out[0] = k; out[1] = o; s_o = o; }
    __syncthreads();
    const uint64_t ru = level_runner_up(pkey, pord, pkey2, nblk, s_o, red);
    if (threadIdx.x == 0) out[5] = keys_tie(k, ru, probs[0].wd, probs[0].wh, probs[0].ws);
}

__global__ void level_scores_kernel(LevelPtrs B, int stored, uint64_t *__restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    const uint64_t o = out[1];
    const uint64_t at = stored ? o : 0;
    out[2] = (o != ~0ull) ? (uint64_t)__double_as_longlong(B.cum[at]) : KEY_INF;
    out[3] = (o != ~0ull) ? (uint64_t)__double_as_longlong(B.last_score[at]) : KEY_INF;
}


/* one launch per level, leaf to root: the cursor (a device word) carries the node index up the parent links */
__global__ void trace_kernel(LevelPtrs L, int l, int depth, uint32_t leaf, uint32_t leaf_slot, uint32_t *__restrict__ cursor,
                             DevResult *__restrict__ results)
{
    if (threadIdx.x || blockIdx.x) return;
    DevResult &R = results[0];
    const uint32_t i = (l == depth) ? leaf_slot : *cursor;
    R.traj[l] = mpc_level_state{L.x[i], L.y[i], L.h[i], L.v[i], L.rpm[i], L.cmg[i], L.sog[i],
                                L.last_r[i], L.last_t[i], L.cum[i]};
    if (l == depth) {
        R.best_last = L.last_score[i]; R.best_cum = L.cum[i]; R.key_bits = key_of(L.last_score[i]);
        R.ordinal = leaf; R.refined = 0; R.overflow = 0;
    }
    *cursor = L.parent[i];
}

/* all levels in one launch (the blind path): leaf view by value (a sliced level is looked up unsliced, see the host) */
__global__ void trace_all_kernel(const LevelPtrs *__restrict__ levels, LevelPtrs leaf_level, int depth, uint32_t leaf,
                                 uint32_t leaf_slot, DevResult *__restrict__ results)
{
    if (threadIdx.x || blockIdx.x) return;
    DevResult &R = results[0];
    uint32_t i = leaf_slot;
    for (int l = depth; l >= 0; l--) {
        const LevelPtrs L = (l == depth) ? leaf_level : levels[l];
        R.traj[l] = mpc_level_state{L.x[i], L.y[i], L.h[i], L.v[i], L.rpm[i], L.cmg[i], L.sog[i], L.last_r[i], L.last_t[i], L.cum[i]};
        if (l == depth) {
            R.best_last = L.last_score[i]; R.best_cum = L.cum[i]; R.key_bits = key_of(L.last_score[i]);
            R.ordinal = leaf; R.refined = 0; R.overflow = 0;
        }
        i = L.parent[i];
    }
}

/* ---- the final level's short list -------------------------------------------------------------------------------
 * Of a wide final level nearly every tile lies out of reach of its checkpoint (97.6 % of the 1.2e6 tiles of the depth-7
 * semicircle tree) -- once a bound is known.  So: (1) the tiles that hold the children of ONE promising node (the level
 * above's winner) are expanded first, which leaves a bound behind; (2) tile_select_kernel lists the tiles that bound
 * does not exclude; (3) the expansion runs over the list only, and goes on tightening the bound and skipping as before.
 * Which tiles are listed changes no result: an excluded tile holds no leaf within the near-tie window of any score
 * already seen, let alone of the level's minimum. */
__global__ void seed_tiles_kernel(LevelPtrs A, uint32_t node, uint32_t *__restrict__ list)
{
    if (threadIdx.x || blockIdx.x) return;
    const uint32_t a = A.offset[node], b = A.offset[node + 1];
    list[0] = a >> 8; list[1] = (max(b, a + 1) - 1) >> 8;
}

/* 256 tiles per workgroup: their parents are consecutive nodes; every node lowers the margin of the (one or two) tiles
 * its children fall into, then a lane per tile decides */
__global__ __launch_bounds__(256) void tile_select_kernel(LevelPtrs A, uint32_t n_parent, const uint32_t *__restrict__ bpar,
                                                          uint32_t n_tiles, const uint64_t *bound, uint32_t *__restrict__ list,
                                                          uint32_t *__restrict__ count, const uint32_t *__restrict__ parent_flags)
{
    /* parent_flags (optional): the ghost flags of the PARENTS' tiles (ghost_select_kernel, when the parents' level was
     * expanded with ghosts): a ghost lowers no tile's margin, so its offsets and margin are not even read */
    __shared__ uint32_t s_m[256];                                    /* bits of non-negative floats order like the floats */
    const uint32_t t0 = blockIdx.x * 256, nt = min(256u, n_tiles - t0);
    s_m[threadIdx.x] = 0x7F800000u;
    __syncthreads();
    const uint32_t p_lo = bpar[t0], p_hi = t0 + nt < n_tiles ? min(bpar[t0 + nt] + 1, n_parent) : n_parent;
    for (uint32_t p = p_lo + threadIdx.x; p < p_hi; p += 256) {
        if (parent_flags && parent_flags[p >> 8]) continue;
        const uint32_t a = A.offset[p], b = A.offset[p + 1];
        if (b <= a) continue;
        const uint32_t bits = __float_as_uint(fmaxf(A.margin[p], 0.0f));
        const uint32_t ta = max(a >> 8, t0), tb = min((b - 1) >> 8, t0 + nt - 1);
        for (uint32_t t = ta; t <= tb; t++) atomicMin(&s_m[t - t0], bits);
    }
    __syncthreads();
    if (threadIdx.x < nt && !margin_beyond(__uint_as_float(s_m[threadIdx.x]), tree_bound_read(bound)))
        list[atomicAdd(count, 1u)] = t0 + threadIdx.x;
}

/* ---- the level above a final level: ghosts ----------------------------------------------------------------------------
 * When the level being expanded is the last one that will have children (the caller's max_levels says so), most of its
 * nodes matter neither to its own record nor to the final level's: a parent whose children all end out of reach of this
 * level's checkpoint AND whose grandchildren all end out of reach of the next one (node_margin over one and two levels,
 * against bounds from two real scores: a greedy descent from the level above's winner, tree_greedy_seed_kernel) has
 * GHOST children -- they get their place in the level (count, hence offsets and ordinals) and nothing else.  A tile all
 * of whose parents are such is flagged: ghost_fill_kernel writes 8 bytes per child of it, and the expansion -- simulation and
 * 84 bytes per child -- runs over the list of the other tiles.  Exactness: the seeds are scores of leaves that exist, so each level's minimum is at most its seed, and a ghost
 * (or a ghost's child) scores above the near-tie window of the seed. */
template <bool FAST>
__global__ __launch_bounds__(64) void tree_greedy_seed_kernel(const DevProblem *__restrict__ probs, LevelPtrs A, uint32_t lead, int level,
                                                              double refine_rel, double *__restrict__ seed, uint64_t *bound)
{
    __shared__ __attribute__((aligned(16))) double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    __shared__ double rec[12];                                        /* the best child as a one-node level */
    __shared__ float rec_margin;
    const DevProblem &P = probs[0];
    double *cpL = lds, *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    stage_problem(P, cpL, subL);
    LevelPtrs S{rec + 0, rec + 1, rec + 2, rec + 3, rec + 4, rec + 5, rec + 6, rec + 7, rec + 8, rec + 9, rec + 10,
                nullptr, nullptr, nullptr, &rec_margin};
    const uint32_t j = threadIdx.x;
    const uint32_t n1 = A.offset[lead + 1] - A.offset[lead];
    uint64_t k1 = ~0ull, o1 = j;
    if (FAST) {
        FastChild ch;
        if (j < n1) { ch = make_child_fast(P, cpL, subL, A, lead, j, level); k1 = key_of(ch.score); }
        uint64_t k = k1, o = o1;
        wave_argmin(k, o);
        k = __shfl(k, 0, 64); o = __shfl(o, 0, 64);
        if (j == o && k < KEY_INF) {
            rec[0] = ch.s.x; rec[1] = ch.s.y; rec[2] = ch.s.sh; rec[3] = ch.s.v; rec[4] = ch.s.rpm; rec[5] = ch.s.ch; rec[6] = 0;
            rec[7] = ch.r; rec[8] = ch.t; rec[9] = 0; rec[10] = 0; rec_margin = 0.0f;
        }
        k1 = k;
    } else {
        Child ch;
        if (j < n1) { ch = make_child(P, cpL, subL, A, lead, j, level); k1 = key_of(ch.score); }
        uint64_t k = k1, o = o1;
        wave_argmin(k, o);
        k = __shfl(k, 0, 64); o = __shfl(o, 0, 64);
        if (j == o && k < KEY_INF) {
            rec[0] = ch.s.x; rec[1] = ch.s.y; rec[2] = ch.s.h; rec[3] = ch.s.v; rec[4] = ch.s.rpm; rec[5] = ch.s.cmg; rec[6] = ch.s.sog;
            rec[7] = ch.r; rec[8] = ch.t; rec[9] = ch.cum; rec[10] = ch.score; rec_margin = 0.0f;
        }
        k1 = k;
    }
    __syncthreads();
    uint64_t k2 = ~0ull, o2 = j;
    if (k1 < KEY_INF) {                                               /* uniform */
        const uint32_t n2 = control_count(P, rec[7], rec[8]);
        if (j < n2) k2 = key_of(FAST ? make_child_fast(P, cpL, subL, S, 0, j, level + 1).score : make_child(P, cpL, subL, S, 0, j, level + 1).score);
        wave_argmin(k2, o2);
    }
    if (threadIdx.x == 0) {
        const double rel = FAST ? 2.0 * refine_rel : 1.0e-9;          /* the windows of tree_bound_update */
        const uint64_t ks[2] = {k1, k2};
        for (int q = 0; q < 2; q++) {
            double dmax = __longlong_as_double(-1ll);                 /* NaN: no bound */
            if (ks[q] < KEY_INF) {
                const double d = log(mpc_refine_limit(__longlong_as_double((long long)ks[q]), rel, P.wd, P.wh, P.ws) / P.wd);
                if (d >= 0.0) dmax = d;
            }
            seed[q] = dmax;
        }
        tree_bound_update(bound, k2, P, rel);                         /* the final level starts with the grandchild's score as its bound */
    }
}

/* 256 tiles of the level to be expanded per workgroup, as tile_select_kernel: flags[t] = 1 when every parent of the tile's
 * children has only ghosts for children (see above); lists the tiles that are not */
template <bool FAST>
__global__ __launch_bounds__(256) void ghost_select_kernel(const DevProblem *__restrict__ probs, LevelPtrs A, uint32_t n_parent,
                                                           const uint32_t *__restrict__ bpar, uint32_t n_tiles, int level,
                                                           const double *__restrict__ seed, uint32_t *__restrict__ flags,
                                                           uint32_t *__restrict__ real_list, uint32_t *__restrict__ real_tiles,
                                                           int children_decided)
{
    __shared__ uint32_t s_f[256];
    const DevProblem &P = probs[0];
    const uint32_t t0 = blockIdx.x * 256, nt = min(256u, n_tiles - t0);
    s_f[threadIdx.x] = 1u;
    __syncthreads();
    const double dm1 = seed[0], dm2 = seed[1];
    const double *c2 = P.cp + (level + 1) * CP_STRIDE;               /* the grandchildren's checkpoint */
    const uint32_t p_lo = bpar[t0], p_hi = t0 + nt < n_tiles ? min(bpar[t0 + nt] + 1, n_parent) : n_parent;
    for (uint32_t p = p_lo + threadIdx.x; p < p_hi; p += 256) {
        const uint32_t a = A.offset[p], b = A.offset[p + 1];
        if (b <= a) continue;
        bool ghosts = children_decided || margin_beyond(A.margin[p], dm1);   /* (the children's own level may have its record already) */
        if (ghosts) {
            float m2;
            if (FAST) m2 = node_margin<2>(P, c2[6], c2[7], A.x[p], A.y[p], A.v[p], A.rpm[p], A.h[p], A.cmg[p]);
            else {
                double sh, ch;
                sincos(A.h[p], &sh, &ch);
                m2 = node_margin<2>(P, c2[0] - 2.0 * P.reach_cx, c2[1] - 2.0 * P.reach_cy, A.x[p], A.y[p], A.v[p], A.rpm[p], sh, ch);
            }
            ghosts = margin_beyond(m2, dm2) && m2 != MARGIN_GHOST;
        }
        if (!ghosts) {
            const uint32_t ta = max(a >> 8, t0), tb = min((b - 1) >> 8, t0 + nt - 1);
            for (uint32_t t = ta; t <= tb; t++) s_f[t - t0] = 0u;
        }
    }
    __syncthreads();
    if (threadIdx.x < nt) {
        flags[t0 + threadIdx.x] = s_f[threadIdx.x];
        if (!s_f[threadIdx.x]) real_list[atomicAdd(real_tiles, 1u)] = t0 + threadIdx.x;   /* the expansion's list, in any order */
    }
}

/* the ghosts' place in the level: the child count (its control set's size) and the margin that says "ghost", for the
 * children that lie in flagged tiles.  A lane per PARENT: a control set's size is (rudders) x (throttles), each a function
 * of one last control only, and a parent's children carry at most six different rudders and five throttles -- eleven
 * small evaluations per parent instead of a parent search, a control_pick and a control_count per child */
__device__ inline uint32_t rudder_count(const DevProblem &P, double last_r)
{
    const double tol = P.tol, low_r = last_r - P.gr, high_r = last_r + P.gr;
    return 3u + (P.minr + tol < low_r && fabs(P.midr - low_r) > tol) + (fabs(P.midr - high_r) > tol && P.maxr > high_r + tol) +
           (fabs(P.minr - last_r) > tol && fabs(P.midr - last_r) > tol && fabs(P.maxr - last_r) > tol);
}

__device__ inline uint32_t throttle_count(const DevProblem &P, double last_t)
{
    const double tol = P.tol, low_t = last_t - P.gt, high_t = last_t + P.gt;
    return 2u + (P.mint + tol < low_t) + (P.maxt > high_t + tol) + (fabs(P.mint - last_t) > tol && fabs(P.maxt - last_t) > tol);
}

__global__ __launch_bounds__(256) void ghost_fill_kernel(const DevProblem *__restrict__ probs, LevelPtrs A, uint32_t n_parent,
                                                         LevelPtrs B, const uint32_t *__restrict__ flags)
{
    /* 256 consecutive parents per workgroup: their children are one run of the level; the counts go through LDS so that
     * the level is written in order, a lane per child (a lane per parent writing its own children scatters 4-byte stores) */
    __shared__ uint8_t s_cnt[256 * 30];
    __shared__ uint32_t s_run[2];
    const uint32_t p0 = blockIdx.x * 256, p = p0 + threadIdx.x;
    if (threadIdx.x == 0) { s_run[0] = A.offset[p0]; s_run[1] = A.offset[min(p0 + 256u, n_parent)]; }
    __syncthreads();
    const uint32_t base = s_run[0], end = s_run[1];
    bool any = false;
    for (uint32_t t = (base >> 8) + threadIdx.x; t <= ((max(end, base + 1) - 1) >> 8); t += 256) any |= flags[t] != 0;
    if (!__syncthreads_or(any)) return;                               /* no ghost among these parents' children */
    if (p < n_parent) {
        const uint32_t a = A.offset[p], b = A.offset[p + 1];
        const DevProblem &P = probs[0];
        const double last_r = A.last_r[p], last_t = A.last_t[p], tol = P.tol;
        /* the children's rudders and throttles in the order of control_pick, and the size of each one's own set */
        const double low_r = last_r - P.gr, high_r = last_r + P.gr, low_t = last_t - P.gt, high_t = last_t + P.gt;
        const bool r3 = P.minr + tol < low_r && fabs(P.midr - low_r) > tol;
        const bool r4 = fabs(P.midr - high_r) > tol && P.maxr > high_r + tol;
        const bool r5 = fabs(P.minr - last_r) > tol && fabs(P.midr - last_r) > tol && fabs(P.maxr - last_r) > tol;
        const bool t2 = P.mint + tol < low_t, t3 = P.maxt > high_t + tol;
        const bool t4 = fabs(P.mint - last_t) > tol && fabs(P.maxt - last_t) > tol;
        uint32_t cr[6], ct[5], nr = 3, nt = 2;
        cr[0] = rudder_count(P, P.minr); cr[1] = rudder_count(P, P.midr); cr[2] = rudder_count(P, P.maxr);
        cr[3] = cr[4] = cr[5] = 0;
        ct[0] = throttle_count(P, P.mint); ct[1] = throttle_count(P, P.maxt);
        ct[2] = ct[3] = ct[4] = 0;
        const uint32_t c_low_r = rudder_count(P, low_r), c_high_r = rudder_count(P, high_r), c_last_r = rudder_count(P, last_r);
        const uint32_t c_low_t = throttle_count(P, low_t), c_high_t = throttle_count(P, high_t), c_last_t = throttle_count(P, last_t);
        /* (selects, not run-time subscripts: cr[] / ct[] stay in registers) */
        if (r3) { cr[3] = c_low_r; nr = 4; }
        if (r4) { if (nr == 3) cr[3] = c_high_r; else cr[4] = c_high_r; nr++; }
        if (r5) { if (nr == 3) cr[3] = c_last_r; else if (nr == 4) cr[4] = c_last_r; else cr[5] = c_last_r; nr++; }
        if (t2) { ct[2] = c_low_t; nt = 3; }
        if (t3) { if (nt == 2) ct[2] = c_high_t; else ct[3] = c_high_t; nt++; }
        if (t4) { if (nt == 2) ct[2] = c_last_t; else if (nt == 3) ct[3] = c_last_t; else ct[4] = c_last_t; nt++; }
        uint32_t at = a - base;
        if (b - a == nr * nt) {                                       /* (always: the scan counted the same set) */
#pragma unroll
            for (uint32_t ti = 0; ti < 5; ti++) {
#pragma unroll
                for (uint32_t ri = 0; ri < 6; ri++) {
                    if (ti < nt && ri < nr) s_cnt[at++] = (uint8_t)(cr[ri] * ct[ti]);   /* throttle outer, rudder inner (controller.cpp:174-182) */
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t c = base + threadIdx.x; c < end; c += 256) {
        if (flags[c >> 8]) { B.count[c] = s_cnt[c - base]; B.margin[c] = MARGIN_GHOST; }
    }
}

#define TILE_LIST_MIN 8192u      /* tiles (2.1e6 leaves) from which a final level is worth the list's two launches and one read-back */

/* ---- host ------------------------------------------------------------------------------------------------------ */
static void free_level(TreeLevelBuf &b)
{
    (void)hipFree(b.x); (void)hipFree(b.y); (void)hipFree(b.h); (void)hipFree(b.v); (void)hipFree(b.rpm);
    (void)hipFree(b.cmg); (void)hipFree(b.sog); (void)hipFree(b.last_r); (void)hipFree(b.last_t); (void)hipFree(b.cum);
    (void)hipFree(b.last_score); (void)hipFree(b.parent); (void)hipFree(b.count); (void)hipFree(b.offset);
    (void)hipFree(b.margin);
    b = TreeLevelBuf();
}

void mpc_tree_free(mpc_hip_handle *h)
{
    for (auto &b : h->tree) free_level(b);
    h->tree.clear();
    (void)hipFree(h->d_scan_tmp); h->d_scan_tmp = nullptr; h->cap_scan_tmp = 0;
    (void)hipFree(h->d_level_ptrs); h->d_level_ptrs = nullptr;
    (void)hipFree(h->d_bpar); h->d_bpar = nullptr; h->cap_bpar = 0;
    (void)hipFree(h->d_tile_list); h->d_tile_list = nullptr; h->cap_tile_list = 0;
    (void)hipFree(h->d_tree_seed); h->d_tree_seed = nullptr;
    (void)hipFree(h->d_tile_flags); h->d_tile_flags = nullptr; h->cap_tile_flags = 0;
    (void)hipFree(h->d_tree_meta); h->d_tree_meta = nullptr;
    (void)hipHostFree(h->h_tree_meta); h->h_tree_meta = nullptr;
}

static int ensure_level(mpc_hip_handle *h, TreeLevelBuf &b, size_t n)
{
    if (n <= b.cap) return MPC_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    free_level(b);
    size_t cap = 1024;
    while (cap < n) cap <<= 1;
    if (cap > n + n / 4 && n > (1u << 20)) cap = n + n / 8;      /* large levels: little slack */
    double **dbl[] = {&b.x, &b.y, &b.h, &b.v, &b.rpm, &b.cmg, &b.sog, &b.last_r, &b.last_t, &b.cum, &b.last_score};
    for (double **p : dbl) HIP_TRY(hipMalloc(p, sizeof(double) * cap));
    HIP_TRY(hipMalloc(&b.parent, sizeof(uint32_t) * cap));
    HIP_TRY(hipMalloc(&b.count, sizeof(uint32_t) * cap));
    HIP_TRY(hipMalloc(&b.offset, sizeof(uint32_t) * (cap + 1)));
    HIP_TRY(hipMalloc(&b.margin, sizeof(float) * cap));
    b.cap = cap;
    return MPC_OK;
}

static int publish_level(mpc_hip_handle *h, int l)
{
    h->h_level_ptrs[l] = ptrs(h->tree[l]);
    HIP_TRY(hipMemcpyAsync(h->d_level_ptrs + l, &h->h_level_ptrs[l], sizeof(LevelPtrs), hipMemcpyHostToDevice, h->stream));
    return MPC_OK;
}

/* resume (mpc_hip_solve_tree_deep): levels 0..level of the last search on this handle -- same problem, stored in full
 * (h->tree_full_upto) -- are taken as they are and the search goes on from there; best_leaf = that level's winner;
 * store_last: the last level asked for IS stored and nothing is expanded from it (extending the stored part) */
struct TreeResume {
    int level; uint64_t best_leaf; bool store_last;
    bool record_known;      /* the record of level + 1 is known already (a shallower search found it): its nodes matter as parents only
                             * -- ghosts by their grandchildren alone, no reduction of that level */
    uint64_t beat_key;      /* the final level's best score so far (KEY_INF: none): a search that does not beat it returns no trajectory */
};

static int solve_tree_once(mpc_hip_handle *h, const mpc_problem *h_problem, int max_levels, int precision,
                           mpc_time_fn now, void *now_ctx, double end_time, uint64_t max_nodes,
                           int split_level, uint32_t shard, uint32_t n_shards,
                           mpc_tree_level *h_level_best, mpc_result *h_result, const TreeResume *resume = nullptr)
{
    if (resume && (resume->level < 1 || resume->level > h->tree_full_upto || precision != h->tree_precision || resume->level >= max_levels + (resume->store_last ? 1 : 0) ||
                   (n_shards > 1 && split_level != resume->level))) {
        mpc_set_err("solve_tree: nothing stored to resume from");
        return MPC_ERR_INVALID;
    }
    const bool fast = precision == MPC_PREC_FAST;
    memset(h_result, 0, sizeof *h_result);
    const bool own_cap = max_nodes == 0;      /* no caller budget: the library's own storage cap */
    if (max_nodes == 0) max_nodes = 1ull << 28;
    if (max_nodes > 0xFFFFFFF0ull) max_nodes = 0xFFFFFFF0ull;
    if (int s = mpc_hip_upload_problems(h, h_problem, nullptr, 1)) return s;
    if ((int)h->tree.size() < max_levels + 1) h->tree.resize(max_levels + 1);
    for (auto &b : h->tree) b.first = 0;
    if (!h->d_level_ptrs) {
        HIP_TRY(hipMalloc(&h->d_level_ptrs, sizeof(LevelPtrs) * (MPC_MAX_DEPTH + 2)));
        h->h_level_ptrs.assign(MPC_MAX_DEPTH + 2, LevelPtrs{});
    }

    hipEvent_t *ev = &h->ev[4 * (h->runs % TIMING_RING)];
    h->timed[h->runs % TIMING_RING] = 1;
    h->total_from_k1[h->runs % TIMING_RING] = 0;
    HIP_TRY(hipEventRecord(ev[0], h->stream));
    if (!resume) {
        if (int s = ensure_level(h, h->tree[0], 1)) return s;
        h->tree[0].n = h->tree[0].full_n = 1;
        if (int s = publish_level(h, 0)) return s;
        h->tree_full_upto = 0;
        h->tree_precision = precision;
    } else {
        h->tree[resume->level].n = h->tree[resume->level].full_n;          /* (a piece before this one left its slice there) */
        if (int s = publish_level(h, resume->level)) return s;
        h->tree_full_upto = resume->level;
    }

    h->tree_stopped_by_width = false;
    int ghost_level = -1;                     /* the level (if any) that was expanded with ghosts: h->d_tile_flags are its tiles' */
    int completed = resume ? resume->level : 0;
    uint64_t best_leaf = resume ? resume->best_leaf : 0, steps = 0, evaluated = 0;
    bool leaf_stored = true, overflow = false, stop = false;
    uint32_t last_tie = 0;
    uint64_t last_key = KEY_INF;              /* the deepest completed level's winning score (sync path) */
    auto record_level = [&](int it, uint64_t nchild, const uint64_t *w, bool stored, uint64_t tie) {
        completed = it + 1;
        best_leaf = w[1];
        last_tie = tie != 0;
        leaf_stored = stored;
        evaluated += nchild;
        steps += nchild * (uint64_t)h_problem->n_substeps;
        if (h_level_best) {
            h_level_best[it].n_leaves = nchild;
            h_level_best[it].best_leaf = w[1];
            memcpy(&h_level_best[it].best_score, &w[3], sizeof(double));
            memcpy(&h_level_best[it].best_cumulative, &w[2], sizeof(double));
            h_level_best[it].libm_tie = tie != 0; h_level_best[it].reserved = 0;
        }
    };
    /* every shard has expanded the split level in full; from there on this one keeps a contiguous slice of its nodes
     * (DFS order is preserved within and across shards) */
    auto slice_level = [&](int level, bool stored) -> int {
        TreeLevelBuf &B = h->tree[level];
        const uint64_t nchild = B.full_n;
        if (!stored) { mpc_set_err("solve_tree_shard: split_level must be below max_levels"); return MPC_ERR_INVALID; }
        if (nchild < n_shards) { mpc_set_err("solve_tree_shard: the split level has fewer nodes than shards"); return MPC_ERR_INVALID; }
        const uint64_t base = nchild / n_shards, extra = nchild % n_shards;
        B.first = (size_t)(shard * base + std::min<uint64_t>(shard, extra));
        B.n = (size_t)(base + (shard < extra ? 1 : 0));
        return publish_level(h, level);
    };

    /* ---- the blind shallow levels (tree_post_kernel): no host round trip until the last of them ---- */
    int blind = 0;
    {
        const int lim = n_shards > 1 ? std::min(max_levels, split_level) : max_levels;
        uint64_t bound = 1;
        static const bool sync_levels = getenv("MPC_HIP_TREE_SYNC_LEVELS") != nullptr;     /* diagnostic: every level the long way */
        /* a caller's node budget bounds memory and work: only levels whose 30^l bound fits it are expanded blind (their
         * buffers and grids are sized for the bound); the levels beyond go the synchronised way, which sizes by the real
         * counts and stops at the budget before allocating */
        while (!resume && !sync_levels && blind < lim && bound * 30 <= BLIND_MAX_NODES && (own_cap || bound * 30 <= max_nodes)) { bound *= 30; blind++; }
    }
    if (resume && n_shards > 1) { if (int s = slice_level(resume->level, true)) return s; }
    if (now && now(now_ctx) >= end_time) blind = 0;                       /* the loop below breaks at once */
    if (!h->d_tree_meta) {
        HIP_TRY(hipMalloc(&h->d_tree_meta, sizeof(uint64_t) * (TREE_META * (MPC_MAX_DEPTH + 2) + 2)));   /* + the pruning bound's two words */
        HIP_TRY(hipHostMalloc(&h->h_tree_meta, sizeof(uint64_t) * TREE_META * (MPC_MAX_DEPTH + 2)));
    }
    if (blind > 0) {
        uint64_t bound = 1;
        for (int l = 1; l <= blind; l++) {
            bound *= 30;
            TreeLevelBuf &B = h->tree[l];
            B.first = 0;
            if (int s = ensure_level(h, B, l < max_levels ? bound : 1)) return s;
            if (int s = publish_level(h, l)) return s;
        }
        const size_t nbound = (size_t)bound / 256 + 2;
        if (nbound > h->cap_bpar) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_bpar); h->d_bpar = nullptr; h->cap_bpar = 0;
            HIP_TRY(hipMalloc(&h->d_bpar, sizeof(uint32_t) * nbound));
            h->cap_bpar = nbound;
        }
        if (int s = mpc_ensure_partials(h, (size_t)(bound + 255) / 256)) return s;
    } else if (h->cap_bpar < 2) {
        (void)hipFree(h->d_bpar); h->d_bpar = nullptr; h->cap_bpar = 0;
        HIP_TRY(hipMalloc(&h->d_bpar, sizeof(uint32_t) * 64));
        h->cap_bpar = 64;
    }
    /* the pruning bound of the unstored final level (tree_bound_update): "none yet" = all ones (as a double: NaN) */
    uint64_t *d_bound = h->d_tree_meta + TREE_META * (MPC_MAX_DEPTH + 2);
    /* (the pieces of a resumed search look for the level's ONE winner: what an earlier piece found bounds the later ones) */
    if (!(resume && n_shards > 1 && shard > 0)) HIP_TRY(hipMemsetAsync(d_bound, 0xFF, sizeof(uint64_t) * 2, h->stream));
    if (!resume)
        hipLaunchKernelGGL(root_kernel, dim3(1), dim3(1), 0, h->stream, h->d_probs, ptrs(h->tree[0]), (int)fast, h->d_bpar,
                           blind > 0 ? h->d_tree_meta : (uint64_t *)nullptr);
    HIP_TRY(hipGetLastError());
    if (blind > 0) {
        uint64_t pb = 1;
        for (int it = 0; it < blind; it++) {
            const uint64_t cb = pb * 30;                                   /* <= 30 children per node (6 x 5) */
            const uint32_t nblk = (uint32_t)((cb + 255) / 256);
            const bool stored = it + 1 < max_levels;
            const uint64_t *mp = h->d_tree_meta + (size_t)it * TREE_META;
            TreeLevelBuf &A = h->tree[it], &B = h->tree[it + 1];
            if (fast) {
                if (stored) hipLaunchKernelGGL(expand_fast_kernel<true>, dim3(nblk), dim3(256), 0, h->stream, h->d_probs, ptrs(A), (uint32_t)pb, h->d_bpar, ptrs(B),
                                               (uint32_t)cb, it, (double)MPC_REFINE_REL, 1u, h->d_pkey, h->d_pord, h->d_pcnt, mp, d_bound, (const uint32_t *)nullptr, 0u);
                else hipLaunchKernelGGL(expand_fast_kernel<false>, dim3(nblk), dim3(256), 0, h->stream, h->d_probs, ptrs(A), (uint32_t)pb, h->d_bpar, ptrs(B),
                                        (uint32_t)cb, it, (double)MPC_REFINE_REL, 1u, h->d_pkey, h->d_pord, h->d_pcnt, mp, d_bound, (const uint32_t *)nullptr, 0u);
                hipLaunchKernelGGL(tree_post_kernel<true>, dim3(1), dim3(256), 0, h->stream, h->d_probs, h->d_level_ptrs, it, (int)stored,
                                   (double)MPC_REFINE_REL, 1u, h->d_pkey, h->d_pord, h->d_pcnt, h->d_pkey2, h->d_winners, h->d_tree_meta,
                                   (int)(stored && it + 1 < blind), h->d_bpar);
            } else {
                if (stored) hipLaunchKernelGGL(expand_kernel<true>, dim3(nblk), dim3(256), 0, h->stream, h->d_probs, ptrs(A), (uint32_t)pb, h->d_bpar, ptrs(B),
                                               (uint32_t)cb, it, 1u, h->d_pkey, h->d_pord, h->d_pkey2, mp, d_bound, (const uint32_t *)nullptr, 0u);
                else hipLaunchKernelGGL(expand_kernel<false>, dim3(nblk), dim3(256), 0, h->stream, h->d_probs, ptrs(A), (uint32_t)pb, h->d_bpar, ptrs(B),
                                        (uint32_t)cb, it, 1u, h->d_pkey, h->d_pord, h->d_pkey2, mp, d_bound, (const uint32_t *)nullptr, 0u);
                hipLaunchKernelGGL(tree_post_kernel<false>, dim3(1), dim3(256), 0, h->stream, h->d_probs, h->d_level_ptrs, it, (int)stored,
                                   0.0, 1u, h->d_pkey, h->d_pord, h->d_pcnt, h->d_pkey2, h->d_winners, h->d_tree_meta,
                                   (int)(stored && it + 1 < blind), h->d_bpar);
            }
            pb = cb;
            if (!stored) { blind = it + 1; break; }
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->h_tree_meta, h->d_tree_meta, sizeof(uint64_t) * TREE_META * (size_t)(blind + 1), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        /* deadline (controller.cpp:101-103): if the clock has passed end_time by now, the levels that completed before it
         * still count -- each level's completion time is the caller's clock now minus the GPU's own (100 MHz) clock
         * difference between that level's reduction and the last one's (a clock that does not run in seconds never gets here) */
        const double t_after = now ? now(now_ctx) : 0.0;
        const uint64_t clk_ref = h->h_tree_meta[(size_t)blind * TREE_META + 6];
        for (int it = 0; it < blind; it++) {
            const uint64_t *m = h->h_tree_meta + (size_t)(it + 1) * TREE_META;
            const uint64_t nchild = m[0];
            const bool stored = it + 1 < max_levels;
            if (!own_cap && nchild > max_nodes) { stop = true; break; }   /* wider than the caller's budget: ends the search */
            if (fast && m[5]) { overflow = true; break; }                 /* > LIST_CAP tied leaves: the strict expansion decides */
            if (now && t_after >= end_time && t_after - (double)(clk_ref - m[6]) * 1.0e-8 >= end_time) { stop = true; break; }
            h->tree[it + 1].n = h->tree[it + 1].full_n = stored ? nchild : 1;
            if (stored && (n_shards == 1 || it + 1 <= split_level)) h->tree_full_upto = it + 1;
            record_level(it, nchild, m + 1, stored, m[7]);
            if (n_shards > 1 && it + 1 == split_level) { if (int s = slice_level(it + 1, stored)) return s; }
            if (!stored) { stop = true; break; }
        }
        if (completed < blind) stop = true;
    }
    for (int it = resume ? resume->level : blind; it < max_levels && !stop && !overflow; it++) {
        if (now && now(now_ctx) >= end_time) break;                       /* controller.cpp:101-103 */
        TreeLevelBuf &A = h->tree[it];
        const uint32_t n = (uint32_t)A.n;
        /* children per node and their offsets */
        const uint32_t ntile = (n + SCAN_TILE - 1) / SCAN_TILE;
        if ((size_t)ntile + 2 > h->cap_scan_tmp) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_scan_tmp); h->d_scan_tmp = nullptr;
            HIP_TRY(hipMalloc(&h->d_scan_tmp, sizeof(uint32_t) * ((size_t)ntile + 2) * 2));
            h->cap_scan_tmp = ((size_t)ntile + 2) * 2;
        }
        /* (every expansion leaves its children's own child counts behind; only the root has none yet) */
        if (it == 0) hipLaunchKernelGGL(count_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n);
        const LevelPtrs PA = ptrs(A);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(ntile), dim3(SCAN_BLOCK), 0, h->stream, PA.count, n, h->d_scan_tmp);
        hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_BLOCK), 0, h->stream, h->d_scan_tmp, ntile);
        const size_t nbound = (size_t)n * 30 / 256 + 2;                    /* <= 30 children per node (6 x 5) */
        if (nbound > h->cap_bpar) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_bpar); h->d_bpar = nullptr; h->cap_bpar = 0;
            HIP_TRY(hipMalloc(&h->d_bpar, sizeof(uint32_t) * nbound));
            h->cap_bpar = nbound;
        }
        hipLaunchKernelGGL(scan_apply_kernel, dim3(ntile), dim3(SCAN_BLOCK), 0, h->stream, PA.count, n, h->d_scan_tmp, PA.offset, h->d_bpar);
        HIP_TRY(hipGetLastError());
        uint32_t tot[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(h->h_pair, h->d_scan_tmp + ntile, sizeof(uint32_t) * 2, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(tot, h->h_pair, sizeof tot);
        /* a level wider than the caller's budget (or than 32-bit ordinals) ends the search like a deadline; one that is
         * only wider than the library's own storage cap is still expanded -- without stores, as the final level */
        const bool too_wide = tot[0] > max_nodes;
        if (tot[1] != 0 || (too_wide && !own_cap)) { h->tree_stopped_by_width = true; break; }
        const uint32_t nchild = tot[0];
        TreeLevelBuf &B = h->tree[it + 1];
        /* the last level asked for is never a parent level: expanded without stores (strict: into a one-node buffer
         * that receives the winner's record; fast: the winner comes from its path) */
        const bool stored = (it + 1 < max_levels || (resume && resume->store_last)) && !too_wide;
        B.first = 0;
        if (int s = ensure_level(h, B, stored ? nchild : 1)) {
            if (s == MPC_ERR_HIP) { (void)hipGetLastError(); break; }     /* out of memory: stop like a deadline */
            return s;
        }
        B.n = B.full_n = nchild;
        if (int s = publish_level(h, it + 1)) return s;
        uint32_t n_tiles = (nchild + 255) / 256;
        if (int s = mpc_ensure_partials(h, (n_tiles + EXPAND_TILES - 1) / EXPAND_TILES + 4096)) return s;
        /* diagnostic, with MPC_HIP_TREE_SYNC_LEVELS: the list (and the ghosts) from this many tiles on (1: for every final level
         * -- how the random runs of tools/fuzz_tree.py reach these paths with their shallow trees) */
        static const uint32_t list_min = getenv("MPC_HIP_TREE_LIST_MIN") ? (uint32_t)strtoul(getenv("MPC_HIP_TREE_LIST_MIN"), nullptr, 10) : TILE_LIST_MIN;
        /* a wide final level: the short list of its tiles (seed_tiles_kernel / tile_select_kernel above) */
        const uint32_t *d_list = nullptr;
        bool empty_level = false;              /* no node of the level can matter: nothing is expanded, nothing selected */
        if (!stored && n_tiles >= list_min) {
            if ((size_t)n_tiles + 4 > h->cap_tile_list) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                (void)hipFree(h->d_tile_list); h->d_tile_list = nullptr; h->cap_tile_list = 0;
                HIP_TRY(hipMalloc(&h->d_tile_list, sizeof(uint32_t) * ((size_t)n_tiles + 4)));
                h->cap_tile_list = (size_t)n_tiles + 4;
            }
            /* the node whose children go first: the level above's winner where this shard holds it, else the middle one */
            const uint64_t lead = completed == it && best_leaf >= A.first && best_leaf - A.first < n ? best_leaf - A.first : n / 2;
            HIP_TRY(hipMemsetAsync(h->d_tile_list, 0, sizeof(uint32_t), h->stream));
            hipLaunchKernelGGL(seed_tiles_kernel, dim3(1), dim3(1), 0, h->stream, ptrs(A), (uint32_t)lead, h->d_tile_list + 1);
            if (fast)
                hipLaunchKernelGGL(expand_fast_kernel<false>, dim3(1), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n, h->d_bpar, ptrs(B),
                                   nchild, it, (double)MPC_REFINE_REL, 2u, h->d_pkey, h->d_pord, h->d_pcnt, (const uint64_t *)nullptr, d_bound,
                                   (const uint32_t *)h->d_tile_list + 1, 2u);
            else
                hipLaunchKernelGGL(expand_kernel<false>, dim3(1), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n, h->d_bpar, ptrs(B), nchild, it,
                                   2u, h->d_pkey, h->d_pord, h->d_pkey2, (const uint64_t *)nullptr, d_bound, (const uint32_t *)h->d_tile_list + 1, 2u);
            hipLaunchKernelGGL(tile_select_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, h->stream, ptrs(A), n, h->d_bpar, n_tiles,
                               (const uint64_t *)d_bound, h->d_tile_list + 4, h->d_tile_list,
                               ghost_level == it && A.first == 0 ? (const uint32_t *)h->d_tile_flags : (const uint32_t *)nullptr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h->h_pair, h->d_tile_list, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            uint32_t listed = 0;
            memcpy(&listed, h->h_pair, sizeof listed);
            if (listed > n_tiles) { mpc_set_err("solve_tree: tile selection overran its list"); return MPC_ERR_HIP; }
            /* (nothing listed: no tile within reach of a bound that came from elsewhere -- an earlier piece, the greedy descent) */
            d_list = h->d_tile_list + 4; n_tiles = listed; empty_level = listed == 0;
        }
        /* the level above a final level (the last one asked for comes next): its tiles of ghosts (ghost_select_kernel) */
        static const bool ghosts_on = !(getenv("MPC_HIP_TREE_GHOSTS") && atoi(getenv("MPC_HIP_TREE_GHOSTS")) == 0);   /* diagnostic: 0 = none */
        const bool known_record = resume && resume->record_known && it == resume->level && stored;
        if (stored && ghosts_on && it + 2 == max_levels && completed == it && it >= 1 && n_tiles >= list_min) {
            if ((size_t)n_tiles + 4 > h->cap_tile_list) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                (void)hipFree(h->d_tile_list); h->d_tile_list = nullptr; h->cap_tile_list = 0;
                HIP_TRY(hipMalloc(&h->d_tile_list, sizeof(uint32_t) * ((size_t)n_tiles + 4)));
                h->cap_tile_list = (size_t)n_tiles + 4;
            }
            if ((size_t)n_tiles > h->cap_tile_flags) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                (void)hipFree(h->d_tile_flags); h->d_tile_flags = nullptr; h->cap_tile_flags = 0;
                HIP_TRY(hipMalloc(&h->d_tile_flags, sizeof(uint32_t) * (size_t)n_tiles));
                h->cap_tile_flags = (size_t)n_tiles;
            }
            if (!h->d_tree_seed) HIP_TRY(hipMalloc(&h->d_tree_seed, sizeof(double) * 2));
            /* the greedy descent starts from the level's winner: of this shard's slice where it lies in it (else the middle
             * node), of the WHOLE level in a resumed search -- the level above the slice is there in full, and the pieces
             * look for the level's one winner, so a score from anywhere bounds them */
            TreeLevelBuf lead_view = A;
            uint64_t lead = best_leaf >= A.first && best_leaf - A.first < n ? best_leaf - A.first : n / 2;
            if (resume && it == resume->level) { lead_view.first = 0; lead = resume->best_leaf < A.full_n ? resume->best_leaf : A.first + n / 2; }
            HIP_TRY(hipMemsetAsync(h->d_tile_list, 0, sizeof(uint32_t), h->stream));
            if (fast) {
                hipLaunchKernelGGL(tree_greedy_seed_kernel<true>, dim3(1), dim3(64), 0, h->stream, h->d_probs, ptrs(lead_view), (uint32_t)lead, it,
                                   (double)MPC_REFINE_REL, h->d_tree_seed, d_bound);
                hipLaunchKernelGGL(ghost_select_kernel<true>, dim3((n_tiles + 255) / 256), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n,
                                   h->d_bpar, n_tiles, it, (const double *)h->d_tree_seed, h->d_tile_flags, h->d_tile_list + 4, h->d_tile_list,
                                   (int)known_record);
            } else {
                hipLaunchKernelGGL(tree_greedy_seed_kernel<false>, dim3(1), dim3(64), 0, h->stream, h->d_probs, ptrs(lead_view), (uint32_t)lead, it,
                                   0.0, h->d_tree_seed, d_bound);
                hipLaunchKernelGGL(ghost_select_kernel<false>, dim3((n_tiles + 255) / 256), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n,
                                   h->d_bpar, n_tiles, it, (const double *)h->d_tree_seed, h->d_tile_flags, h->d_tile_list + 4, h->d_tile_list,
                                   (int)known_record);
            }
            hipLaunchKernelGGL(ghost_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n, ptrs(B),
                               (const uint32_t *)h->d_tile_flags);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h->h_pair, h->d_tile_list, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            uint32_t real = 0;
            memcpy(&real, h->h_pair, sizeof real);
            if (real > n_tiles) { mpc_set_err("solve_tree: ghost selection overran its list"); return MPC_ERR_HIP; }
            /* (no real tile: a slice all of whose nodes are out of reach of bounds that came from outside it) */
            d_list = h->d_tile_list + 4; n_tiles = real; empty_level = real == 0;
            ghost_level = it + 1;
        }
        const uint32_t tpb = std::min<uint32_t>(EXPAND_TILES, std::max<uint32_t>(1, n_tiles / 4096));   /* >= 4096 workgroups first */
        const uint32_t nblk = (n_tiles + tpb - 1) / tpb;
        if (int s = mpc_ensure_partials(h, nblk)) return s;
        if (empty_level) {
            /* what the reductions report for a level without a finite score */
            h->h_pair[0] = KEY_INF; h->h_pair[1] = ~0ull; h->h_pair[2] = KEY_INF; h->h_pair[3] = KEY_INF; h->h_pair[4] = 0; h->h_pair[5] = 0;
        } else if (fast) {
            if (stored)
                hipLaunchKernelGGL(expand_fast_kernel<true>, dim3(nblk), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n, h->d_bpar, ptrs(B),
                                   nchild, it, (double)MPC_REFINE_REL, tpb, h->d_pkey, h->d_pord, h->d_pcnt, (const uint64_t *)nullptr, d_bound,
                                   d_list, n_tiles);
            else
                hipLaunchKernelGGL(expand_fast_kernel<false>, dim3(nblk), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n, h->d_bpar, ptrs(B),
                                   nchild, it, (double)MPC_REFINE_REL, tpb, h->d_pkey, h->d_pord, h->d_pcnt, (const uint64_t *)nullptr, d_bound,
                                   d_list, n_tiles);
            if (!known_record)
                hipLaunchKernelGGL(tree_finish_kernel, dim3(1), dim3(TREE_FINISH_THREADS), 0, h->stream, h->d_probs, h->d_level_ptrs, it, n, nchild,
                                   (double)MPC_REFINE_REL, tpb, h->d_pkey, h->d_pord, h->d_pcnt, nblk, h->d_bpar, h->d_winners, d_list, n_tiles);
        } else {
            if (stored)
                hipLaunchKernelGGL(expand_kernel<true>, dim3(nblk), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n, h->d_bpar, ptrs(B), nchild, it,
                                   tpb, h->d_pkey, h->d_pord, h->d_pkey2, (const uint64_t *)nullptr, d_bound, d_list, n_tiles);
            else
                hipLaunchKernelGGL(expand_kernel<false>, dim3(nblk), dim3(256), 0, h->stream, h->d_probs, ptrs(A), n, h->d_bpar, ptrs(B), nchild, it,
                                   tpb, h->d_pkey, h->d_pord, h->d_pkey2, (const uint64_t *)nullptr, d_bound, d_list, n_tiles);
            if (!known_record) {
                hipLaunchKernelGGL(level_reduce_kernel, dim3(1), dim3(TREE_FINISH_THREADS), 0, h->stream, h->d_probs, h->d_pkey, h->d_pord, h->d_pkey2, nblk, h->d_winners);
                if (!stored)
                    hipLaunchKernelGGL(leaf_record_kernel, dim3(1), dim3(64), 0, h->stream, h->d_probs, ptrs(A), n, ptrs(B), it, h->d_winners);
                hipLaunchKernelGGL(level_scores_kernel, dim3(1), dim3(1), 0, h->stream, ptrs(B), (int)stored, h->d_winners);
            }
        }
        HIP_TRY(hipGetLastError());
        if (known_record && !empty_level) {          /* nobody asks for this level's winner again */
            h->h_pair[0] = KEY_INF; h->h_pair[1] = ~0ull; h->h_pair[2] = KEY_INF; h->h_pair[3] = KEY_INF; h->h_pair[4] = 0; h->h_pair[5] = 0;
        } else if (!empty_level) {
            HIP_TRY(hipMemcpyAsync(h->h_pair, h->d_winners, sizeof(uint64_t) * 6, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        if (fast && h->h_pair[4]) { overflow = true; break; }            /* > LIST_CAP tied leaves: the strict expansion decides */
        if (now && now(now_ctx) >= end_time) break;                       /* finished too late: iteration discarded */
        record_level(it, nchild, h->h_pair, stored, h->h_pair[5]);
        last_key = h->h_pair[0];
        /* (stored in full: all children of a full level, none of them a ghost) */
        if (stored && ghost_level != it + 1 && h->tree_full_upto == it && !(n_shards > 1 && it >= split_level)) h->tree_full_upto = it + 1;
        if (n_shards > 1 && it + 1 == split_level) { if (int s = slice_level(it + 1, stored)) return s; }
        if (!stored) { h->tree_stopped_by_width = too_wide; break; }     /* nothing to expand the next level from */
    }
    HIP_TRY(hipEventRecord(ev[1], h->stream));
    if (overflow) {
        HIP_TRY(hipEventRecord(ev[3], h->stream));
        h->runs++;
        return MPC_INTERNAL_RERUN_STRICT;
    }
    if (completed == 0 || best_leaf == ~0ull) {
        HIP_TRY(hipEventRecord(ev[3], h->stream));
        h->runs++;
        h_result->status = MPC_ERR_NO_ITERATION;                          /* controller.cpp:196 */
        h_result->levels = resume ? completed : 0;                        /* (a piece that holds nothing within reach of the level's winner still counted its leaves) */
        return MPC_ERR_NO_ITERATION;
    }
    if (resume && completed == max_levels && last_key >= resume->beat_key) {
        /* a piece whose best does not beat an earlier piece's (ties keep the earlier: DFS order): the same answer, no trajectory */
        HIP_TRY(hipEventRecord(ev[3], h->stream));
        h->runs++;
        h_result->status = MPC_ERR_NO_ITERATION;
        h_result->levels = completed;
        return MPC_ERR_NO_ITERATION;
    }
    if (fast) {
        hipLaunchKernelGGL(path_trace_kernel, dim3(1), dim3(64), 0, h->stream, h->d_probs, h->d_level_ptrs, completed - 1,
                           (uint32_t)h->tree[completed - 1].n, (uint32_t)best_leaf, h->d_results);
    } else {
        /* best_leaf indexes level `completed` as it was EXPANDED -- whole, from the buffer's start.  A sharded search
         * that ends at its split level has sliced that level since (first != 0 for shard > 0): the leaf is looked up
         * in the unsliced view.  Levels above it are reached through parent links, which are relative to the view
         * their children were expanded from (the sliced one below a split level): the published pointers. */
        TreeLevelBuf view = h->tree[completed];
        view.first = 0;
        hipLaunchKernelGGL(trace_all_kernel, dim3(1), dim3(1), 0, h->stream, h->d_level_ptrs, ptrs(view), completed,
                           (uint32_t)best_leaf, leaf_stored ? (uint32_t)best_leaf : 0u, h->d_results);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[3], h->stream));
    h->runs++;
    HIP_TRY(hipMemcpyAsync(h->h_results, h->d_results, sizeof(DevResult), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const DevResult &R = h->h_results[0];
    float k1 = 0, total = 0;
    (void)hipEventElapsedTime(&k1, ev[0], ev[1]);
    (void)hipEventElapsedTime(&total, ev[0], ev[3]);
    h_result->levels = completed;
    h_result->best_ordinal = R.ordinal;
    h_result->best_last = R.best_last; h_result->best_cumulative = R.best_cum;
    h_result->first_rudder = R.traj[1].rudder; h_result->first_throttle = R.traj[1].throttle;
    h_result->evaluated = evaluated; h_result->rollout_steps = steps;
    h_result->libm_tie = last_tie;
    h_result->kernel_ms = k1; h_result->total_ms = total;
    memcpy(h_result->trajectory, R.traj, sizeof(mpc_level_state) * (size_t)(completed + 1));
    h_result->status = MPC_OK;
    if (!isfinite(h_result->first_rudder) || !isfinite(h_result->first_throttle)) {
        h_result->status = MPC_ERR_NON_FINITE;                            /* controller.cpp:201-202 */
        return MPC_ERR_NON_FINITE;
    }
    h->ran = false;
    return MPC_OK;
}

static int solve_tree_checked(mpc_hip_handle *h, const mpc_problem *h_problem, int max_levels, int precision,
                              mpc_time_fn now, void *now_ctx, double end_time, uint64_t max_nodes,
                              int split_level, uint32_t shard, uint32_t n_shards,
                              mpc_tree_level *h_level_best, mpc_result *h_result, const TreeResume *resume = nullptr)
{
    if (!h || !h_problem || !h_result) { mpc_set_err("solve_tree: NULL argument"); return MPC_ERR_INVALID; }
    if (precision != MPC_PREC_F64 && precision != MPC_PREC_FAST) { mpc_set_err("unknown precision"); return MPC_ERR_INVALID; }
    if (n_shards == 0 || shard >= n_shards || (n_shards > 1 && split_level < 1)) {
        mpc_set_err("solve_tree_shard: shard index / count / split level out of range");
        return MPC_ERR_INVALID;
    }
    if (int s = mpc_check_problem(h_problem, precision)) return s;
    if (max_levels > h_problem->depth) { mpc_set_err("solve_tree: max_levels exceeds the checkpoints supplied"); return MPC_ERR_INVALID; }
    if (precision == MPC_PREC_FAST)
        for (int s = 0; s < h_problem->n_substeps; s++)
            if (h_problem->substeps[s] > 0.1000001) {
                mpc_set_err("MPC_PREC_FAST needs substeps <= 0.1 s (VehicleState.h:50); use MPC_PREC_F64");
                return MPC_ERR_INVALID;
            }
    if (int s = mpc_use_device(h)) return s;
    if (precision == MPC_PREC_FAST) if (int s = mpc_check_fast_model(h, 1)) return s;
    if (max_levels <= 0) {                        /* controller.cpp:80 never entered -> :196 */
        memset(h_result, 0, sizeof *h_result);
        h_result->status = MPC_ERR_NO_ITERATION;
        return MPC_ERR_NO_ITERATION;
    }
    int s = solve_tree_once(h, h_problem, max_levels, precision, now, now_ctx, end_time, max_nodes, split_level, shard,
                            n_shards, h_level_best, h_result, resume);
    if (s == MPC_INTERNAL_RERUN_STRICT) {
        /* (a resumed search has the fast expansion's node records above it: the strict one cannot go on from those) */
        if (resume) { mpc_set_err("solve_tree: too many tied leaves in a resumed search"); return MPC_ERR_INVALID; }
        s = solve_tree_once(h, h_problem, max_levels, MPC_PREC_F64, now, now_ctx, end_time, max_nodes, split_level, shard,
                            n_shards, h_level_best, h_result);
    }
    return s;
}

extern "C" int mpc_hip_solve_tree(mpc_hip_handle *h, const mpc_problem *h_problem, int max_levels, int precision,
                                  mpc_time_fn now, void *now_ctx, double end_time, uint64_t max_nodes,
                                  mpc_tree_level *h_level_best, mpc_result *h_result)
{
    return solve_tree_checked(h, h_problem, max_levels, precision, now, now_ctx, end_time, max_nodes, 0, 0, 1,
                              h_level_best, h_result);
}

extern "C" int mpc_hip_solve_tree_shard(mpc_hip_handle *h, const mpc_problem *h_problem, int max_levels, int precision,
                                        mpc_time_fn now, void *now_ctx, double end_time, uint64_t max_nodes,
                                        int split_level, uint32_t shard, uint32_t n_shards,
                                        mpc_tree_level *h_level_best, mpc_result *h_result)
{
    return solve_tree_checked(h, h_problem, max_levels, precision, now, now_ctx, end_time, max_nodes, split_level, shard,
                              n_shards, h_level_best, h_result);
}

/* mpc_hip.h: the plain search, then one level at a time in pieces (this library's shards, one after the other on this
 * handle) for as long as the clock and the levels asked for allow */
extern "C" int mpc_hip_solve_tree_deep(mpc_hip_handle *h, const mpc_problem *h_problem, int max_levels, int precision,
                                       mpc_time_fn now, void *now_ctx, double end_time,
                                       mpc_tree_level *h_level_best, mpc_result *h_result)
{
    if (!h || !h_problem || !h_result) { mpc_set_err("solve_tree_deep: NULL argument"); return MPC_ERR_INVALID; }
    std::vector<mpc_tree_level> lv((size_t)std::max(max_levels, 1));
    /* diagnostic: MPC_HIP_TREE_DEEP_FROM=k lets the plain search end at level k as if level k + 1 did not fit, so that the
     * random runs of tools/fuzz_tree.py reach the pieces with trees the oracle can enumerate */
    static const int deep_from = getenv("MPC_HIP_TREE_DEEP_FROM") ? atoi(getenv("MPC_HIP_TREE_DEEP_FROM")) : 0;
    const bool forced = deep_from >= 2 && max_levels > deep_from;
    int s = solve_tree_checked(h, h_problem, forced ? deep_from : max_levels, precision, now, now_ctx, end_time, 0, 0, 0, 1, lv.data(), h_result);
    if (s != MPC_OK) return s;
    int L = h_result->levels;
    if (forced && L == deep_from) h->tree_stopped_by_width = true;
    if (h_level_best) for (int l = 0; l < L; l++) h_level_best[l] = lv[(size_t)l];
    const uint64_t nsub = (uint64_t)h_problem->n_substeps;
    std::vector<mpc_tree_level> plv((size_t)std::max(max_levels, 1));
    std::vector<mpc_result> keep(2);                          /* [0] the best piece so far, [1] the piece just searched */
    bool wide = h->tree_stopped_by_width;
    while (L < max_levels && wide) {
        if (now && now(now_ctx) >= end_time) break;
        /* Level `target` is searched from the nodes of level target - 2, which must be there in full: a piece takes a
         * contiguous slice of them, gives their children (level target - 1) ghosts' treatment and searches the final level.
         * Nothing above the slice is expanded again (TreeResume).  After the plain search level L - 1 is in full; one level
         * further down it has to be stored first -- in full, no ghosts: 3.0e8 nodes, 10 ms, 32 GB for depth 9. */
        const int target = L + 1, need = target - 2, prec = h->tree_precision;
        if (need < 1) break;
        if (h->tree_full_upto < need) {
            if (h->tree_full_upto < 1) break;
            const TreeResume ext{h->tree_full_upto, lv[(size_t)h->tree_full_upto - 1].best_leaf, true, false, KEY_INF};
            s = solve_tree_checked(h, h_problem, need, prec, now, now_ctx, end_time, 1ull << 30, 0, 0, 1, plv.data(), &keep[1], &ext);
            if (s != MPC_OK || h->tree_full_upto < need) break;                  /* the clock, memory, or a level beyond 32-bit indices */
        }
        /* pieces: so many that a piece's leaves fit 32-bit ordinals and its slice of level target - 1 the storage cap
         * (the next level's size: the last level's growth, at most 30-fold, and half as much again for uneven subtrees; a
         * piece that does not fit after all shows at its own scan, and the level is started over with twice the pieces) */
        const double leaves_L = (double)lv[(size_t)L - 1].n_leaves;
        const double growth = L >= 2 ? std::min(30.0, leaves_L / (double)std::max<uint64_t>(lv[(size_t)L - 2].n_leaves, 1)) : 30.0;
        uint32_t pieces = (uint32_t)std::max(2.0, ceil(std::max(leaves_L * growth * 1.5 / 4.0e9, leaves_L * 1.5 / (double)(1u << 28))));
        TreeResume from{need, lv[(size_t)need - 1].best_leaf, false, true, KEY_INF};     /* (level target - 1's record: lv[], found before) */
        bool done = false, abandoned = false;
        for (int attempt = 0; attempt < 3 && !done && !abandoned; attempt++, pieces *= 2) {
            if (lv[(size_t)need - 1].n_leaves < pieces) { abandoned = true; break; }
            uint64_t total = 0, best_key = ~0ull, best_ord = ~0ull, runner_up = ~0ull;
            uint32_t tie = 0;
            double k_ms = 0, t_ms = 0;
            bool refit = false;
            const double t_level = now ? now(now_ctx) : 0.0;
            for (uint32_t piece = 0; piece < pieces; piece++) {
                if (now) {
                    /* controller.cpp:101-103: an unfinished iteration does not count -- so a level whose remaining pieces, at
                     * the pace of those done, would end after the deadline is given up now rather than at the deadline */
                    const double t = now(now_ctx);
                    if (t >= end_time || (piece > 0 && t + 0.9 * (t - t_level) / piece * (pieces - piece) >= end_time)) { abandoned = true; break; }
                }
                from.beat_key = best_key < KEY_INF ? best_key : KEY_INF;
                s = solve_tree_checked(h, h_problem, target, prec, now, now_ctx, end_time, 0, need, piece, pieces, plv.data(), &keep[1], &from);
                if (s == MPC_ERR_NO_ITERATION && keep[1].levels == target) {         /* nothing in this piece beats what is found: its leaves only count */
                    const mpc_tree_level &e = plv[(size_t)target - 1];
                    uint64_t ek;
                    memcpy(&ek, &e.best_score, sizeof ek);
                    if (e.best_score >= 0.0 && ek < KEY_INF) runner_up = std::min(runner_up, ek);   /* (it may still tie: libm_tie) */
                    total += plv[(size_t)target - 1].n_leaves;
                    continue;
                }
                if (s != MPC_OK) { abandoned = true; break; }
                if (keep[1].levels < target) {                                        /* the clock, or a piece that still does not fit */
                    if (h->tree_stopped_by_width && !(now && now(now_ctx) >= end_time)) refit = true; else abandoned = true;
                    break;
                }
                const mpc_tree_level &t = plv[(size_t)target - 1];
                uint64_t key;
                memcpy(&key, &t.best_score, sizeof key);
                if (!(t.best_score >= 0.0)) key = KEY_INF;
                const uint64_t ord = total + t.best_leaf;
                k_ms += keep[1].kernel_ms; t_ms += keep[1].total_ms;
                if (key < best_key) {                                                 /* pieces come in DFS order: ties keep the first */
                    runner_up = std::min(runner_up, best_key);
                    best_key = key; best_ord = ord; tie = t.libm_tie;
                    keep[0] = keep[1];
                } else {
                    runner_up = std::min(runner_up, key);
                }
                total += t.n_leaves;
            }
            if (abandoned) break;
            if (refit) continue;
            if (best_key >= KEY_INF) { abandoned = true; break; }
            /* libm_tie of the whole level: the winner's own piece says so, or another piece's best lies as near */
            if (runner_up < KEY_INF) {
                double a, b;
                memcpy(&a, &best_key, sizeof a); memcpy(&b, &runner_up, sizeof b);
                if (b <= mpc_tie_limit(a, h_problem->w_distance, h_problem->w_heading, h_problem->w_speed)) tie = 1;
            }
            mpc_tree_level &out = lv[(size_t)target - 1];
            out = plv[(size_t)target - 1];
            out.n_leaves = total; out.best_leaf = best_ord; out.libm_tie = tie; out.reserved = 0;
            out.best_score = keep[0].best_last; out.best_cumulative = keep[0].best_cumulative;
            const uint64_t before = h_result->evaluated, steps_before = h_result->rollout_steps;
            const double k0 = h_result->kernel_ms, t0 = h_result->total_ms;
            *h_result = keep[0];
            h_result->best_ordinal = best_ord;
            h_result->levels = target;
            h_result->libm_tie = tie;
            h_result->evaluated = before + total; h_result->rollout_steps = steps_before + total * nsub;
            h_result->kernel_ms = (float)(k0 + k_ms); h_result->total_ms = (float)(t0 + t_ms);
            if (h_level_best) h_level_best[target - 1] = out;
            L = target;
            done = true;
        }
        if (!done) break;
    }
    return MPC_OK;
}
