/*
 * mpc_kernels.h -- device code of the candidate-set path (included by mpc_hip.hip only).
 *
 *   K1 rollout_kernel   one lane per candidate: D levels x S Euler substeps + tracking cost per level;
 *                       per-problem data (checkpoints, substep schedule, current) staged in LDS; costs written
 *                       coalesced (SoA); optional per-level state snapshots (SoA) so that the winner's
 *                       trajectory needs no second rollout; wavefront-shuffle + LDS argmin -> one
 *                       (key, ordinal, near-tie count) triple per workgroup.
 *   K2 finish_kernel    one workgroup per problem: min over the workgroup triples; (fast mode) every candidate
 *                       whose key is within MPC_REFINE_REL of the minimum goes on a list; a list longer than
 *                       one is re-evaluated with the strict FP64 model and decided by (score, ordinal); the
 *                       winner's trajectory is read from the snapshots (or rolled out again) -> result.
 */
#ifndef MPC_KERNELS_H
#define MPC_KERNELS_H

#include "../../include/mpc_candidates.h"
#include "mpc_internal.h"

using namespace mpc;

struct ControlSource {
    const float *rud, *thr;      /* explicit: [depth][n] */
    uint64_t n;                  /* stride of the explicit arrays */
    uint64_t seed;               /* seeded */
    double last_r, last_t;
    const float *staged;         /* explicit controls of THIS lane staged in LDS by ctl_chunk_store(), or nullptr */
    uint32_t staged_stride;      /* = blockDim.x */
    const double *pregen;        /* seeded controls of THIS lane generated into LDS by the prologue (PREGEN), or nullptr */
};

/* K1 reads every level's controls of its candidates ONCE into LDS ([level][rudder|throttle][lane], FP32), CTL_CHUNK
 * levels (2 * CTL_CHUNK loads) in flight together: one memory latency per kernel for depth <= 32 instead of one per
 * level (a lone wave per SIMD has nothing to hide those behind).  The last chunk repeats the last level rather
 * than branch (same address, same LDS word).  depth is a KERNEL ARGUMENT here, not P.depth, so the loads do not
 * wait for a scalar load of the problem record. */
#define CTL_CHUNK 32
/* One chunk in two halves, so that K1 can put the problem record's loads between them: all of it is one basic block
 * (clamped indices instead of branches; lanes beyond the set load the last candidate's controls into their own LDS
 * words), which is what lets every load be issued before the first wait. */
__device__ inline void ctl_chunk_load(const float *__restrict__ rud, const float *__restrict__ thr, uint64_t n,
                                      uint64_t index, int depth, int base, float (&a)[CTL_CHUNK], float (&b)[CTL_CHUNK])
{
#pragma unroll
    for (int j = 0; j < CTL_CHUNK; j++) {
        const int l = min(base + j, depth - 1);
        a[j] = rud[(uint64_t)l * n + index];
        b[j] = thr[(uint64_t)l * n + index];
    }
}

__device__ inline void ctl_chunk_store(float *ctl, int depth, int base, const float (&a)[CTL_CHUNK],
                                       const float (&b)[CTL_CHUNK])
{
#pragma unroll
    for (int j = 0; j < CTL_CHUNK; j++) {
        const int l = min(base + j, depth - 1);
        ctl[(2 * l) * blockDim.x + threadIdx.x] = a[j];
        ctl[(2 * l + 1) * blockDim.x + threadIdx.x] = b[j];
    }
    /* each lane reads back only what it wrote: no barrier needed */
}

/* STAGED is a compile-time choice on purpose: a run-time choice between the LDS copy and the global arrays makes
 * hipcc select between two pointers and emit flat_ loads, whose vmcnt(0)+lgkmcnt(0) wait also drains every
 * outstanding snapshot store at the top of every level. */
/* PREGEN (small seeded sets: one-wave workgroups): the prologue has generated every level's controls into LDS as
 * doubles ([level][rudder|throttle][lane]), all levels' SplitMix64 chains interleaved, so that the level loop reads two
 * LDS words instead of running a ~40-instruction dependent integer chain alone in its basic block -- on a lone wave
 * that chain cost 7 us per 65,536-candidate solve (round 1).  Launches that fill the SIMDs generate in place. */
template <bool SEEDED, bool STAGED = false, bool PREGEN = false>
__device__ inline void fetch_control(const ControlSource &c, uint64_t ordinal, uint64_t index, int level,
                                     double &r, double &t)
{
    if (SEEDED && PREGEN) {
        r = c.pregen[(2 * level) * c.staged_stride];
        t = c.pregen[(2 * level + 1) * c.staged_stride];
    } else if (SEEDED) {
        mpc_candidate_control(c.seed, ordinal, (uint32_t)level, c.last_r, c.last_t, &r, &t);
    } else if (STAGED) {
        r = (double)c.staged[(2 * level) * c.staged_stride];
        t = (double)c.staged[(2 * level + 1) * c.staged_stride];
    } else {
        r = (double)c.rud[(uint64_t)level * c.n + index];
        t = (double)c.thr[(uint64_t)level * c.n + index];
    }
}

__device__ inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Per-level state snapshots, pair-SoA: K1 writes the state after every level as 16-byte pairs (one
 * global_store_dwordx4 per pair, a wave's 64 lanes = 1 KiB contiguous), pair q of level l of candidate i at
 * base[((l*SNAP_PAIRS + q)*stride + i)*2 .. +1]; field f lives in pair f/2, slot f%2.
 *   strict: x y | h v | rpm cmg | sog cum | last -      fast: x y | sh ch | v rpm | cum last   (x, y through the water)
 * Passed BY VALUE: a pointer to it would put the record in scratch memory and turn every store into a flat_ store
 * that the LDS waits then have to drain.  base == nullptr = no snapshots. */
#define SNAP_PAIRS 5
#define SNAP_FIELDS (2 * SNAP_PAIRS)
/* Only every 2^shift-th level is written (the states after levels S-1, 2S-1, ...: slot (l+1)/S - 1): K2's thread for
 * level l starts from the nearest slot below (or the start state) and integrates the <= S-1 levels in between with the
 * code K1 ran -- the same bits -- all levels in parallel.  The host picks the stride per run: every level (shift 0) for
 * a run whose result is wanted at once -- K2 then only reads (6 us instead of 12 on config 2), and nobody counts the 84
 * MB K1 writes in 39 us; every 4th (shift 2, 21 MB: the problem's own bytes are 12 MB) for a stream of solves, whose
 * finish rides under the next rollout launch where its extra 6 us cost nothing, and for sets too big for every level. */
#ifndef SNAP_SHIFT_STREAM
#define SNAP_SHIFT_STREAM 2
#endif
__host__ __device__ inline int snap_slots(int depth, int shift) { return depth >> shift; }
/* K1's side.  The stores are buffer stores with the WRITE-THROUGH cache policy (sc0 sc1): 94 MB of snapshots per
 * config-2 solve otherwise sit dirty in the XCDs' L2s until the end-of-kernel write-back, which then delays the
 * next dispatch (measured: plain 42.1 us K1 / 58.7 us per solve, nontemporal 40.0 / 56.3, write-through 39.6 / 55.6;
 * no snapshots at all 38.5).  The descriptor also moves the level/pair part of the address into a scalar offset:
 * no 64-bit vector address arithmetic per store. */
typedef uint32_t mpc_u32x4 __attribute__((ext_vector_type(4)));
#define MPC_AUX_WRITE_THROUGH 17          /* sc0 (1) | sc1 (16) */
struct SnapW {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t pair_bytes;         /* 16 * candidates of the problem: distance between consecutive pairs */
    uint32_t lane_off;           /* 16 * this lane's candidate index */
    uint32_t mask;               /* 2^shift - 1: level l is written when ((l + 1) & mask) == 0, into slot ((l + 1) >> shift) - 1 */
    int shift;
    bool on;
    __device__ inline void put2(int l, int q, double a, double b) const
    {
        union { double d[2]; mpc_u32x4 v; } u;
        u.d[0] = a; u.d[1] = b;
        __builtin_amdgcn_raw_buffer_store_b128(u.v, rsrc, (int)lane_off, (int)((uint32_t)(l * SNAP_PAIRS + q) * pair_bytes),
                                               MPC_AUX_WRITE_THROUGH);
    }
};

__device__ inline SnapW snap_writer(double *problem_base, uint64_t count, int snap_depth, int shift, uint64_t i)
{
    SnapW w;
    const uint64_t bytes = (uint64_t)snap_slots(snap_depth, shift) * SNAP_PAIRS * count * 16;      /* <= MPC_SNAPSHOT_BYTES_MAX (host) */
    w.shift = shift; w.mask = (1u << shift) - 1u;
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc(problem_base, 0, (int)(uint32_t)bytes, 0x00020000);
    w.pair_bytes = (uint32_t)(count * 16);
    w.lane_off = (uint32_t)(i * 16);
    w.on = problem_base != nullptr && bytes != 0;
    return w;
}

__device__ inline SnapW snap_none()
{
    SnapW w;
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
    w.pair_bytes = 0; w.lane_off = 0; w.mask = 0; w.shift = 0; w.on = false;
    return w;
}

/* K2's side: plain loads of the winner's pairs */
struct Snap {
    const double *base; uint64_t stride, i;
    __device__ inline double get(int l, int f) const
    {
        return base[(((uint64_t)l * SNAP_PAIRS + (f >> 1)) * stride + i) * 2 + (f & 1)];
    }
};

/* Both evaluators: UNIFORM = every substep of a level has the same length (BASELINE configs 2-5: 20 x 0.05 s), so
 * the schedule row lives in registers instead of being re-read from LDS 400 times per candidate. */

/* one level and its score, the strict model: the ONE spelling K1 (eval_strict) and K2 (re-integration between
 * snapshot slots) share, so that both produce the same bits */
template <bool UNIFORM>
__device__ inline double strict_level_and_score(const DevProblem &P, const Model &M, const Weights &w, const double *cpL,
                                                const double *subL, const double *row0, int l, double r, double t, Veh &s)
{
    if (UNIFORM) strict_level<0>(s, r, t, row0, P.nsub, P.cx, P.cy, M);
    else strict_level<SUB_STRIDE>(s, r, t, subL, P.nsub, P.cx, P.cy, M);
    return strict_score(cpL + l * CP_STRIDE, s, w);
}

/* strict evaluation of one candidate; traj (optional) receives depth+1 level states */
template <bool SEEDED, bool UNIFORM = false, bool STAGED = false, bool PREGEN = false>
__device__ inline void eval_strict(const DevProblem &P, const double *cpL, const double *subL,
                                   const ControlSource &c, uint64_t ordinal, uint64_t index,
                                   double &last, double &cum, mpc_level_state *traj, const SnapW snap)
{
    Veh s{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0};
    const Weights w{P.wd, P.wh, P.ws};
    const Model M = P.model;                  /* wave-uniform: lives in scalar registers for the whole rollout */
    const double row0[1] = {subL[0]};
    last = 0.0; cum = 0.0;
    if (traj) traj[0] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
    for (int l = 0; l < P.depth; l++) {
        double r, t;
        fetch_control<SEEDED, STAGED, PREGEN>(c, ordinal, index, l, r, t);
        last = strict_level_and_score<UNIFORM>(P, M, w, cpL, subL, row0, l, r, t, s);
        cum += last;
        if (traj)
            traj[l + 1] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, s.cmg, s.sog,
                                          clampd(r, -1, 1), clampd(t, 0, 1), cum};
        if (snap.on && ((uint32_t)(l + 1) & snap.mask) == 0) {
            const int q = ((l + 1) >> snap.shift) - 1;
            snap.put2(q, 0, s.x, s.y); snap.put2(q, 1, s.h, s.v); snap.put2(q, 2, s.rpm, s.cmg);
            snap.put2(q, 3, s.sog, cum); snap.put2(q, 4, last, 0.0);
        }
    }
}

/* fast state at the end of level l -> the public level state */
__device__ inline mpc_level_state fast_to_level_state(const double *cp, const double *q_last, double x, double y,
                                                      double sh, double ch, double v, double rpm, double delta,
                                                      double r, double t, double cum)
{
    double h = atan2(sh, ch);
    if (h < 0) h += TWO_PI;
    const double gdx = fma(delta, sh, q_last[3]), gdy = fma(delta, ch, q_last[4]);
    const double sog = sqrt(gdx * gdx + gdy * gdy) * q_last[5];
    /* cp[0] - cp[6] = cx * T_l: the current's share of the displacement, common to all candidates */
    return mpc_level_state{x + (cp[0] - cp[6]), y + (cp[1] - cp[7]), h, v, rpm, PI_2 - atan2(gdy, gdx), sog,
                           clampd(r, -1, 1), clampd(t, 0, 1), cum};
}

/* one level and its score, the fast model: shared by K1 and K2 like strict_level_and_score.  The checkpoint's words are
 * read before the substep loop (a lone wave has nothing to hide their LDS latency behind). */
template <bool UNIFORM>
__device__ inline double fast_level_and_score(const DevProblem &P, const Model &M, const Weights &w, const double *cpL,
                                              const double *subL, const double *row0, const double *q_last, int l,
                                              double r, double t, FastVeh &s)
{
    const double *cpv = cpL + l * CP_STRIDE;
    double cp[CP_STRIDE];
    cp[3] = cpv[3]; cp[4] = cpv[4]; cp[5] = cpv[5]; cp[6] = cpv[6]; cp[7] = cpv[7];
    /* ONE spelling of the fast model everywhere (K1, K2, the pair kernel, the single rollout; the tree expansion uses the
     * same one): a candidate's cost must not depend on how its set was supplied, sized or scheduled.  It is the LEAN one
     * (mpc_model.h: three-address Horner steps, the 17-instruction exp) since a stream of solves runs two waves per SIMD
     * and is VALU-bound like the dense launches: +2 % on configs 2, 4 and 5 alike (round 1's lone-wave kernel was 0.6 us
     * FASTER with the compiler's spelling, whose copies filled dependent-issue bubbles).  Costs moved by <= 9e-15
     * relative, no winner changed. */
#ifndef MPC_FAST_LEAN
#define MPC_FAST_LEAN true
#endif
    if (UNIFORM) fast_level<0, MPC_FAST_LEAN>(s, r, t, row0, P.nsub, M);
    else if (P.head_uniform) fast_level_head_uniform<MPC_FAST_LEAN>(s, r, t, subL, P.nsub, M, P.head_uniform == 2);
    else fast_level<SUB_STRIDE, MPC_FAST_LEAN>(s, r, t, subL, P.nsub, M);
    return fast_score<MPC_FAST_LEAN>(cp, s, w, q_last);
}

template <bool SEEDED, bool UNIFORM = false, bool STAGED = false, bool PREGEN = false>
__device__ inline void eval_fast(const DevProblem &P, const double *cpL, const double *subL,
                                 const ControlSource &c, uint64_t ordinal, uint64_t index,
                                 double &last, double &cum, mpc_level_state *traj, const SnapW snap)
{
    FastVeh s;
    s.x = P.sx; s.y = P.sy; s.v = P.sv; s.rpm = P.srpm; s.delta = 0;
    sincos(P.sh, &s.sh, &s.ch);
    const Weights w{P.wd, P.wh, P.ws};
    const Model M = P.model;                  /* wave-uniform: scalar registers */
    const double *q_last = subL + (P.nsub - 1) * SUB_STRIDE;
    const double row0[6] = {subL[0], subL[1], subL[2], subL[3], subL[4], subL[5]};
    last = 0.0; cum = 0.0;
    if (traj) traj[0] = mpc_level_state{s.x, s.y, P.sh, s.v, s.rpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
    /* A lone wave per SIMD has nothing to hide a latency behind, so what a level reads from LDS is requested one
     * substep loop ahead: level l+1's controls and level l's checkpoint are in flight while level l integrates. */
    double r_next = 0, t_next = 0;
    if (P.depth > 0) fetch_control<SEEDED, STAGED, PREGEN>(c, ordinal, index, 0, r_next, t_next);
    for (int l = 0; l < P.depth; l++) {
        const double r = r_next, t = t_next;
        fetch_control<SEEDED, STAGED, PREGEN>(c, ordinal, index, min(l + 1, P.depth - 1), r_next, t_next);
        last = fast_level_and_score<UNIFORM>(P, M, w, cpL, subL, row0, q_last, l, r, t, s);
        cum += last;
        if (traj)
            traj[l + 1] = fast_to_level_state(cpL + l * CP_STRIDE, q_last, s.x, s.y, s.sh, s.ch, s.v, s.rpm, s.delta,
                                              r, t, cum);
        if (snap.on && ((uint32_t)(l + 1) & snap.mask) == 0) {
            const int q = ((l + 1) >> snap.shift) - 1;
            snap.put2(q, 0, s.x, s.y); snap.put2(q, 1, s.sh, s.ch); snap.put2(q, 2, s.v, s.rpm);
            snap.put2(q, 3, cum, last);
        }
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* K2: per problem -- reduce, refine list, strict decision among near-ties, the winner's trajectory        */
/* ------------------------------------------------------------------------------------------------------ */
#define FINISH_LDS_WORDS 16          /* uint64 words of small shared state: red[8], s_win[2], s_n */

/* p = problem; lds = cpL (subL follows, as stage_problem_store lays them out); s_list[LIST_CAP]; small[FINISH_LDS_WORDS].
 * Works for any workgroup size that is a multiple of 64 and at least MPC_MAX_DEPTH lanes. */
template <bool SEEDED, uint32_t FIN_U>
__device__ inline void finish_body(const FinishArgs &A, uint32_t p, double *lds, uint64_t *s_list, uint64_t *small)
{
    const DevProblem *__restrict__ probs = A.probs;
    const float *__restrict__ rud = A.rud, *__restrict__ thr = A.thr;
    const uint64_t first = A.first, count = A.count;
    const int selection = A.selection, fast = A.fast, uniform = A.uniform;
    const double refine_rel = A.refine_rel;
    const uint32_t nblk = A.nblk, k1_block = A.k1_block;
    const uint64_t *__restrict__ pkey = A.pkey, *__restrict__ pord = A.pord;
    const uint32_t *__restrict__ pcnt = A.pcnt;
    const double *__restrict__ keys = A.keys;
    double *__restrict__ snap_base = A.snap_base;
    const int snap_depth = A.snap_depth, snap_shift = A.snap_shift;
    uint64_t *__restrict__ winners = A.winners, *__restrict__ sink = A.sink;
    const uint32_t sw = A.sink_words;              /* 2: (score bits, ordinal); 2 + 2 * depth: + the winner's controls */
    DevResult *__restrict__ results = A.results;
    uint64_t *red = small, *s_win = small + 8;
    uint32_t &s_n = *(uint32_t *)(small + 10);
    const DevProblem &P = probs[p];
    double *cpL = lds, *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    if (threadIdx.x == 0) s_n = 0;
    const ControlSource c{rud, thr, count, P.seed, P.last_r, P.last_t, nullptr, 0, nullptr};
    DevResult &R = results[p];
    const uint64_t *pk = pkey + (uint64_t)p * nblk, *po = pord + (uint64_t)p * nblk;
    const uint32_t *pc = pcnt + (uint64_t)p * nblk;

    /* 1. the workgroup triples' minimum (lowest ordinal among equal keys).  K2 is a chain of memory latencies, so
     *    every trip keeps FIN_U triples per thread in flight (indices clamped, not predicated: a repeated triple
     *    does not change a minimum), and the first trip's loads travel together with the problem record's. */
    uint64_t kk[FIN_U], oo[FIN_U];
    uint32_t cc[FIN_U];
#pragma unroll
    for (uint32_t u = 0; u < FIN_U; u++) {
        const uint32_t j = min(u * blockDim.x + threadIdx.x, nblk - 1);
        kk[u] = pk[j]; oo[u] = po[j]; cc[u] = pc[j];
    }
    StagedProblem staged;
    stage_problem_load(P, staged);
    stage_problem_store(staged, cpL);
    uint64_t k = ~0ull, o = ~0ull;
#pragma unroll
    for (uint32_t u = 0; u < FIN_U; u++) lex_min(k, o, kk[u], oo[u]);
    for (uint32_t base = FIN_U * blockDim.x; base < nblk; base += FIN_U * blockDim.x) {     /* > 1024 workgroups */
        uint64_t k2[FIN_U], o2[FIN_U];
#pragma unroll
        for (uint32_t u = 0; u < FIN_U; u++) {
            const uint32_t j = min(base + u * blockDim.x + threadIdx.x, nblk - 1);
            k2[u] = pk[j]; o2[u] = po[j];
        }
#pragma unroll
        for (uint32_t u = 0; u < FIN_U; u++) lex_min(k, o, k2[u], o2[u]);
    }
    block_argmin(k, o, red);
    if (threadIdx.x == 0) { s_win[0] = k; s_win[1] = o; }
    __syncthreads();
    k = s_win[0]; o = s_win[1];
    /* no finite score anywhere: nothing is ever below the reference's initial minScore = DBL_MAX
     * (controller.cpp:81,160), so nothing is selected */
    uint64_t win = k < KEY_INF ? o : ~0ull;
    uint32_t n = 0, overflow = 0, tie = 0;
    /* strict runs: is there a second candidate within MPC_TIE_REL of the winner (mpc_result.libm_tie)?  In another
     * workgroup its triple shows it; inside the winner's own, K1's near count does (strict runs count against that window). */
    if (!fast && k < KEY_INF) {
        int t = 0;
#pragma unroll
        for (uint32_t u = 0; u < FIN_U; u++) {
            const uint32_t j = u * blockDim.x + threadIdx.x;
            if (j < nblk) t |= keys_tie(k, kk[u], P.wd, P.wh, P.ws) && (oo[u] != o || cc[u] > 1);
        }
        for (uint32_t j = FIN_U * blockDim.x + threadIdx.x; j < nblk; j += blockDim.x) t |= keys_tie(k, pk[j], P.wd, P.wh, P.ws) && (po[j] != o || pc[j] > 1);
        tie = __syncthreads_or(t) != 0;
    }

    /* 2. fast mode: the candidates within refine_rel of the minimum */
    if (fast && k < KEY_INF) {
        const double m = __longlong_as_double((long long)k);
        const double limit = mpc_refine_limit(m, refine_rel, P.wd, P.wh, P.ws);
        auto consider = [&](uint32_t j, uint64_t key, uint64_t ord, uint32_t near) {
            if (__longlong_as_double((long long)key) > limit) return;
            if (near <= 1) {
                const uint32_t at = atomicAdd(&s_n, 1u);
                if (at < LIST_CAP) s_list[at] = ord;
            } else {                                       /* several near-ties inside one K1 workgroup */
                const uint64_t lo = (uint64_t)j * k1_block, hi = min(lo + k1_block, count);
                for (uint64_t e = lo; e < hi; e++)
                    if (keys[(uint64_t)p * count + e] <= limit) {
                        const uint32_t at = atomicAdd(&s_n, 1u);
                        if (at < LIST_CAP) s_list[at] = first + e;
                    }
            }
        };
#pragma unroll
        for (uint32_t u = 0; u < FIN_U; u++) {               /* the first trip's triples are still in registers */
            const uint32_t j = u * blockDim.x + threadIdx.x;
            if (j < nblk) consider(j, kk[u], oo[u], cc[u]);
        }
        for (uint32_t j = FIN_U * blockDim.x + threadIdx.x; j < nblk; j += blockDim.x) consider(j, pk[j], po[j], pc[j]);
        __syncthreads();
        n = s_n;
        if (n > LIST_CAP) { overflow = 1; n = LIST_CAP; }
        /* 3. more than one: the strict model decides, lowest ordinal among equal scores */
        if (n > 1) {
            uint64_t k2 = ~0ull, o2 = ~0ull, second = ~0ull;       /* this thread's best and the key of its next best */
            for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
                const uint64_t ord = s_list[e];
                double last, cum;
                eval_strict<SEEDED>(P, cpL, subL, c, ord, ord - first, last, cum, nullptr, snap_none());
                const uint64_t ke = key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last);
                if (ke < k2 || (ke == k2 && ord < o2)) { second = k2; k2 = ke; o2 = ord; }
                else if (ke < second) second = ke;
            }
            const uint64_t my_k = k2, my_o = o2;
            block_argmin(k2, o2, red);
            if (threadIdx.x == 0) { s_win[0] = k2; s_win[1] = o2; }
            __syncthreads();
            win = s_win[1];
            /* the strict runner-up: every thread offers its best, the winner's holder its next best */
            uint64_t ru = (my_o == win) ? second : my_k, zero = 0;
            const uint64_t wk = s_win[0];
            block_argmin(ru, zero, red);
            if (threadIdx.x == 0) s_win[0] = ru;
            __syncthreads();
            tie = keys_tie(wk, s_win[0], P.wd, P.wh, P.ws);
        } else if (n == 1) {
            win = s_list[0];
        }
    }

    /* 4. the winner's trajectory.  A fast-mode winner that had no rival within refine_rel keeps the fast
     *    model's states and scores (contract 1e-5 relative); otherwise everything returned is strict. */
    const bool keep_fast = fast && n <= 1;
    if (win == ~0ull) {
        if (threadIdx.x == 0) {
            R.key_bits = KEY_INF; R.ordinal = win; R.best_last = 0; R.best_cum = 0; R.refined = 0; R.overflow = overflow; R.tie = 0;
            winners[2 * p] = KEY_INF; winners[2 * p + 1] = win;
            if (sink) { sink[(size_t)p * sw] = KEY_INF; sink[(size_t)p * sw + 1] = win; }
        }
        return;
    }
    if (snap_base && snap_slots(snap_depth, snap_shift) > 0 && (keep_fast || !fast)) {
        /* K1 left every candidate's state after every 2^shift-th level in HBM.  One thread per level: start from the
         * nearest slot below (or the start state) and integrate the < 2^shift levels in between with K1's own
         * code (strict/fast_level_and_score, the flavour K1 ran): the same bits as K1 had in its registers. */
        const Snap snap{snap_base + (uint64_t)p * snap_slots(snap_depth, snap_shift) * SNAP_FIELDS * count, count, win - first};
        const int l = threadIdx.x;
        double *s_tail = (double *)(small + 12);       /* [0] = last, [1] = cum of the last level (words 12, 13 of the small state) */
        if (l < P.depth) {
            const Weights w{P.wd, P.wh, P.ws};
            const Model M = P.model;
            const int g = (l + 1) >> snap_shift;           /* slots at or below level l */
            const int from = g << snap_shift;              /* first level to integrate (== l + 1: none) */
            double r = 0, t = 0, cum = 0, last = 0;
            if (fast) {
                const double *q_last = subL + (P.nsub - 1) * SUB_STRIDE;
                const double row0[6] = {subL[0], subL[1], subL[2], subL[3], subL[4], subL[5]};
                FastVeh s;
                if (g == 0) {
                    s.x = P.sx; s.y = P.sy; s.v = P.sv; s.rpm = P.srpm; s.delta = 0;
                    sincos(P.sh, &s.sh, &s.ch);
                } else {
                    const int q = g - 1;
                    s.x = snap.get(q, 0); s.y = snap.get(q, 1); s.sh = snap.get(q, 2); s.ch = snap.get(q, 3);
                    s.v = snap.get(q, 4); s.rpm = snap.get(q, 5); cum = snap.get(q, 6); last = snap.get(q, 7);
                    s.delta = s.v * q_last[0];             /* v * (last substep's dt), as fast_level leaves it */
                }
                for (int j = from; j <= l; j++) {
                    fetch_control<SEEDED>(c, win, win - first, j, r, t);
                    last = uniform ? fast_level_and_score<true>(P, M, w, cpL, subL, row0, q_last, j, r, t, s)
                                   : fast_level_and_score<false>(P, M, w, cpL, subL, row0, q_last, j, r, t, s);
                    cum += last;
                }
                if (from > l) fetch_control<SEEDED>(c, win, win - first, l, r, t);
                R.traj[l + 1] = fast_to_level_state(cpL + l * CP_STRIDE, q_last, s.x, s.y, s.sh, s.ch, s.v, s.rpm, s.delta, r, t, cum);
            } else {
                const double row0[1] = {subL[0]};
                Veh s{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0};
                if (g > 0) {
                    const int q = g - 1;
                    s = Veh{snap.get(q, 0), snap.get(q, 1), snap.get(q, 2), snap.get(q, 3), snap.get(q, 4), snap.get(q, 5), snap.get(q, 6)};
                    cum = snap.get(q, 7); last = snap.get(q, 8);
                }
                for (int j = from; j <= l; j++) {
                    fetch_control<SEEDED>(c, win, win - first, j, r, t);
                    last = uniform ? strict_level_and_score<true>(P, M, w, cpL, subL, row0, j, r, t, s)
                                   : strict_level_and_score<false>(P, M, w, cpL, subL, row0, j, r, t, s);
                    cum += last;
                }
                if (from > l) fetch_control<SEEDED>(c, win, win - first, l, r, t);
                R.traj[l + 1] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, s.cmg, s.sog, clampd(r, -1, 1), clampd(t, 0, 1), cum};
            }
            if (l == P.depth - 1) { s_tail[0] = last; s_tail[1] = cum; }
            if (sink && sw > 2) {                  /* the winner's controls travel with its record (mpc_hip_combine_shards_exact) */
                sink[(size_t)p * sw + 2 + 2 * l] = (uint64_t)__double_as_longlong(clampd(r, -1, 1));
                sink[(size_t)p * sw + 3 + 2 * l] = (uint64_t)__double_as_longlong(clampd(t, 0, 1));
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double last = s_tail[0], cum = s_tail[1];
            const uint64_t kb = key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last);
            R.traj[0] = mpc_level_state{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
            R.key_bits = kb; R.ordinal = win; R.best_last = last; R.best_cum = cum;
            R.refined = 0; R.overflow = overflow; R.tie = tie;
            winners[2 * p] = kb; winners[2 * p + 1] = win;
            if (sink) { sink[(size_t)p * sw] = kb; sink[(size_t)p * sw + 1] = win; }
        }
    } else if (threadIdx.x == 0) {
        double last = 0, cum = 0;
        /* the same flavour of the fast model as K1 ran (the two differ in rounding), so the result is K1's own */
        if (keep_fast && uniform) eval_fast<SEEDED, true>(P, cpL, subL, c, win, win - first, last, cum, R.traj, snap_none());
        else if (keep_fast) eval_fast<SEEDED, false>(P, cpL, subL, c, win, win - first, last, cum, R.traj, snap_none());
        else eval_strict<SEEDED>(P, cpL, subL, c, win, win - first, last, cum, R.traj, snap_none());
        const uint64_t kb = key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last);
        R.key_bits = kb; R.ordinal = win; R.best_last = last; R.best_cum = cum;
        R.refined = n > 1 ? n : 0; R.overflow = overflow; R.tie = tie;
        winners[2 * p] = kb; winners[2 * p + 1] = win;
        if (sink) {
            sink[(size_t)p * sw] = kb; sink[(size_t)p * sw + 1] = win;
            if (sw > 2)
                for (int l = 0; l < P.depth; l++) {
                    sink[(size_t)p * sw + 2 + 2 * l] = (uint64_t)__double_as_longlong(R.traj[l + 1].rudder);
                    sink[(size_t)p * sw + 3 + 2 * l] = (uint64_t)__double_as_longlong(R.traj[l + 1].throttle);
                }
        }
    }
}

template <bool SEEDED>
__global__ __launch_bounds__(256) void finish_kernel(const FinishArgs A)
{
    __shared__ __attribute__((aligned(16))) double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    __shared__ uint64_t s_list[LIST_CAP];
    __shared__ uint64_t small[FINISH_LDS_WORDS];
    finish_body<SEEDED, 4>(A, blockIdx.x, lds, s_list, small);
}

/* the finishes of the (<= MPC_GROUP_MAX) runs of one launch (mpc_hip_set_pipeline level >= 2), when nothing follows to
 * carry them: blockIdx.y picks the run */
struct FinishGroup { FinishArgs f[MPC_GROUP_MAX]; };
template <bool SEEDED>
__global__ __launch_bounds__(256) void finish_group_kernel(const FinishGroup G)
{
    __shared__ __attribute__((aligned(16))) double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    __shared__ uint64_t s_list[LIST_CAP];
    __shared__ uint64_t small[FINISH_LDS_WORDS];
    finish_body<SEEDED, 4>(G.f[blockIdx.y], blockIdx.x, lds, s_list, small);
}

/* ------------------------------------------------------------------------------------------------------ */
/* K1: rollout + cost + block argmin                                                                       */
/* ------------------------------------------------------------------------------------------------------ */
/* FUSE: the launch carries one more workgroup per problem (the last of each grid row) that is not a rollout
 * workgroup but K2 of the PREVIOUS solve -- its arguments in `prev`, its data in the other half of the handle's
 * double-buffered working memory.  That finish is a chain of memory round trips on one workgroup (~6 us as a kernel of
 * its own, plus a dispatch boundary); here it runs beside a thousand rollout workgroups that do not wait for it, so a
 * stream of solves costs ONE dispatch each.  (Round 1 tried the other fusion -- the LAST-arriving workgroup of solve n
 * finishes solve n -- which serialises the same chain behind the rollout: slower than two kernels.)  Its LDS comes out
 * of the dynamic allocation: the problem staging area it shares by layout, the near-tie list sits where a rollout
 * workgroup keeps its controls (the host sizes the allocation for both). */
/* one rollout workgroup: workgroup `bx` of the `n_rollout_blocks` of its run (problem blockIdx.y) */
template <int PREC, bool SEEDED, bool UNIFORM, bool PREGEN>
__device__ __forceinline__ void rollout_body(const DevProblem *__restrict__ probs,
                                             const float *__restrict__ rud, const float *__restrict__ thr,
                                             uint64_t first, uint64_t count, int selection, double refine_rel,
                                             double *__restrict__ cost_last, double *__restrict__ cost_cum,
                                             double *__restrict__ keys, double *__restrict__ snap_base,
                                             int snap_depth, int snap_shift, uint64_t *__restrict__ pkey,
                                             uint64_t *__restrict__ pord, uint32_t *__restrict__ pcnt,
                                             const uint32_t bx, const uint32_t n_rollout_blocks, double *lds)
{
    double *cpL = lds;
    double *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    uint64_t *red = (uint64_t *)(subL + MPC_MAX_SUBSTEPS * SUB_STRIDE);       /* 8 words argmin + 2 words (+ 2 pad) */
    float *ctlL = (float *)(red + 12);                                         /* [depth][2][blockDim] (explicit sets) */
    const DevProblem &P = probs[blockIdx.y];
    const uint64_t i = (uint64_t)bx * blockDim.x + threadIdx.x;
    /* prologue: every global load of the workgroup is issued before the first wait (problem record, then controls;
     * explicit sets hold one problem, whose depth is snap_depth) */
    StagedProblem staged;
    float ca[CTL_CHUNK], cb[CTL_CHUNK];
    const uint64_t ic = i < count ? i : count - 1;          /* count >= 1 (host) */
    stage_problem_load(P, staged);
    if (!SEEDED) ctl_chunk_load(rud, thr, count, ic, snap_depth, 0, ca, cb);            /* snap_depth >= 1 (host) */
    /* all global loads first, then the LDS writes (left alone, the scheduler interleaves them and the first wait
     * comes with half of the loads still to be issued): 0x020 = VMEM reads, 0x200 = DS writes */
    __builtin_amdgcn_sched_group_barrier(0x020, STAGE_TRIPS + 2 * CTL_CHUNK, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, STAGE_TRIPS + 2 * CTL_CHUNK, 0);
    stage_problem_store(staged, cpL);
    if (!SEEDED) {
        ctl_chunk_store(ctlL, snap_depth, 0, ca, cb);
        for (int base = CTL_CHUNK; base < snap_depth; base += CTL_CHUNK) {       /* deeper than one chunk: a round trip each */
            ctl_chunk_load(rud, thr, count, ic, snap_depth, base, ca, cb);
            ctl_chunk_store(ctlL, snap_depth, base, ca, cb);
        }
    }
    double *genL = (double *)ctlL;                                             /* [depth][2][blockDim] doubles (PREGEN) */
    if (SEEDED && PREGEN) {
        /* every level's controls of this lane, the levels' hash chains independent of one another (unrolled x4: a lone
         * wave issues one chain's next instruction while another's result is in flight); each lane reads back only
         * what it wrote: no barrier */
        const uint64_t ord = first + ic;
        const int depth = P.depth;
#pragma unroll 4
        for (int l = 0; l < depth; l++) {
            double r, t;
            mpc_candidate_control(P.seed, ord, (uint32_t)l, P.last_r, P.last_t, &r, &t);
            genL[(2 * l) * blockDim.x + threadIdx.x] = r;
            genL[(2 * l + 1) * blockDim.x + threadIdx.x] = t;
        }
    }
    uint64_t k = ~0ull, o = ~0ull;
    double mykey = __longlong_as_double((long long)KEY_INF);
    if (i < count) {
        const ControlSource c{rud, thr, count, P.seed, P.last_r, P.last_t,
                              SEEDED ? nullptr : ctlL + threadIdx.x, blockDim.x,
                              (SEEDED && PREGEN) ? genL + threadIdx.x : nullptr};
        const SnapW snap = snap_writer(snap_base ? snap_base + (uint64_t)blockIdx.y * snap_slots(snap_depth, snap_shift) * SNAP_FIELDS * count : nullptr,
                                       count, snap_depth, snap_shift, i);
        double last, cum;
        if (PREC == MPC_PREC_F64) eval_strict<SEEDED, UNIFORM, !SEEDED, PREGEN>(P, cpL, subL, c, first + i, i, last, cum, nullptr, snap);
        else eval_fast<SEEDED, UNIFORM, !SEEDED, PREGEN>(P, cpL, subL, c, first + i, i, last, cum, nullptr, snap);
        const uint64_t at = (uint64_t)blockIdx.y * count + i;
        if (cost_last) cost_last[at] = last;
        if (cost_cum) cost_cum[at] = cum;
        k = key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last);
        o = first + i;
        mykey = __longlong_as_double((long long)k);
        keys[at] = mykey;
    }
    block_argmin(k, o, red);
    /* how many of this workgroup's candidates lie within refine_rel of ITS minimum (a superset test for the
     * global one): 1 in all but freak cases, which lets K2 build the refine list from the triples alone */
    uint32_t *cnt = (uint32_t *)(red + 9);
    if (threadIdx.x == 0) { red[8] = k; *cnt = 0; }
    __syncthreads();
    const double m = __longlong_as_double((long long)red[8]);
    /* strict runs have no refine window; they count the candidates within MPC_TIE_REL of the workgroup's minimum instead,
     * from which K2 sets mpc_result.libm_tie */
    const double near_limit = mpc_refine_limit(m, PREC == MPC_PREC_F64 ? MPC_TIE_REL : refine_rel, P.wd, P.wh, P.ws);
    const bool near = (i < count) && (mykey <= near_limit);
    const uint32_t wave_n = (uint32_t)__popcll(__ballot(near));
    if ((threadIdx.x & 63) == 0 && wave_n) atomicAdd(cnt, wave_n);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t b = (uint64_t)blockIdx.y * n_rollout_blocks + bx;
        pkey[b] = k; pord[b] = o; pcnt[b] = *cnt;
    }
}

template <int PREC, bool SEEDED, bool UNIFORM, bool FUSE = false, bool PREGEN = false>
__global__ __launch_bounds__(256) void rollout_kernel(const DevProblem *__restrict__ probs,
                                                      const float *__restrict__ rud, const float *__restrict__ thr,
                                                      uint64_t first, uint64_t count, int selection, double refine_rel,
                                                      double *__restrict__ cost_last, double *__restrict__ cost_cum,
                                                      double *__restrict__ keys, double *__restrict__ snap_base,
                                                      int snap_depth, int snap_shift, uint64_t *__restrict__ pkey,
                                                      uint64_t *__restrict__ pord, uint32_t *__restrict__ pcnt,
                                                      const FinishArgs prev)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const uint32_t n_rollout_blocks = FUSE ? gridDim.x - 1 : gridDim.x;
    if (FUSE && blockIdx.x == n_rollout_blocks) {                              /* uniform per workgroup */
        /* the near-tie list sits where a rollout workgroup keeps its controls; FINISH_LDS_WORDS (16) > the 12 argmin
         * words before them: the small state goes behind the list */
        uint64_t *s_list = (uint64_t *)(lds + MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE) + 12;
        finish_body<SEEDED, 16>(prev, blockIdx.y, lds, s_list, s_list + LIST_CAP);
        return;
    }
    rollout_body<PREC, SEEDED, UNIFORM, PREGEN>(probs, rud, thr, first, count, selection, refine_rel, cost_last, cost_cum, keys,
                                                snap_base, snap_depth, snap_shift, pkey, pord, pcnt, blockIdx.x, n_rollout_blocks, lds);
}

/* Several runs in ONE launch (mpc_hip_set_pipeline level 2..MPC_GROUP_MAX).  A 65,536-candidate set is one wave per
 * SIMD, and a lone wave issues a VALU instruction every ~6.2 cycles instead of every 4: the second wave per SIMD can only
 * come from another solve -- and every launch of a stream pays ~10 us that do not depend on the work (the boundary between
 * dependent dispatches, the workgroups' prologue, the tail), which a launch of four solves pays once instead of twice.
 * The grid is run 0's workgroups, then run 1's, ... (end[w] = one past run w's last workgroup; an unused run has none),
 * then one finish workgroup per PENDING run (the previous launch's runs) -- fast uniform rollouts in one-wave workgroups
 * only.  The run's arguments are picked by a workgroup-uniform index into the kernel argument block (scalar loads, one
 * copy of the body). */
struct GroupArgs { RunArgs run[MPC_GROUP_MAX]; FinishArgs prev[MPC_GROUP_MAX]; uint32_t end[MPC_GROUP_MAX]; };

template <bool SEEDED>
__global__ __launch_bounds__(256) void rollout_group_kernel(const DevProblem *__restrict__ probs, const GroupArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const uint32_t nr = A.end[MPC_GROUP_MAX - 1];
    if (blockIdx.x >= nr) {
        /* (the finish workgroups FIRST in the grid instead: 1 % slower -- measured, round 3) */
        uint64_t *s_list = (uint64_t *)(lds + MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE) + 12;
        finish_body<SEEDED, 16>(A.prev[blockIdx.x - nr], blockIdx.y, lds, s_list, s_list + LIST_CAP);
        return;
    }
    const uint32_t bx = blockIdx.x;
    uint32_t w = 0;
#pragma unroll
    for (uint32_t q = 0; q + 1 < MPC_GROUP_MAX; q++) w += bx >= A.end[q] ? 1u : 0u;      /* workgroup-uniform */
    const RunArgs &R = A.run[w];
    /* (seeded sets generate their controls level by level here: PREGEN's 20 KB of LDS per workgroup would not let
     * eight workgroups share a CU) */
    rollout_body<MPC_PREC_FAST, SEEDED, true, false>(probs, R.rud, R.thr, R.first, R.count, R.selection, R.rel, R.cost_last, R.cost_cum,
                                                      R.keys, R.snap, R.snap_depth, R.snap_shift, R.pkey, R.pord, R.pcnt,
                                                      bx - (w ? A.end[w - 1] : 0u), R.nblk, lds);
}

/* multi-GPU: lexicographic (score bits, global ordinal) minimum over the gathered per-shard winners */
__global__ __launch_bounds__(256) void combine_shards_kernel(const uint64_t *__restrict__ gathered, uint32_t n_shards,
                                                             uint32_t n_problems, uint64_t ordinal_stride,
                                                             uint64_t *__restrict__ out)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_problems) return;
    uint64_t k = ~0ull, o = ~0ull;
    for (uint32_t s = 0; s < n_shards; s++) {
        const uint64_t *g = gathered + ((uint64_t)s * n_problems + p) * 2;
        const uint64_t ord = g[1] == ~0ull ? ~0ull : g[1] + (uint64_t)s * ordinal_stride;
        lex_min(k, o, g[0], ord);
    }
    out[2 * p] = k; out[2 * p + 1] = o;
}

/* A winner record with its controls: [0] score bits, [1] ordinal, then (rudder, throttle) per level as doubles --
 * 2 + 2 * depth words.  pack_records_kernel builds them from the results of the last run (mpc_hip_copy_winner_records);
 * the finish kernel writes the same layout into a record sink (mpc_hip_set_winner_record_sink). */
__global__ __launch_bounds__(64) void pack_records_kernel(const DevResult *__restrict__ results, const uint64_t *__restrict__ winners,
                                                          const DevProblem *__restrict__ probs, uint32_t words, uint64_t *__restrict__ dst)
{
    const uint32_t p = blockIdx.x, l = threadIdx.x;
    uint64_t *rec = dst + (size_t)p * words;
    if (l == 0) { rec[0] = winners[2 * p]; rec[1] = winners[2 * p + 1]; }
    if ((int)l < probs[p].depth && 3 + 2 * l < words) {
        rec[2 + 2 * l] = (uint64_t)__double_as_longlong(results[p].traj[l + 1].rudder);
        rec[3 + 2 * l] = (uint64_t)__double_as_longlong(results[p].traj[l + 1].throttle);
    }
}

/* multi-GPU: the EXACT cross-shard decision in one kernel, after ONE all-gather of the winner records.  Inside a shard,
 * candidates within MPC_REFINE_REL of the shard's minimum are already decided by the strict model; ACROSS shards the
 * records carry fast scores (good to ~1.2e-7 relative), which cannot order two shards that close.  One wave per row
 * (solve x problem), one lane per shard: where more than one shard lies within refine_rel of the row's minimum, those
 * lanes evaluate their shard's winner once more with the strict model -- the reference's arithmetic -- FROM THE CONTROLS
 * IN ITS RECORD, so every rank can do it for every shard and all ranks decide alike without a second exchange; then the
 * lexicographic (score bits, global ordinal) minimum: the reference's first strict minimum (controller.cpp:158-166),
 * because the true minimum always lies inside the near set (refine_rel is 9x the fast model's error) and every record
 * outside it is larger than every strict score inside it.  refine_rel < 0 (strict-mode runs): plain minimum. */
__global__ __launch_bounds__(64) void combine_exact_kernel(const DevProblem *__restrict__ probs, uint32_t n_probs, int selection,
                                                           const uint64_t *__restrict__ gathered, uint32_t n_shards,
                                                           uint32_t n_rows, uint32_t words, uint64_t ordinal_stride,
                                                           double refine_rel, uint64_t *__restrict__ out,
                                                           uint32_t *__restrict__ tied_rows)
{
    const uint32_t row = blockIdx.x, s = threadIdx.x;
    const bool mine = s < n_shards;
    const uint64_t *rec = gathered + ((size_t)(mine ? s : 0) * n_rows + row) * words;
    uint64_t key = mine ? rec[0] : ~0ull;
    const uint64_t reported = mine ? rec[1] : ~0ull;
    const uint64_t ordinal = reported == ~0ull ? ~0ull : reported + (uint64_t)s * ordinal_stride;
    uint64_t kmin = key;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) kmin = min(kmin, (uint64_t)__shfl_xor((unsigned long long)kmin, m, 64));
    if (refine_rel >= 0.0 && kmin < KEY_INF) {
        const double m = __longlong_as_double((long long)kmin);
        const DevProblem &Pw = probs[row % n_probs];
        const bool near = mine && reported != ~0ull &&
                          __longlong_as_double((long long)key) <= mpc_refine_limit(m, refine_rel, Pw.wd, Pw.wh, Pw.ws);
        const uint32_t n_near = (uint32_t)__popcll(__ballot(near));
        if (n_near > 1 && s == 0) atomicAdd(tied_rows, 1u);            /* for reporting */
        if (n_near > 1 && near) {
            const DevProblem &P = probs[row % n_probs];
            const Weights w{P.wd, P.wh, P.ws};
            const Model M = P.model;
            Veh v{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0};
            double last = 0, cum = 0;
            for (int l = 0; l < P.depth && 3 + 2 * l < (int)words; l++) {
                const double r = __longlong_as_double((long long)rec[2 + 2 * l]), t = __longlong_as_double((long long)rec[3 + 2 * l]);
                last = strict_level_and_score<false>(P, M, w, P.cp, P.sub, P.sub, l, r, t, v);
                cum += last;
            }
            key = key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last);
        }
    }
    uint64_t k = key, o = ordinal;
    wave_argmin(k, o);
    if (s == 0) { out[2 * (size_t)row] = k; out[2 * (size_t)row + 1] = o; }
}

/* single explicit control sequence (double precision controls), mpc_hip_rollout */
__global__ __launch_bounds__(64) void rollout_one_kernel(const DevProblem *__restrict__ probs,
                                                         const double *__restrict__ controls, int precision,
                                                         DevResult *__restrict__ results)
{
    __shared__ double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    const DevProblem &P = probs[0];
    double *cpL = lds, *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    stage_problem(P, cpL, subL);
    if (threadIdx.x != 0) return;
    DevResult &R = results[0];
    const Weights w{P.wd, P.wh, P.ws};
    double last = 0, cum = 0;
    R.traj[0] = mpc_level_state{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
    if (precision == MPC_PREC_F64) {
        Veh s{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0};
        for (int l = 0; l < P.depth; l++) {
            const double r = controls[l], t = controls[MPC_MAX_DEPTH + l];
            strict_level(s, r, t, subL, P.nsub, P.cx, P.cy, P.model);
            last = strict_score(cpL + l * CP_STRIDE, s, w);
            cum += last;
            R.traj[l + 1] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, s.cmg, s.sog,
                                            clampd(r, -1, 1), clampd(t, 0, 1), cum};
        }
    } else {
        FastVeh s;
        s.x = P.sx; s.y = P.sy; s.v = P.sv; s.rpm = P.srpm; s.delta = 0;
        sincos(P.sh, &s.sh, &s.ch);
        const double *q_last = subL + (P.nsub - 1) * SUB_STRIDE;
        for (int l = 0; l < P.depth; l++) {
            const double r = controls[l], t = controls[MPC_MAX_DEPTH + l];
            fast_level<SUB_STRIDE, MPC_FAST_LEAN>(s, r, t, subL, P.nsub, P.model);
            last = fast_score<MPC_FAST_LEAN>(cpL + l * CP_STRIDE, s, w, q_last);
            cum += last;
            R.traj[l + 1] = fast_to_level_state(cpL + l * CP_STRIDE, q_last, s.x, s.y, s.sh, s.ch, s.v, s.rpm, s.delta,
                                                r, t, cum);
        }
    }
    R.key_bits = key_of(last); R.ordinal = 0; R.best_last = last; R.best_cum = cum; R.refined = 0; R.overflow = 0; R.tie = 0;
}

#endif
