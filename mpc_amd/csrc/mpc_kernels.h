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
    const float *staged;         /* explicit controls of THIS lane staged in LDS by stage_controls(), or nullptr */
    uint32_t staged_stride;      /* = blockDim.x */
};

/* K1 reads every level's controls of its candidates ONCE, all loads in flight together, into LDS
 * ([level][rudder|throttle][lane], FP32): one memory latency per kernel instead of one per level (a lone wave per
 * SIMD has nothing to hide those behind; rocprof showed a third of its life in s_waitcnt). */
__device__ inline void stage_controls(const float *__restrict__ rud, const float *__restrict__ thr, uint64_t n,
                                      uint64_t index, bool live, int depth, float *ctl)
{
    if (live) {
#pragma unroll 8
        for (int l = 0; l < depth; l++) {
            ctl[(2 * l) * blockDim.x + threadIdx.x] = rud[(uint64_t)l * n + index];
            ctl[(2 * l + 1) * blockDim.x + threadIdx.x] = thr[(uint64_t)l * n + index];
        }
    }
    /* each lane reads back only what it wrote: no barrier needed */
}

/* STAGED is a compile-time choice on purpose: a run-time choice between the LDS copy and the global arrays makes
 * hipcc select between two pointers and emit flat_ loads, whose vmcnt(0)+lgkmcnt(0) wait also drains every
 * outstanding snapshot store at the top of every level. */
template <bool SEEDED, bool STAGED = false>
__device__ inline void fetch_control(const ControlSource &c, uint64_t ordinal, uint64_t index, int level,
                                     double &r, double &t)
{
    if (SEEDED) {
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

/* Per-level state snapshots, SoA (passed BY VALUE: a pointer to it would put the record in scratch memory and turn
 * every store into a flat_ store that the LDS waits then have to drain; base == nullptr = no snapshots): field f of level l of candidate i at base[(l*SNAP_FIELDS + f)*stride + i].
 * strict: x y h v rpm cmg sog cum last      fast: x y sh ch v rpm delta cum last  (x, y through the water) */
#define SNAP_FIELDS 9
struct Snap {
    double *base; uint64_t stride, i;
    __device__ inline void put(int l, int f, double v) const { base[((uint64_t)l * SNAP_FIELDS + f) * stride + i] = v; }
    __device__ inline double get(int l, int f) const { return base[((uint64_t)l * SNAP_FIELDS + f) * stride + i]; }
};

/* Both evaluators: UNIFORM = every substep of a level has the same length (BASELINE configs 2-5: 20 x 0.05 s), so
 * the schedule row lives in registers instead of being re-read from LDS 400 times per candidate. */

/* strict evaluation of one candidate; traj (optional) receives depth+1 level states */
template <bool SEEDED, bool UNIFORM = false, bool STAGED = false>
__device__ inline void eval_strict(const DevProblem &P, const double *cpL, const double *subL,
                                   const ControlSource &c, uint64_t ordinal, uint64_t index,
                                   double &last, double &cum, mpc_level_state *traj, const Snap snap)
{
    Veh s{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0};
    const Weights w{P.wd, P.wh, P.ws};
    const double row0[1] = {subL[0]};
    last = 0.0; cum = 0.0;
    if (traj) traj[0] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
    for (int l = 0; l < P.depth; l++) {
        double r, t;
        fetch_control<SEEDED, STAGED>(c, ordinal, index, l, r, t);
        if (UNIFORM) strict_level<0>(s, r, t, row0, P.nsub, P.cx, P.cy);
        else strict_level<SUB_STRIDE>(s, r, t, subL, P.nsub, P.cx, P.cy);
        last = strict_score(cpL + l * CP_STRIDE, s, w);
        cum += last;
        if (traj)
            traj[l + 1] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, s.cmg, s.sog,
                                          clampd(r, -1, 1), clampd(t, 0, 1), cum};
        if (snap.base) {
            snap.put(l, 0, s.x); snap.put(l, 1, s.y); snap.put(l, 2, s.h); snap.put(l, 3, s.v);
            snap.put(l, 4, s.rpm); snap.put(l, 5, s.cmg); snap.put(l, 6, s.sog); snap.put(l, 7, cum);
            snap.put(l, 8, last);
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

template <bool SEEDED, bool UNIFORM = false, bool STAGED = false>
__device__ inline void eval_fast(const DevProblem &P, const double *cpL, const double *subL,
                                 const ControlSource &c, uint64_t ordinal, uint64_t index,
                                 double &last, double &cum, mpc_level_state *traj, const Snap snap)
{
    FastVeh s;
    s.x = P.sx; s.y = P.sy; s.v = P.sv; s.rpm = P.srpm; s.delta = 0;
    sincos(P.sh, &s.sh, &s.ch);
    const Weights w{P.wd, P.wh, P.ws};
    const double *q_last = subL + (P.nsub - 1) * SUB_STRIDE;
    const double row0[3] = {subL[0], subL[1], subL[2]};
    last = 0.0; cum = 0.0;
    if (traj) traj[0] = mpc_level_state{s.x, s.y, P.sh, s.v, s.rpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
    for (int l = 0; l < P.depth; l++) {
        double r, t;
        fetch_control<SEEDED, STAGED>(c, ordinal, index, l, r, t);
        if (UNIFORM) fast_level<0>(s, r, t, row0, P.nsub);
        else fast_level<SUB_STRIDE>(s, r, t, subL, P.nsub);
        last = fast_score(cpL + l * CP_STRIDE, s, w, q_last);
        cum += last;
        if (traj)
            traj[l + 1] = fast_to_level_state(cpL + l * CP_STRIDE, q_last, s.x, s.y, s.sh, s.ch, s.v, s.rpm, s.delta,
                                              r, t, cum);
        if (snap.base) {
            snap.put(l, 0, s.x); snap.put(l, 1, s.y); snap.put(l, 2, s.sh); snap.put(l, 3, s.ch);
            snap.put(l, 4, s.v); snap.put(l, 5, s.rpm); snap.put(l, 6, s.delta); snap.put(l, 7, cum);
            snap.put(l, 8, last);
        }
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* K1: rollout + cost + block argmin                                                                       */
/* ------------------------------------------------------------------------------------------------------ */
template <int PREC, bool SEEDED, bool UNIFORM>
__global__ __launch_bounds__(256) void rollout_kernel(const DevProblem *__restrict__ probs,
                                                      const float *__restrict__ rud, const float *__restrict__ thr,
                                                      uint64_t first, uint64_t count, int selection, double refine_rel,
                                                      double *__restrict__ cost_last, double *__restrict__ cost_cum,
                                                      double *__restrict__ keys, double *__restrict__ snap_base,
                                                      int snap_depth, uint64_t *__restrict__ pkey,
                                                      uint64_t *__restrict__ pord, uint32_t *__restrict__ pcnt)
{
    extern __shared__ double lds[];
    const DevProblem &P = probs[blockIdx.y];
    double *cpL = lds;
    double *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    uint64_t *red = (uint64_t *)(subL + MPC_MAX_SUBSTEPS * SUB_STRIDE);       /* 8 words argmin + 2 words (+ 2 pad) */
    float *ctlL = (float *)(red + 12);                                         /* [depth][2][blockDim] (explicit sets) */
    stage_problem(P, cpL, subL);

    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!SEEDED) stage_controls(rud, thr, count, i, i < count, P.depth, ctlL);
    uint64_t k = ~0ull, o = ~0ull;
    double mykey = __longlong_as_double((long long)KEY_INF);
    if (i < count) {
        const ControlSource c{rud, thr, count, P.seed, P.last_r, P.last_t,
                              SEEDED ? nullptr : ctlL + threadIdx.x, blockDim.x};
        const Snap snap{snap_base ? snap_base + (uint64_t)blockIdx.y * snap_depth * SNAP_FIELDS * count : nullptr, count, i};
        double last, cum;
        if (PREC == MPC_PREC_F64) eval_strict<SEEDED, UNIFORM, !SEEDED>(P, cpL, subL, c, first + i, i, last, cum, nullptr, snap);
        else eval_fast<SEEDED, UNIFORM, !SEEDED>(P, cpL, subL, c, first + i, i, last, cum, nullptr, snap);
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
    const bool near = (i < count) && (mykey <= m + m * refine_rel);
    const uint32_t wave_n = (uint32_t)__popcll(__ballot(near));
    if ((threadIdx.x & 63) == 0 && wave_n) atomicAdd(cnt, wave_n);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t b = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
        pkey[b] = k; pord[b] = o; pcnt[b] = *cnt;
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* K1, wave-pair form (fast model): the same arithmetic split over TWO waves per 64 candidates.            */
/*                                                                                                          */
/* A 65,536-candidate set is only 1024 wavefronts -- one per SIMD on 256 CUs x 4 SIMDs -- and one wave      */
/* alone keeps its SIMD's FP64 pipe ~55 % busy (dependent-issue latency; rocprof SQ_ACTIVE_INST_VALU).      */
/* The model splits cleanly: the speed half (rpm, v) never reads the pose, the pose half (sin h, cos h, x,  */
/* y, cost) needs only a = yaw*dt and delta = v*dt per substep.  Wave 0 of a 128-lane workgroup runs the    */
/* speed half one chunk of PAIR_CHUNK substeps AHEAD and hands (a, delta) to wave 1 through a two-slot LDS   */
/* ring; one workgroup barrier per chunk.  2048 waves -> two per SIMD -> the pipe stays fed.                */
/* Bit-identical to rollout_kernel<MPC_PREC_FAST>: same device functions, same order.                       */
/* ------------------------------------------------------------------------------------------------------ */
#define PAIR_CHUNK 10

template <bool SEEDED>
__global__ __launch_bounds__(128) void rollout_pair_kernel(const DevProblem *__restrict__ probs,
                                                           const float *__restrict__ rud, const float *__restrict__ thr,
                                                           uint64_t first, uint64_t count, int selection, double refine_rel,
                                                           double *__restrict__ cost_last, double *__restrict__ cost_cum,
                                                           double *__restrict__ keys, double *__restrict__ snap_base,
                                                           int snap_depth, uint64_t *__restrict__ pkey,
                                                           uint64_t *__restrict__ pord, uint32_t *__restrict__ pcnt)
{
    extern __shared__ double lds[];
    const DevProblem &P = probs[blockIdx.y];
    double *cpL = lds;
    double *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    double *ring = subL + MPC_MAX_SUBSTEPS * SUB_STRIDE;          /* [2][PAIR_CHUNK][2][64] */
    stage_problem(P, cpL, subL);

    const int role = threadIdx.x >> 6, lane = threadIdx.x & 63;   /* 0: speed half, 1: pose half */
    const uint64_t i = (uint64_t)blockIdx.x * 64 + lane;
    const bool live = i < count;
    const uint64_t ic = live ? i : count - 1;                      /* idle lanes shadow the last candidate */
    const int nsub = P.nsub, depth = P.depth;
    const int cpl = (nsub + PAIR_CHUNK - 1) / PAIR_CHUNK;          /* chunks per level */
    const int total = depth * cpl;
    const ControlSource c{rud, thr, count, P.seed, P.last_r, P.last_t, nullptr, 0};
    const Snap snap{snap_base + (uint64_t)blockIdx.y * snap_depth * SNAP_FIELDS * count, count, ic};
    const bool do_snap = snap_base && live;
    const Weights w{P.wd, P.wh, P.ws};
    const double *q_last = subL + (nsub - 1) * SUB_STRIDE;

    /* speed half state */
    double rpm = P.srpm, v = P.sv;
    FastLevelCoef coef{0, 0, 0};
    /* pose half state */
    FastVeh s;
    s.x = P.sx; s.y = P.sy; s.v = 0; s.rpm = 0; s.delta = 0; s.sh = 0; s.ch = 1;
    if (role == 1) sincos(P.sh, &s.sh, &s.ch);
    double last = 0.0, cum = 0.0;

    for (int t = 0; t <= total; t++) {
        if (role == 0) {
            if (t < total) {
                const int l = t / cpl, ck = t - l * cpl;
                const int k0 = ck * PAIR_CHUNK, k1 = min(k0 + PAIR_CHUNK, nsub);
                if (ck == 0) {
                    double r, th;
                    fetch_control<SEEDED>(c, first + ic, ic, l, r, th);
                    coef = fast_level_coef(r, th);
                }
                double *slot = ring + (size_t)(t & 1) * PAIR_CHUNK * 128 + lane;
#pragma unroll 5
                for (int k = k0; k < k1; k++) {
                    double a, delta;
                    fast_speed_substep(rpm, v, coef, subL + k * SUB_STRIDE, a, delta);
                    slot[(k - k0) * 128] = a;
                    slot[(k - k0) * 128 + 64] = delta;
                }
                if (ck == cpl - 1 && do_snap) { snap.put(l, 4, v); snap.put(l, 5, rpm); }
            }
        } else {
            if (t >= 1) {
                const int tt = t - 1;
                const int l = tt / cpl, ck = tt - l * cpl;
                const int k0 = ck * PAIR_CHUNK, k1 = min(k0 + PAIR_CHUNK, nsub);
                const double *slot = ring + (size_t)(tt & 1) * PAIR_CHUNK * 128 + lane;
#pragma unroll 5
                for (int k = k0; k < k1; k++) {
                    const double a = slot[(k - k0) * 128];
                    s.delta = slot[(k - k0) * 128 + 64];
                    fast_pose_substep(s.x, s.y, s.sh, s.ch, a, s.delta);
                }
                if (ck == cpl - 1) {
                    last = fast_score(cpL + l * CP_STRIDE, s, w, q_last);
                    cum += last;
                    if (do_snap) {
                        snap.put(l, 0, s.x); snap.put(l, 1, s.y); snap.put(l, 2, s.sh); snap.put(l, 3, s.ch);
                        snap.put(l, 6, s.delta); snap.put(l, 7, cum); snap.put(l, 8, last);
                    }
                }
            }
        }
        __syncthreads();
    }

    /* the pose wave owns the result: costs, key, argmin and near-tie count of the workgroup's 64 candidates */
    if (role == 1) {
        uint64_t k = ~0ull, o = ~0ull;
        double mykey = __longlong_as_double((long long)KEY_INF);
        if (live) {
            const uint64_t at = (uint64_t)blockIdx.y * count + i;
            if (cost_last) cost_last[at] = last;
            if (cost_cum) cost_cum[at] = cum;
            k = key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last);
            o = first + i;
            mykey = __longlong_as_double((long long)k);
            keys[at] = mykey;
        }
        wave_argmin(k, o);
        const double m = __longlong_as_double((long long)k);
        const uint32_t n_near = (uint32_t)__popcll(__ballot(live && (mykey <= m + m * refine_rel)));
        if (lane == 0) {
            const uint64_t b = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
            pkey[b] = k; pord[b] = o; pcnt[b] = n_near;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* K2: per problem -- reduce, refine list, strict decision among near-ties, the winner's trajectory        */
/* ------------------------------------------------------------------------------------------------------ */
template <bool SEEDED>
__global__ __launch_bounds__(256) void finish_kernel(const DevProblem *__restrict__ probs,
                                                     const float *__restrict__ rud, const float *__restrict__ thr,
                                                     uint64_t first, uint64_t count, int selection, int fast,
                                                     double refine_rel, uint32_t nblk, uint32_t k1_block,
                                                     const uint64_t *__restrict__ pkey, const uint64_t *__restrict__ pord,
                                                     const uint32_t *__restrict__ pcnt, const double *__restrict__ keys,
                                                     double *__restrict__ snap_base, int snap_depth,
                                                     uint64_t *__restrict__ winners, DevResult *__restrict__ results)
{
    __shared__ double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    __shared__ uint64_t s_list[LIST_CAP];
    __shared__ uint64_t red[8];
    __shared__ uint64_t s_win[2];
    __shared__ uint32_t s_n;
    const uint32_t p = blockIdx.x;
    const DevProblem &P = probs[p];
    double *cpL = lds, *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    if (threadIdx.x == 0) s_n = 0;
    stage_problem(P, cpL, subL);
    const ControlSource c{rud, thr, count, P.seed, P.last_r, P.last_t, nullptr, 0};
    DevResult &R = results[p];

    /* 1. the workgroup triples' minimum (lowest ordinal among equal keys) */
    uint64_t k = ~0ull, o = ~0ull;
    for (uint32_t j = threadIdx.x; j < nblk; j += blockDim.x)
        lex_min(k, o, pkey[(uint64_t)p * nblk + j], pord[(uint64_t)p * nblk + j]);
    block_argmin(k, o, red);
    if (threadIdx.x == 0) { s_win[0] = k; s_win[1] = o; }
    __syncthreads();
    k = s_win[0]; o = s_win[1];
    uint64_t win = o;
    uint32_t n = 0, overflow = 0;

    /* 2. fast mode: the candidates within refine_rel of the minimum */
    if (fast) {
        const double m = __longlong_as_double((long long)k);
        const double limit = m + m * refine_rel;
        for (uint32_t j = threadIdx.x; j < nblk; j += blockDim.x) {
            const uint64_t b = (uint64_t)p * nblk + j;
            if (__longlong_as_double((long long)pkey[b]) <= limit) {
                if (pcnt[b] <= 1) {
                    const uint32_t at = atomicAdd(&s_n, 1u);
                    if (at < LIST_CAP) s_list[at] = pord[b];
                } else {                                   /* several near-ties inside one K1 workgroup */
                    const uint64_t lo = (uint64_t)j * k1_block, hi = min(lo + k1_block, count);
                    for (uint64_t e = lo; e < hi; e++)
                        if (keys[(uint64_t)p * count + e] <= limit) {
                            const uint32_t at = atomicAdd(&s_n, 1u);
                            if (at < LIST_CAP) s_list[at] = first + e;
                        }
                }
            }
        }
        __syncthreads();
        n = s_n;
        if (n > LIST_CAP) { overflow = 1; n = LIST_CAP; }
        /* 3. more than one: the strict model decides, lowest ordinal among equal scores */
        if (n > 1) {
            uint64_t k2 = ~0ull, o2 = ~0ull;
            for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
                const uint64_t ord = s_list[e];
                double last, cum;
                eval_strict<SEEDED>(P, cpL, subL, c, ord, ord - first, last, cum, nullptr, Snap{nullptr, 0, 0});
                lex_min(k2, o2, key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last), ord);
            }
            block_argmin(k2, o2, red);
            if (threadIdx.x == 0) s_win[1] = o2;
            __syncthreads();
            win = s_win[1];
        } else if (n == 1) {
            win = s_list[0];
        }
    }

    /* 4. the winner's trajectory.  A fast-mode winner that had no rival within refine_rel keeps the fast
     *    model's states and scores (contract 1e-5 relative); otherwise everything returned is strict. */
    const bool keep_fast = fast && n <= 1;
    if (win == ~0ull) {
        if (threadIdx.x == 0) {
            R.key_bits = KEY_INF; R.ordinal = win; R.best_last = 0; R.best_cum = 0; R.refined = 0; R.overflow = overflow;
            winners[2 * p] = KEY_INF; winners[2 * p + 1] = win;
        }
        return;
    }
    if (snap_base && (keep_fast || !fast)) {
        /* K1 left every candidate's per-level state in HBM: one thread per level converts the winner's */
        const Snap snap{snap_base + (uint64_t)p * snap_depth * SNAP_FIELDS * count, count, win - first};
        const int l = threadIdx.x;
        if (l < P.depth) {
            double r, t;
            fetch_control<SEEDED>(c, win, win - first, l, r, t);
            if (fast)
                R.traj[l + 1] = fast_to_level_state(cpL + l * CP_STRIDE, subL + (P.nsub - 1) * SUB_STRIDE,
                                                    snap.get(l, 0), snap.get(l, 1), snap.get(l, 2), snap.get(l, 3),
                                                    snap.get(l, 4), snap.get(l, 5), snap.get(l, 6), r, t, snap.get(l, 7));
            else
                R.traj[l + 1] = mpc_level_state{snap.get(l, 0), snap.get(l, 1), snap.get(l, 2), snap.get(l, 3),
                                                snap.get(l, 4), snap.get(l, 5), snap.get(l, 6),
                                                clampd(r, -1, 1), clampd(t, 0, 1), snap.get(l, 7)};
        }
        if (threadIdx.x == 0) {
            const double last = snap.get(P.depth - 1, 8), cum = snap.get(P.depth - 1, 7);
            const uint64_t kb = key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last);
            R.traj[0] = mpc_level_state{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
            R.key_bits = kb; R.ordinal = win; R.best_last = last; R.best_cum = cum;
            R.refined = 0; R.overflow = overflow;
            winners[2 * p] = kb; winners[2 * p + 1] = win;
        }
    } else if (threadIdx.x == 0) {
        double last = 0, cum = 0;
        if (keep_fast) eval_fast<SEEDED>(P, cpL, subL, c, win, win - first, last, cum, R.traj, Snap{nullptr, 0, 0});
        else eval_strict<SEEDED>(P, cpL, subL, c, win, win - first, last, cum, R.traj, Snap{nullptr, 0, 0});
        const uint64_t kb = key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last);
        R.key_bits = kb; R.ordinal = win; R.best_last = last; R.best_cum = cum;
        R.refined = n > 1 ? n : 0; R.overflow = overflow;
        winners[2 * p] = kb; winners[2 * p + 1] = win;
    }
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
            strict_level(s, r, t, subL, P.nsub, P.cx, P.cy);
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
            fast_level(s, r, t, subL, P.nsub);
            last = fast_score(cpL + l * CP_STRIDE, s, w, q_last);
            cum += last;
            R.traj[l + 1] = fast_to_level_state(cpL + l * CP_STRIDE, q_last, s.x, s.y, s.sh, s.ch, s.v, s.rpm, s.delta,
                                                r, t, cum);
        }
    }
    R.key_bits = key_of(last); R.ordinal = 0; R.best_last = last; R.best_cum = cum; R.refined = 0; R.overflow = 0;
}

#endif
