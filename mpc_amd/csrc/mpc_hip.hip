/*
 * mpc_hip.hip -- kernels and C-ABI (include/mpc_hip.h) of the MI355X implementation of the afb2001/mpc hot path.
 *
 * Pipeline of one evaluation (all launches on the handle's stream, nothing allocated per call once warm):
 *   K1 rollout_kernel   one lane per candidate: D levels x S Euler substeps + tracking cost per level;
 *                       per-problem data (checkpoints, substep schedule, current) staged in LDS; costs written
 *                       coalesced (SoA); wavefront-shuffle + LDS argmin -> one (key, ordinal) pair per block.
 *   K2 reduce_kernel    per problem: min over the block pairs; threshold for the refine list.
 *   K3 compact_kernel   (fast mode) candidates whose key is within MPC_REFINE_REL of the minimum -> list.
 *   K4 finish_kernel    per problem: strict FP64 re-evaluation of the list, argmin (score, ordinal), strict
 *                       rollout of the winner with every level's state stored -> result + packed winner.
 * Selection follows controller.cpp:158-166: strictly smaller LAST-level score wins, so the lowest ordinal wins
 * ties; MPC_SELECT_CUMULATIVE compares the cumulative Score instead.
 *
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (strict code must not contract; the fast code
 * spells its FMAs out).
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/mpc_candidates.h"
#include "../../include/mpc_hip.h"
#include "mpc_model.h"

using namespace mpc;

/* ------------------------------------------------------------------------------------------------------ */
/* device-side problem / result records                                                                   */
/* ------------------------------------------------------------------------------------------------------ */
struct DevProblem {
    double sx, sy, sh, sv, srpm;
    double last_r, last_t;
    double cx, cy;
    double wd, wh, ws;
    double gr, gt, minr, midr, maxr, mint, maxt, tol;
    uint64_t seed;
    int32_t depth, nsub;
    double cp[MPC_MAX_DEPTH * CP_STRIDE];
    double sub[MPC_MAX_SUBSTEPS * SUB_STRIDE];
};

struct DevResult {
    uint64_t key_bits, ordinal;
    double best_last, best_cum;
    uint32_t refined, overflow;
    mpc_level_state traj[MPC_MAX_DEPTH + 1];
};

#define LIST_CAP 1024u
#define KEY_INF 0x7FF0000000000000ull

static thread_local char g_err[512] = "";
static void set_err(const char *fmt, const char *a = "", const char *b = "")
{
    snprintf(g_err, sizeof g_err, fmt, a, b);
}
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) { set_err("%s failed: %s", #expr, hipGetErrorString(e_)); return MPC_ERR_HIP; } \
    } while (0)

/* ------------------------------------------------------------------------------------------------------ */
/* device helpers                                                                                         */
/* ------------------------------------------------------------------------------------------------------ */
__device__ inline uint64_t key_of(double cost)
{
    /* scores are >= 0 (weights >= 0), so the bit pattern orders like the value; NaN and negatives -> +inf */
    return (cost >= 0.0) ? (uint64_t)__double_as_longlong(cost) : KEY_INF;
}

__device__ inline void lex_min(uint64_t &k, uint64_t &o, uint64_t k2, uint64_t o2)
{
    if (k2 < k || (k2 == k && o2 < o)) { k = k2; o = o2; }
}

__device__ inline void wave_argmin(uint64_t &k, uint64_t &o)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t k2 = __shfl_xor((unsigned long long)k, m, 64);
        const uint64_t o2 = __shfl_xor((unsigned long long)o, m, 64);
        lex_min(k, o, k2, o2);
    }
}

/* block argmin: result valid in thread 0.  red must hold 2 * (blockDim.x / 64) uint64. */
__device__ inline void block_argmin(uint64_t &k, uint64_t &o, uint64_t *red)
{
    wave_argmin(k, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    if (nw == 1) return;
    if (lane == 0) { red[2 * wave] = k; red[2 * wave + 1] = o; }
    __syncthreads();
    if (wave == 0) {
        k = (lane < nw) ? red[2 * lane] : ~0ull;
        o = (lane < nw) ? red[2 * lane + 1] : ~0ull;
        wave_argmin(k, o);
    }
}

/* stage checkpoints and substep schedule of one problem into LDS */
__device__ inline void stage_problem(const DevProblem &P, double *cpL, double *subL)
{
    const int ncp = P.depth * CP_STRIDE, nsb = P.nsub * SUB_STRIDE;
    for (int i = threadIdx.x; i < ncp; i += blockDim.x) cpL[i] = P.cp[i];
    for (int i = threadIdx.x; i < nsb; i += blockDim.x) subL[i] = P.sub[i];
    __syncthreads();
}

struct ControlSource {
    const float *rud, *thr;      /* explicit: [depth][n] */
    uint64_t n;                  /* stride of the explicit arrays */
    uint64_t seed;               /* seeded */
    double last_r, last_t;
};

template <bool SEEDED>
__device__ inline void fetch_control(const ControlSource &c, uint64_t ordinal, uint64_t index, int level,
                                     double &r, double &t)
{
    if (SEEDED) {
        mpc_candidate_control(c.seed, ordinal, (uint32_t)level, c.last_r, c.last_t, &r, &t);
    } else {
        r = (double)c.rud[(uint64_t)level * c.n + index];
        t = (double)c.thr[(uint64_t)level * c.n + index];
    }
}

/* strict evaluation of one candidate; traj (optional) receives depth+1 level states */
template <bool SEEDED>
__device__ inline void eval_strict(const DevProblem &P, const double *cpL, const double *subL,
                                   const ControlSource &c, uint64_t ordinal, uint64_t index,
                                   double &last, double &cum, mpc_level_state *traj)
{
    Veh s{P.sx, P.sy, P.sh, P.sv, P.srpm, 0.0, 0.0};
    const Weights w{P.wd, P.wh, P.ws};
    last = 0.0; cum = 0.0;
    if (traj) traj[0] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
    for (int l = 0; l < P.depth; l++) {
        double r, t;
        fetch_control<SEEDED>(c, ordinal, index, l, r, t);
        strict_level(s, r, t, subL, P.nsub, P.cx, P.cy);
        last = strict_score(cpL + l * CP_STRIDE, s, w);
        cum += last;
        if (traj) {
            r = r < -1 ? -1 : (r > 1 ? 1 : r);
            t = t < 0 ? 0 : (t > 1 ? 1 : t);
            traj[l + 1] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, s.cmg, s.sog, r, t, cum};
        }
    }
}

template <bool SEEDED>
__device__ inline void eval_fast(const DevProblem &P, const double *cpL, const double *subL,
                                 const ControlSource &c, uint64_t ordinal, uint64_t index,
                                 double &last, double &cum, mpc_level_state *traj)
{
    FastVeh s;
    s.x = P.sx; s.y = P.sy; s.v = P.sv; s.rpm = P.srpm; s.dx = 0; s.dy = 0;
    sincos(P.sh, &s.sh, &s.ch);
    const Weights w{P.wd, P.wh, P.ws};
    const double inv_dt_last = P.nsub > 0 ? subL[(P.nsub - 1) * SUB_STRIDE + 5] : 0.0;
    last = 0.0; cum = 0.0;
    if (traj) traj[0] = mpc_level_state{s.x, s.y, P.sh, s.v, s.rpm, 0.0, 0.0, P.last_r, P.last_t, 0.0};
    for (int l = 0; l < P.depth; l++) {
        double r, t;
        fetch_control<SEEDED>(c, ordinal, index, l, r, t);
        fast_level(s, r, t, subL, P.nsub);
        last = fast_score(cpL + l * CP_STRIDE, s, w, inv_dt_last);
        cum += last;
        if (traj) {
            double h = atan2(s.sh, s.ch);
            if (h < 0) h += TWO_PI;
            const double sog = sqrt(s.dx * s.dx + s.dy * s.dy) * inv_dt_last;
            r = r < -1 ? -1 : (r > 1 ? 1 : r);
            t = t < 0 ? 0 : (t > 1 ? 1 : t);
            traj[l + 1] = mpc_level_state{s.x, s.y, h, s.v, s.rpm, PI_2 - atan2(s.dy, s.dx), sog, r, t, cum};
        }
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* K1: rollout + cost + block argmin                                                                       */
/* ------------------------------------------------------------------------------------------------------ */
template <int PREC, bool SEEDED>
__global__ __launch_bounds__(256) void rollout_kernel(const DevProblem *__restrict__ probs,
                                                      const float *__restrict__ rud, const float *__restrict__ thr,
                                                      uint64_t first, uint64_t count, int selection,
                                                      double *__restrict__ cost_last, double *__restrict__ cost_cum,
                                                      double *__restrict__ keys, uint64_t *__restrict__ pkey,
                                                      uint64_t *__restrict__ pord)
{
    extern __shared__ double lds[];
    const DevProblem &P = probs[blockIdx.y];
    double *cpL = lds;
    double *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    uint64_t *red = (uint64_t *)(subL + MPC_MAX_SUBSTEPS * SUB_STRIDE);
    stage_problem(P, cpL, subL);

    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t k = ~0ull, o = ~0ull;
    if (i < count) {
        const ControlSource c{rud, thr, count, P.seed, P.last_r, P.last_t};
        double last, cum;
        if (PREC == MPC_PREC_F64) eval_strict<SEEDED>(P, cpL, subL, c, first + i, i, last, cum, nullptr);
        else eval_fast<SEEDED>(P, cpL, subL, c, first + i, i, last, cum, nullptr);
        const uint64_t at = (uint64_t)blockIdx.y * count + i;
        if (cost_last) cost_last[at] = last;
        if (cost_cum) cost_cum[at] = cum;
        const double key = selection == MPC_SELECT_CUMULATIVE ? cum : last;
        k = key_of(key);
        o = first + i;
        keys[at] = __longlong_as_double((long long)k);
    }
    block_argmin(k, o, red);
    if (threadIdx.x == 0) {
        const uint64_t b = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
        pkey[b] = k; pord[b] = o;
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* K2: per problem, reduce the block pairs; set the refine threshold                                       */
/* ------------------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void reduce_kernel(const uint64_t *__restrict__ pkey,
                                                     const uint64_t *__restrict__ pord, uint32_t nblk,
                                                     double refine_rel, uint64_t *__restrict__ winners,
                                                     double *__restrict__ threshold, uint32_t *__restrict__ list_count)
{
    __shared__ uint64_t red[8];
    const uint32_t p = blockIdx.x;
    uint64_t k = ~0ull, o = ~0ull;
    for (uint32_t j = threadIdx.x; j < nblk; j += blockDim.x)
        lex_min(k, o, pkey[(uint64_t)p * nblk + j], pord[(uint64_t)p * nblk + j]);
    block_argmin(k, o, red);
    if (threadIdx.x == 0) {
        winners[2 * p] = k; winners[2 * p + 1] = o;
        const double m = __longlong_as_double((long long)k);
        threshold[p] = m + m * refine_rel;        /* inf stays inf */
        list_count[p] = 0;
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* K3: candidates within the threshold -> refine list                                                      */
/* ------------------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void compact_kernel(const double *__restrict__ keys, uint64_t first,
                                                      uint64_t count, const double *__restrict__ threshold,
                                                      uint32_t *__restrict__ list_count, uint64_t *__restrict__ list)
{
    const uint32_t p = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (keys[(uint64_t)p * count + i] <= threshold[p]) {
        const uint32_t at = atomicAdd(&list_count[p], 1u);
        if (at < LIST_CAP) list[(uint64_t)p * LIST_CAP + at] = first + i;
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* K4: strict re-evaluation of the list, argmin, rollout of the winner                                     */
/* ------------------------------------------------------------------------------------------------------ */
template <bool SEEDED>
__global__ __launch_bounds__(64) void finish_kernel(const DevProblem *__restrict__ probs,
                                                    const float *__restrict__ rud, const float *__restrict__ thr,
                                                    uint64_t first, uint64_t count, int selection, int use_list,
                                                    const uint32_t *__restrict__ list_count,
                                                    const uint64_t *__restrict__ list,
                                                    uint64_t *__restrict__ winners, DevResult *__restrict__ results)
{
    __shared__ double lds[MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE];
    __shared__ uint64_t bcast[2];
    const uint32_t p = blockIdx.x;
    const DevProblem &P = probs[p];
    double *cpL = lds, *subL = lds + MPC_MAX_DEPTH * CP_STRIDE;
    stage_problem(P, cpL, subL);
    const ControlSource c{rud, thr, count, P.seed, P.last_r, P.last_t};
    DevResult &R = results[p];

    uint64_t win = winners[2 * p + 1];
    uint32_t n = 0, overflow = 0;
    if (use_list) {
        n = list_count[p];
        if (n > LIST_CAP) { overflow = 1; n = LIST_CAP; }
        if (n == 1) win = list[(uint64_t)p * LIST_CAP];
        else if (n > 1) {
            uint64_t k = ~0ull, o = ~0ull;
            for (uint32_t e = threadIdx.x; e < n; e += 64) {
                const uint64_t ord = list[(uint64_t)p * LIST_CAP + e];
                double last, cum;
                eval_strict<SEEDED>(P, cpL, subL, c, ord, ord - first, last, cum, nullptr);
                lex_min(k, o, key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last), ord);
            }
            wave_argmin(k, o);
            win = o;
        }
    }
    if (threadIdx.x == 0) {
        double last = 0, cum = 0;
        if (win != ~0ull) eval_strict<SEEDED>(P, cpL, subL, c, win, win - first, last, cum, R.traj);
        const uint64_t kb = (win != ~0ull) ? key_of(selection == MPC_SELECT_CUMULATIVE ? cum : last) : KEY_INF;
        R.key_bits = kb; R.ordinal = win; R.best_last = last; R.best_cum = cum;
        R.refined = n; R.overflow = overflow;
        winners[2 * p] = kb; winners[2 * p + 1] = win;
    }
    (void)bcast;
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
            double r = controls[l], t = controls[MPC_MAX_DEPTH + l];
            strict_level(s, r, t, subL, P.nsub, P.cx, P.cy);
            last = strict_score(cpL + l * CP_STRIDE, s, w);
            cum += last;
            r = r < -1 ? -1 : (r > 1 ? 1 : r);
            t = t < 0 ? 0 : (t > 1 ? 1 : t);
            R.traj[l + 1] = mpc_level_state{s.x, s.y, s.h, s.v, s.rpm, s.cmg, s.sog, r, t, cum};
        }
    } else {
        FastVeh s;
        s.x = P.sx; s.y = P.sy; s.v = P.sv; s.rpm = P.srpm; s.dx = 0; s.dy = 0;
        sincos(P.sh, &s.sh, &s.ch);
        const double inv_dt_last = P.nsub > 0 ? subL[(P.nsub - 1) * SUB_STRIDE + 5] : 0.0;
        for (int l = 0; l < P.depth; l++) {
            double r = controls[l], t = controls[MPC_MAX_DEPTH + l];
            fast_level(s, r, t, subL, P.nsub);
            last = fast_score(cpL + l * CP_STRIDE, s, w, inv_dt_last);
            cum += last;
            double h = atan2(s.sh, s.ch);
            if (h < 0) h += TWO_PI;
            const double sog = sqrt(s.dx * s.dx + s.dy * s.dy) * inv_dt_last;
            r = r < -1 ? -1 : (r > 1 ? 1 : r);
            t = t < 0 ? 0 : (t > 1 ? 1 : t);
            R.traj[l + 1] = mpc_level_state{s.x, s.y, h, s.v, s.rpm, PI_2 - atan2(s.dy, s.dx), sog, r, t, cum};
        }
    }
    R.key_bits = key_of(last); R.ordinal = 0; R.best_last = last; R.best_cum = cum; R.refined = 0; R.overflow = 0;
}

/* ------------------------------------------------------------------------------------------------------ */
/* host side                                                                                               */
/* ------------------------------------------------------------------------------------------------------ */
struct mpc_hip_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   /* 0 start, 1 K1 begin, 2 K1 end, 3 end */

    DevProblem *d_probs = nullptr, *h_probs = nullptr;          /* h_probs pinned */
    uint32_t cap_probs = 0, n_probs = 0;
    DevResult *d_results = nullptr, *h_results = nullptr;       /* h_results pinned */
    uint64_t *d_winners = nullptr;
    double *d_threshold = nullptr;
    uint32_t *d_list_count = nullptr;
    uint64_t *d_list = nullptr;
    uint64_t *d_pkey = nullptr, *d_pord = nullptr;
    size_t cap_partials = 0;
    double *d_keys = nullptr;
    size_t cap_keys = 0;
    double *d_controls = nullptr;                               /* 2 * MPC_MAX_DEPTH, mpc_hip_rollout */

    /* description of the last run, for fetch_results */
    uint64_t last_count = 0, last_first = 0;
    int last_precision = 0, last_selection = 0, last_seeded = 0;
    bool ran = false;
    std::vector<int32_t> depth, nsub;
};

static int use_device(const mpc_hip_handle *h)
{
    HIP_TRY(hipSetDevice(h->device));
    return MPC_OK;
}

extern "C" int mpc_hip_abi_version(void) { return MPC_HIP_ABI_VERSION; }

extern "C" const char *mpc_hip_status_string(int status)
{
    switch (status) {
    case MPC_OK: return "ok";
    case MPC_ERR_INVALID: return "invalid argument";
    case MPC_ERR_NO_DEVICE: return "no usable HIP device";
    case MPC_ERR_HIP: return "HIP runtime error";
    case MPC_ERR_EMPTY_PLAN: return "Cannot run MPC on an empty plan";
    case MPC_ERR_NO_ITERATION: return "MPC failed to complete a single iteration";
    case MPC_ERR_NON_FINITE: return "MPC selected non-finite controls";
    case MPC_ERR_STALE_TRAJECTORY: return "trajectory number is no longer valid";
    case MPC_ERR_TOO_LARGE: return "problem exceeds the library limits";
    default: return "unknown status";
    }
}

extern "C" const char *mpc_hip_last_error(void) { return g_err; }

extern "C" void mpc_hip_default_problem(mpc_problem *p)
{
    memset(p, 0, sizeof *p);
    p->w_distance = 1.0; p->w_heading = 20.0; p->w_speed = 5.0;          /* cfg/mpc.cfg:11-15 */
    p->rudder_granularity = 0.0625; p->throttle_granularity = 0.125;     /* cfg/mpc.cfg:8-9   */
    p->min_rudder = -1; p->mid_rudder = 0; p->max_rudder = 1;            /* controller.h:146  */
    p->min_throttle = 0; p->max_throttle = 1;                            /* controller.h:147  */
    p->tolerance = 1.0e-5;                                               /* controller.h:231  */
}

extern "C" int mpc_hip_substep_schedule(double t, double *out, int cap)
{
    int n = 0;
    while (t > 0) {                      /* VehicleState.h:49-53 */
        const double d = t < 0.1 ? t : 0.1;
        t -= 0.1;
        if (out && n < cap) out[n] = d;
        n++;
    }
    return n;
}

extern "C" int mpc_hip_create(int device, void *stream, mpc_hip_handle **out)
{
    if (!out) { set_err("mpc_hip_create: out is NULL"); return MPC_ERR_INVALID; }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_err("no HIP device visible"); return MPC_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { set_err("device index out of range"); return MPC_ERR_NO_DEVICE; }
    mpc_hip_handle *h = new mpc_hip_handle();
    h->device = device;
    HIP_TRY(hipSetDevice(device));
    if (stream) h->stream = (hipStream_t)stream;
    else { HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    for (auto &e : h->ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipMalloc(&h->d_controls, sizeof(double) * 2 * MPC_MAX_DEPTH));
    *out = h;
    return MPC_OK;
}

extern "C" void mpc_hip_destroy(mpc_hip_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_probs); (void)hipHostFree(h->h_probs);
    (void)hipFree(h->d_results); (void)hipHostFree(h->h_results);
    (void)hipFree(h->d_winners); (void)hipFree(h->d_threshold); (void)hipFree(h->d_list_count);
    (void)hipFree(h->d_list); (void)hipFree(h->d_pkey); (void)hipFree(h->d_pord); (void)hipFree(h->d_keys);
    (void)hipFree(h->d_controls);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int mpc_hip_synchronize(mpc_hip_handle *h)
{
    if (!h) { set_err("handle is NULL"); return MPC_ERR_INVALID; }
    if (int s = use_device(h)) return s;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MPC_OK;
}

static int ensure_problem_capacity(mpc_hip_handle *h, uint32_t n)
{
    if (n <= h->cap_probs) return MPC_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_probs); (void)hipHostFree(h->h_probs); (void)hipFree(h->d_results);
    (void)hipHostFree(h->h_results); (void)hipFree(h->d_winners); (void)hipFree(h->d_threshold);
    (void)hipFree(h->d_list_count); (void)hipFree(h->d_list);
    h->d_probs = nullptr; h->h_probs = nullptr; h->d_results = nullptr; h->h_results = nullptr;
    h->d_winners = nullptr; h->d_threshold = nullptr; h->d_list_count = nullptr; h->d_list = nullptr;
    h->cap_probs = 0;
    HIP_TRY(hipMalloc(&h->d_probs, sizeof(DevProblem) * n));
    HIP_TRY(hipHostMalloc(&h->h_probs, sizeof(DevProblem) * n));
    HIP_TRY(hipMalloc(&h->d_results, sizeof(DevResult) * n));
    HIP_TRY(hipHostMalloc(&h->h_results, sizeof(DevResult) * n));
    HIP_TRY(hipMalloc(&h->d_winners, sizeof(uint64_t) * 2 * n));
    HIP_TRY(hipMalloc(&h->d_threshold, sizeof(double) * n));
    HIP_TRY(hipMalloc(&h->d_list_count, sizeof(uint32_t) * n));
    HIP_TRY(hipMalloc(&h->d_list, sizeof(uint64_t) * (size_t)LIST_CAP * n));
    h->cap_probs = n;
    return MPC_OK;
}

static int check_problem(const mpc_problem *p, int precision)
{
    if (p->depth < 1 || p->depth > MPC_MAX_DEPTH) { set_err("problem depth out of range [1, MPC_MAX_DEPTH]"); return MPC_ERR_TOO_LARGE; }
    if (p->n_substeps < 1 || p->n_substeps > MPC_MAX_SUBSTEPS) { set_err("problem n_substeps out of range [1, MPC_MAX_SUBSTEPS]"); return MPC_ERR_TOO_LARGE; }
    for (int s = 0; s < p->n_substeps; s++) {
        if (!(p->substeps[s] > 0)) { set_err("substep lengths must be > 0"); return MPC_ERR_INVALID; }
        if (precision == MPC_PREC_FAST && p->substeps[s] > 0.1000001) {
            set_err("MPC_PREC_FAST needs substeps <= 0.1 s (VehicleState.h:50); use MPC_PREC_F64"); return MPC_ERR_INVALID;
        }
    }
    return MPC_OK;
}

static void pack_problem(const mpc_problem *p, uint64_t seed, DevProblem *d)
{
    d->sx = p->start_x; d->sy = p->start_y; d->sh = p->start_heading; d->sv = p->start_speed;
    d->srpm = rpm_from_speed(p->start_speed);
    d->last_r = p->last_rudder; d->last_t = p->last_throttle;
    d->cx = p->current_x; d->cy = p->current_y;
    d->wd = p->w_distance; d->wh = p->w_heading; d->ws = p->w_speed;
    d->gr = p->rudder_granularity; d->gt = p->throttle_granularity;
    d->minr = p->min_rudder; d->midr = p->mid_rudder; d->maxr = p->max_rudder;
    d->mint = p->min_throttle; d->maxt = p->max_throttle; d->tol = p->tolerance;
    d->seed = seed; d->depth = p->depth; d->nsub = p->n_substeps;
    for (int l = 0; l < p->depth; l++) {
        double *c = d->cp + l * CP_STRIDE;
        c[0] = p->checkpoints[l].x; c[1] = p->checkpoints[l].y;
        c[2] = p->checkpoints[l].heading; c[3] = p->checkpoints[l].speed;
        c[4] = sin(p->checkpoints[l].heading); c[5] = cos(p->checkpoints[l].heading);
    }
    for (int s = 0; s < p->n_substeps; s++) {
        double *q = d->sub + s * SUB_STRIDE;
        const double dt = p->substeps[s];
        q[0] = dt; q[1] = MAX_RPM_CHANGE_RATE * dt; q[2] = dt / MASS;
        q[3] = p->current_x * dt; q[4] = p->current_y * dt; q[5] = 1.0 / dt;
    }
}

extern "C" int mpc_hip_upload_problems(mpc_hip_handle *h, const mpc_problem *h_problems, const uint64_t *h_seeds,
                                       uint32_t n_problems)
{
    if (!h || !h_problems || n_problems == 0) { set_err("upload_problems: NULL handle/problems or zero problems"); return MPC_ERR_INVALID; }
    if (int s = use_device(h)) return s;
    for (uint32_t p = 0; p < n_problems; p++)
        if (int s = check_problem(&h_problems[p], MPC_PREC_F64)) return s;
    if (int s = ensure_problem_capacity(h, n_problems)) return s;
    HIP_TRY(hipStreamSynchronize(h->stream));     /* the pinned staging buffer may still be in flight */
    h->depth.resize(n_problems); h->nsub.resize(n_problems);
    for (uint32_t p = 0; p < n_problems; p++) {
        pack_problem(&h_problems[p], h_seeds ? h_seeds[p] : 0, &h->h_probs[p]);
        h->depth[p] = h_problems[p].depth; h->nsub[p] = h_problems[p].n_substeps;
    }
    HIP_TRY(hipMemcpyAsync(h->d_probs, h->h_probs, sizeof(DevProblem) * n_problems, hipMemcpyHostToDevice, h->stream));
    h->n_probs = n_problems;
    h->ran = false;
    return MPC_OK;
}

static int ensure_run_capacity(mpc_hip_handle *h, uint64_t count, uint32_t nblk)
{
    const size_t need_keys = (size_t)h->n_probs * count, need_part = (size_t)h->n_probs * nblk;
    if (need_keys > h->cap_keys) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_keys); h->d_keys = nullptr; h->cap_keys = 0;
        HIP_TRY(hipMalloc(&h->d_keys, sizeof(double) * need_keys));
        h->cap_keys = need_keys;
    }
    if (need_part > h->cap_partials) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_pkey); (void)hipFree(h->d_pord); h->d_pkey = h->d_pord = nullptr; h->cap_partials = 0;
        HIP_TRY(hipMalloc(&h->d_pkey, sizeof(uint64_t) * need_part));
        HIP_TRY(hipMalloc(&h->d_pord, sizeof(uint64_t) * need_part));
        h->cap_partials = need_part;
    }
    return MPC_OK;
}

static int run_common(mpc_hip_handle *h, bool seeded, const float *d_rud, const float *d_thr, uint64_t first,
                      uint64_t count, int precision, int selection, double *d_cost_last, double *d_cost_cum)
{
    if (!h) { set_err("handle is NULL"); return MPC_ERR_INVALID; }
    if (h->n_probs == 0) { set_err("no problems uploaded"); return MPC_ERR_INVALID; }
    if (count == 0) { set_err("candidate count is zero"); return MPC_ERR_INVALID; }
    if (precision != MPC_PREC_FAST && precision != MPC_PREC_F64) { set_err("unknown precision"); return MPC_ERR_INVALID; }
    if (selection != MPC_SELECT_LAST_STEP && selection != MPC_SELECT_CUMULATIVE) { set_err("unknown selection"); return MPC_ERR_INVALID; }
    if (!seeded && (!d_rud || !d_thr)) { set_err("control arrays are NULL"); return MPC_ERR_INVALID; }
    if (!seeded && h->n_probs != 1) { set_err("explicit control arrays take exactly one problem"); return MPC_ERR_INVALID; }
    if (count > (1ull << 40)) { set_err("candidate count too large"); return MPC_ERR_TOO_LARGE; }
    if (int s = use_device(h)) return s;
    if (precision == MPC_PREC_FAST)
        for (uint32_t p = 0; p < h->n_probs; p++)
            for (int s = 0; s < h->nsub[p]; s++)
                if (h->h_probs[p].sub[s * SUB_STRIDE] > 0.1000001) {
                    set_err("MPC_PREC_FAST needs substeps <= 0.1 s (VehicleState.h:50); use MPC_PREC_F64");
                    return MPC_ERR_INVALID;
                }
    /* one wave per workgroup while the set is small (spreads waves over all SIMDs), 256 lanes otherwise */
    const uint32_t block = (count * h->n_probs >= (1u << 20)) ? 256u : 64u;
    const uint64_t nblk64 = (count + block - 1) / block;
    if (nblk64 > 0x7FFFFFFFull) { set_err("candidate count too large for one launch"); return MPC_ERR_TOO_LARGE; }
    const uint32_t nblk = (uint32_t)nblk64;
    if (h->n_probs > 65535) { set_err("more than 65535 problems in one batch"); return MPC_ERR_TOO_LARGE; }
    if (int s = ensure_run_capacity(h, count, nblk)) return s;

    const dim3 grid(nblk, h->n_probs), blk(block);
    const size_t lds = sizeof(double) * (MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE) + sizeof(uint64_t) * 8;
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
#define LAUNCH_K1(PREC, SEEDED)                                                                               \
    hipLaunchKernelGGL((rollout_kernel<PREC, SEEDED>), grid, blk, lds, h->stream, h->d_probs, d_rud, d_thr,   \
                       first, count, selection, d_cost_last, d_cost_cum, h->d_keys, h->d_pkey, h->d_pord)
    if (precision == MPC_PREC_F64) { if (seeded) LAUNCH_K1(MPC_PREC_F64, true); else LAUNCH_K1(MPC_PREC_F64, false); }
    else { if (seeded) LAUNCH_K1(MPC_PREC_FAST, true); else LAUNCH_K1(MPC_PREC_FAST, false); }
#undef LAUNCH_K1
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    const double rel = precision == MPC_PREC_FAST ? MPC_REFINE_REL : 0.0;
    hipLaunchKernelGGL(reduce_kernel, dim3(h->n_probs), dim3(256), 0, h->stream, h->d_pkey, h->d_pord, nblk, rel,
                       h->d_winners, h->d_threshold, h->d_list_count);
    HIP_TRY(hipGetLastError());
    const int use_list = precision == MPC_PREC_FAST;
    if (use_list) {
        hipLaunchKernelGGL(compact_kernel, dim3((uint32_t)((count + 255) / 256), h->n_probs), dim3(256), 0, h->stream,
                           h->d_keys, first, count, h->d_threshold, h->d_list_count, h->d_list);
        HIP_TRY(hipGetLastError());
    }
    if (seeded)
        hipLaunchKernelGGL((finish_kernel<true>), dim3(h->n_probs), dim3(64), 0, h->stream, h->d_probs, d_rud, d_thr,
                           first, count, selection, use_list, h->d_list_count, h->d_list, h->d_winners, h->d_results);
    else
        hipLaunchKernelGGL((finish_kernel<false>), dim3(h->n_probs), dim3(64), 0, h->stream, h->d_probs, d_rud, d_thr,
                           first, count, selection, use_list, h->d_list_count, h->d_list, h->d_winners, h->d_results);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    h->last_count = count; h->last_first = first; h->last_precision = precision; h->last_selection = selection;
    h->last_seeded = seeded; h->ran = true;
    return MPC_OK;
}

extern "C" int mpc_hip_run_seeded(mpc_hip_handle *h, uint64_t first_ordinal, uint64_t count, int precision,
                                  int selection, double *d_cost_last, double *d_cost_cum)
{
    return run_common(h, true, nullptr, nullptr, first_ordinal, count, precision, selection, d_cost_last, d_cost_cum);
}

extern "C" int mpc_hip_run_candidates(mpc_hip_handle *h, const float *d_rudders, const float *d_throttles,
                                      uint64_t n, int precision, int selection, double *d_cost_last, double *d_cost_cum)
{
    return run_common(h, false, d_rudders, d_throttles, 0, n, precision, selection, d_cost_last, d_cost_cum);
}

static void fill_result(const mpc_hip_handle *h, uint32_t p, const DevResult &R, float k1_ms, float total_ms,
                        mpc_result *out)
{
    memset(out, 0, sizeof *out);
    const int D = h->depth[p];
    out->levels = D;
    out->best_ordinal = R.ordinal;
    out->best_last = R.best_last; out->best_cumulative = R.best_cum;
    out->first_rudder = R.traj[1].rudder; out->first_throttle = R.traj[1].throttle;
    out->evaluated = h->last_count;
    out->rollout_steps = h->last_count * (uint64_t)D * (uint64_t)h->nsub[p];
    out->refined = R.refined;
    out->kernel_ms = k1_ms; out->total_ms = total_ms;
    memcpy(out->trajectory, R.traj, sizeof(mpc_level_state) * (size_t)(D + 1));
    out->status = MPC_OK;
    if (!isfinite(out->first_rudder) || !isfinite(out->first_throttle)) out->status = MPC_ERR_NON_FINITE;   /* controller.cpp:201-202 */
}

extern "C" int mpc_hip_fetch_results(mpc_hip_handle *h, mpc_result *h_results, uint32_t n_problems)
{
    if (!h || !h_results) { set_err("fetch_results: NULL argument"); return MPC_ERR_INVALID; }
    if (!h->ran) { set_err("fetch_results: nothing has run since the last upload"); return MPC_ERR_INVALID; }
    if (n_problems != h->n_probs) { set_err("fetch_results: n_problems differs from the uploaded batch"); return MPC_ERR_INVALID; }
    if (int s = use_device(h)) return s;
    HIP_TRY(hipMemcpyAsync(h->h_results, h->d_results, sizeof(DevResult) * n_problems, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float k1 = 0, total = 0;
    (void)hipEventElapsedTime(&k1, h->ev[1], h->ev[2]);
    (void)hipEventElapsedTime(&total, h->ev[0], h->ev[3]);
    bool overflow = false;
    for (uint32_t p = 0; p < n_problems; p++) overflow |= h->h_results[p].overflow != 0;
    if (overflow && h->last_precision == MPC_PREC_FAST) {
        /* more than LIST_CAP candidates tie with the minimum to within MPC_REFINE_REL: decide the whole set
         * with the strict kernel (whose block argmin already resolves ties by ordinal) */
        return MPC_ERR_HIP + 1000;   /* handled by the callers below: rerun strict */
    }
    for (uint32_t p = 0; p < n_problems; p++) fill_result(h, p, h->h_results[p], k1, total, &h_results[p]);
    return MPC_OK;
}

extern "C" int mpc_hip_device_winners(mpc_hip_handle *h, uint64_t **d_winners)
{
    if (!h || !d_winners) { set_err("device_winners: NULL argument"); return MPC_ERR_INVALID; }
    *d_winners = h->d_winners;
    return MPC_OK;
}

static int solve_common(mpc_hip_handle *h, const mpc_problem *probs, const uint64_t *seeds, uint32_t n_problems,
                        bool seeded, const float *d_rud, const float *d_thr, uint64_t first, uint64_t count,
                        int precision, int selection, double *d_cost_last, double *d_cost_cum, mpc_result *results)
{
    if (!results) { set_err("result pointer is NULL"); return MPC_ERR_INVALID; }
    if (int s = mpc_hip_upload_problems(h, probs, seeds, n_problems)) return s;
    if (int s = run_common(h, seeded, d_rud, d_thr, first, count, precision, selection, d_cost_last, d_cost_cum)) return s;
    int s = mpc_hip_fetch_results(h, results, n_problems);
    if (s == MPC_ERR_HIP + 1000) {
        /* the reported costs stay the fast ones; only the decision is redone (into the library's own buffers) */
        if (int s2 = run_common(h, seeded, d_rud, d_thr, first, count, MPC_PREC_F64, selection, nullptr, nullptr)) return s2;
        s = mpc_hip_fetch_results(h, results, n_problems);
    }
    return s;
}

extern "C" int mpc_hip_eval_candidates(mpc_hip_handle *h, const mpc_problem *h_problem, const float *d_rudders,
                                       const float *d_throttles, uint64_t n, int precision, int selection,
                                       double *d_cost_last, double *d_cost_cum, mpc_result *h_result)
{
    if (!h_problem) { set_err("problem is NULL"); return MPC_ERR_INVALID; }
    if (int s = check_problem(h_problem, precision)) return s;
    return solve_common(h, h_problem, nullptr, 1, false, d_rudders, d_throttles, 0, n, precision, selection,
                        d_cost_last, d_cost_cum, h_result);
}

extern "C" int mpc_hip_eval_seeded(mpc_hip_handle *h, const mpc_problem *h_problem, uint64_t seed,
                                   uint64_t first_ordinal, uint64_t count, int precision, int selection,
                                   double *d_cost_last, double *d_cost_cum, mpc_result *h_result)
{
    if (!h_problem) { set_err("problem is NULL"); return MPC_ERR_INVALID; }
    if (int s = check_problem(h_problem, precision)) return s;
    return solve_common(h, h_problem, &seed, 1, true, nullptr, nullptr, first_ordinal, count, precision, selection,
                        d_cost_last, d_cost_cum, h_result);
}

extern "C" int mpc_hip_eval_seeded_batch(mpc_hip_handle *h, const mpc_problem *h_problems, const uint64_t *h_seeds,
                                         uint32_t n_problems, uint64_t count, int precision, int selection,
                                         double *d_cost_last, double *d_cost_cum, mpc_result *h_results)
{
    if (!h_problems || !h_seeds) { set_err("problems/seeds are NULL"); return MPC_ERR_INVALID; }
    for (uint32_t p = 0; p < n_problems; p++)
        if (int s = check_problem(&h_problems[p], precision)) return s;
    return solve_common(h, h_problems, h_seeds, n_problems, true, nullptr, nullptr, 0, count, precision, selection,
                        d_cost_last, d_cost_cum, h_results);
}

extern "C" int mpc_hip_rollout(mpc_hip_handle *h, const mpc_problem *h_problem, const double *h_rudders,
                               const double *h_throttles, int precision, mpc_result *h_result)
{
    if (!h || !h_problem || !h_rudders || !h_throttles || !h_result) { set_err("rollout: NULL argument"); return MPC_ERR_INVALID; }
    if (precision != MPC_PREC_FAST && precision != MPC_PREC_F64) { set_err("unknown precision"); return MPC_ERR_INVALID; }
    if (int s = check_problem(h_problem, precision)) return s;
    if (int s = mpc_hip_upload_problems(h, h_problem, nullptr, 1)) return s;
    double ctl[2 * MPC_MAX_DEPTH] = {0};
    for (int l = 0; l < h_problem->depth; l++) { ctl[l] = h_rudders[l]; ctl[MPC_MAX_DEPTH + l] = h_throttles[l]; }
    HIP_TRY(hipMemcpyAsync(h->d_controls, ctl, sizeof ctl, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));      /* ctl is a stack buffer */
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    hipLaunchKernelGGL(rollout_one_kernel, dim3(1), dim3(64), 0, h->stream, h->d_probs, h->d_controls, precision, h->d_results);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    h->last_count = 1; h->last_first = 0; h->last_precision = MPC_PREC_F64; h->last_selection = 0; h->ran = true;
    return mpc_hip_fetch_results(h, h_result, 1);
}

/* tree search: TEMPORARY stub until mpc_tree.hip lands */
extern "C" int mpc_hip_solve_tree(mpc_hip_handle *, const mpc_problem *, int, int, mpc_time_fn, void *, double,
                                  uint64_t, mpc_tree_level *, mpc_result *)
{
    set_err("mpc_hip_solve_tree: not built yet");
    return MPC_ERR_INVALID;
}
