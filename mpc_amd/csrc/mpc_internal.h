/*
 * mpc_internal.h -- records and helpers shared by mpc_hip.hip (candidate sets) and mpc_tree.hip (tree search).
 * Not part of the ABI.
 */
#ifndef MPC_INTERNAL_H
#define MPC_INTERNAL_H

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/mpc_hip.h"
#include "mpc_model.h"

/* ---- device-side problem / result records --------------------------------------------------------------- */
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
#define MPC_SNAPSHOT_BYTES_MAX (2ull << 30)
#define TIMING_RING 1024          /* runs whose HIP-event times mpc_hip_timing can still resolve */
#define MPC_INTERNAL_RERUN_STRICT 1000   /* fetch: refine list overflowed, decide with the strict kernel */

void mpc_set_err(const char *fmt, const char *a = "", const char *b = "");

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) { mpc_set_err("%s failed: %s", #expr, hipGetErrorString(e_)); return MPC_ERR_HIP; } \
    } while (0)

/* one level of the breadth-first tree expansion (mpc_tree.hip), SoA */
struct TreeLevelBuf {
    double *x = nullptr, *y = nullptr, *h = nullptr, *v = nullptr, *rpm = nullptr, *cmg = nullptr, *sog = nullptr;
    double *last_r = nullptr, *last_t = nullptr, *cum = nullptr, *last_score = nullptr;
    uint32_t *parent = nullptr, *count = nullptr, *offset = nullptr;   /* offset has cap + 1 entries */
    size_t cap = 0;
    size_t n = 0;
};

struct mpc_hip_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    /* ring of event quadruples: 0 = K1 begins, 1 = K1 ends, 2 = K2 begins, 3 = last kernel ends.  The candidate-set
     * runs ATTACH them to the two dispatches (hipExtLaunchKernelGGL: the kernels' own begin/end timestamps, no
     * extra packets in the queue); only every timing_period-th run carries them (0 = none), timed[] says which. */
    std::vector<hipEvent_t> ev;
    std::vector<uint8_t> timed;
    uint64_t runs = 0, timing_base = 0;
    uint32_t timing_period = 1;

    DevProblem *d_probs = nullptr, *h_probs = nullptr;          /* h_probs pinned */
    uint32_t cap_probs = 0, n_probs = 0;
    DevResult *d_results = nullptr, *h_results = nullptr;       /* h_results pinned */
    uint64_t *d_winners = nullptr;
    uint32_t *d_list_count = nullptr;                           /* per-problem scratch word (tree trace cursor) */
    uint64_t *d_pkey = nullptr, *d_pord = nullptr;              /* K1 workgroup triples */
    uint32_t *d_pcnt = nullptr;
    double *d_snap = nullptr;                                   /* per-level snapshots */
    size_t cap_snap = 0;
    size_t cap_partials = 0;
    double *d_keys = nullptr;
    size_t cap_keys = 0;
    double *d_controls = nullptr;                               /* 2 * MPC_MAX_DEPTH, mpc_hip_rollout */

    uint64_t last_count = 0, last_first = 0;
    int last_precision = 0, last_selection = 0, last_seeded = 0;
    bool ran = false;
    std::vector<int32_t> depth, nsub;

    /* tree search */
    std::vector<TreeLevelBuf> tree;
    uint32_t *d_scan_tmp = nullptr; size_t cap_scan_tmp = 0;
    uint64_t *h_pair = nullptr;                                  /* pinned, 4 uint64 */
};

int mpc_use_device(const mpc_hip_handle *h);
int mpc_ensure_problem_capacity(mpc_hip_handle *h, uint32_t n);
int mpc_ensure_partials(mpc_hip_handle *h, size_t n);
int mpc_check_problem(const mpc_problem *p, int precision);
void mpc_pack_problem(const mpc_problem *p, uint64_t seed, DevProblem *d);
void mpc_tree_free(mpc_hip_handle *h);

/* ---- device helpers ----------------------------------------------------------------------------------------- */
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

/* block argmin: every thread of the block must call; result valid in thread 0 (all of wave 0).
 * red must hold 2 * (blockDim.x / 64) uint64 of LDS. */
__device__ inline void block_argmin(uint64_t &k, uint64_t &o, uint64_t *red)
{
    wave_argmin(k, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    if (nw == 1) return;
    __syncthreads();                      /* red may still be read from a previous use */
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

#endif
