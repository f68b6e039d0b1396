/*
 * mpc_internal.h -- records and helpers shared by mpc_hip.hip (candidate sets) and mpc_tree.hip (tree search).
 * Not part of the ABI.
 */
#ifndef MPC_INTERNAL_H
#define MPC_INTERNAL_H

#include <hip/hip_runtime.h>

#include <stddef.h>
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
    int32_t head_uniform;        /* 1: substeps 0 .. nsub-2 equally long (nsub >= 2): every schedule simulate() itself makes;
                                  * 2: ... and the last one a negligible remainder (< 2^-40 of the others) */
    int32_t pad0; double pad1;   /* keeps cp 16-byte aligned for the staging vectors */
    /* How far a node can travel through the water in one level (mpc_pack_problem derives it from the hull and the schedule;
     * the tree search's final level leaves out parents none of whose children can come near enough to win, mpc_tree.hip):
     * a node with 0 <= v, v + reach_dv <= reach_vcap and |rpm| <= reach_rpm moves at most v * reach_t + reach_acc, and
     * while v + reach_dv <= reach_sqrtps (the propeller wash, not the hull's speed, is what the rudder works in) its
     * heading turns by at most D, given as reach_sd = sin D, reach_cd = cos D (D >= pi/2: reach_cd < 0, no statement).
     * reach_vcap < 0: no such bound for this hull / schedule. */
    double reach_dv, reach_t, reach_vcap, reach_rpm, reach_cx, reach_cy, reach_acc, reach_sqrtps, reach_sd, reach_cd;
    double reach_acc2, reach_pad;   /* the same path bound over TWO levels: v * 2 reach_t + reach_acc2 (speed bound v + 2 reach_dv, turn 2 D) */
    mpc::Model model;            /* this problem's hull (mpc_hip_set_model; VehicleState.h:86-101) */
    double cp[MPC_MAX_DEPTH * CP_STRIDE];
    double sub[MPC_MAX_SUBSTEPS * SUB_STRIDE];
};

struct DevResult {
    uint64_t key_bits, ordinal;
    double best_last, best_cum;
    uint32_t refined, overflow;
    uint32_t tie, pad;              /* tie: the strict runner-up lies within MPC_TIE_REL of the winner (mpc_result.libm_tie) */
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
    float *margin = nullptr;    /* how far beyond its reach the NEXT level's checkpoint lies from the node (node_margin, mpc_tree.hip) */
    size_t cap = 0;
    size_t n = 0;
    size_t first = 0;      /* multi-GPU shard: the level is the slice [first, first + n) of the full level */
    size_t full_n = 0;     /* the full level's nodes, as expanded (n after a slice is the slice's) */
};

/* device-side view of one TreeLevelBuf */
struct LevelPtrs {
    double *x, *y, *h, *v, *rpm, *cmg, *sog, *last_r, *last_t, *cum, *last_score;
    uint32_t *parent, *count, *offset;
    float *margin;
};

/* The near-tie window above a minimum score m: everything up to m (1 + rel) -- but never narrower than rel x 1 % of the
 * weights' sum.  With w_distance = 0 a minimum can be a lone |reference - actual| that cancels to 1e-4 of its operands;
 * the fast model's error is a fraction of the OPERANDS (tools/fuzz_random_problems.py: <= 2.9e-7 of max(|cost|, that
 * floor)), so a window measured against such an m alone would be narrower than the error it has to cover. */
__host__ __device__ inline double mpc_refine_limit(double m, double rel, double wd, double wh, double ws)
{
    const double floor = 0.01 * (wd + wh + ws);
    return m + (m > floor ? m : floor) * rel;
}

/* everything K2 needs, as one record: the standalone kernel takes it as its argument, and a rollout launch that also
 * finishes the PREVIOUS solve (FUSE, below) carries that solve's record next to its own arguments */
struct FinishArgs {
    const DevProblem *probs;
    const float *rud, *thr;
    uint64_t first, count;
    int selection, fast, uniform;
    double refine_rel;
    uint32_t nblk, k1_block;
    const uint64_t *pkey, *pord;
    const uint32_t *pcnt;
    const double *keys;
    double *snap_base;
    int snap_depth;
    uint64_t *winners, *sink;
    uint32_t sink_words;
    int snap_shift;                 /* snapshots hold every 2^shift-th level */
    DevResult *results;
};
/* how many runs one rollout launch can hold (mpc_hip_set_pipeline level = runs per launch, 2 .. MPC_GROUP_MAX) */
#define MPC_GROUP_MAX 4
/* what K1 needs of one run: rollout_group_kernel takes up to MPC_GROUP_MAX of these */
struct RunArgs {
    const float *rud, *thr;
    uint64_t first, count;
    int selection, snap_depth;
    double rel;
    double *cost_last, *cost_cum, *keys, *snap;
    uint64_t *pkey, *pord;
    uint32_t *pcnt;
    uint32_t nblk;
    int snap_shift;
};
/* a run that is prepared but not launched yet (level 2: it waits for a partner) */
struct StashedRun {
    RunArgs run;
    FinishArgs fin;
    bool seeded, timed;
    size_t lds;
    uint32_t n_probs, slot;
};
/* what K1 writes and K2 reads, twice (mpc_hip_set_pipeline): half 0's workgroup triples are h->d_pkey/d_pord/d_pcnt,
 * shared with the tree search */
struct WorkMem {
    double *d_keys = nullptr; size_t cap_keys = 0;
    double *d_snap = nullptr; size_t cap_snap = 0;
    uint64_t *d_pkey = nullptr; size_t cap_pkey = 0;
    uint64_t *d_pord = nullptr; size_t cap_pord = 0;
    uint32_t *d_pcnt = nullptr; size_t cap_pcnt = 0;
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
    uint64_t *d_winner_sink = nullptr;                          /* caller memory that runs also write the winners to */
    uint32_t winner_sink_words = 2;                             /* 2, or 2 + 2 * depth (records with controls) */
    uint32_t *d_list_count = nullptr;                           /* per-problem scratch word (tree trace cursor) */
    uint64_t *d_pkey = nullptr, *d_pord = nullptr;              /* K1 workgroup triples */
    uint64_t *d_pkey2 = nullptr;                                /* tree search: every expansion workgroup's runner-up key (libm_tie) */
    uint32_t *d_pcnt = nullptr;
    size_t cap_partials = 0;
    WorkMem wm[2 * MPC_GROUP_MAX];                              /* keys, snapshots (and the triples of all but wm[0]); level 1 uses two, level K >= 2 uses 2 K */
    /* deferred finish (mpc_hip.hip) */
    int pipeline = 0;                                           /* 0 off, 1 deferred finish, K >= 2: K runs per launch + deferred finish */
    bool pending = false, pend_seeded = false, pend_timed = false;
    /* level >= 2: the runs waiting for their group to fill, and the runs of the last launch whose finishes are pending */
    int n_stash = 0;
    StashedRun stash[MPC_GROUP_MAX];
    int n_pendg = 0;
    StashedRun pendg[MPC_GROUP_MAX];
    uint32_t wm_rr = 0;
    DevResult *d_results_alt = nullptr;                         /* where the OLDER of two finishes in one launch writes */
    uint64_t *d_winners_alt = nullptr;
    FinishArgs pend{};
    uint32_t pend_n_probs = 0, pend_slot = 0;
    int pend_half = 0;
    std::vector<uint8_t> total_from_k1;
    double *d_controls = nullptr;                               /* 2 * MPC_MAX_DEPTH, mpc_hip_rollout */
    void *d_static_scratch = nullptr;                           /* this device's copy of scratch[] (mpc_hip.hip), not owned */

    const float *last_rud = nullptr, *last_thr = nullptr;       /* the last run's explicit control arrays (caller memory) */
    uint32_t *d_tied_rows = nullptr;                            /* counter of mpc_hip_combine_shards_exact */
    uint64_t last_count = 0, last_first = 0;
    int last_precision = 0, last_selection = 0, last_seeded = 0;
    bool ran = false;
    std::vector<int32_t> depth, nsub;

    std::vector<mpc::Model> models;                              /* mpc_hip_set_model: problem p uses models[p % size] */
    std::vector<uint8_t> models_fast_ok;                         /* ... and whether the fast model's envelope covers it */
    mpc::Model *d_models = nullptr; size_t cap_models = 0;       /* device copy (fleet current estimation) */
    void *d_plans = nullptr; size_t cap_plans = 0;              /* mpc_plan[P] + start times (mpc_plan.hip) */
    void *d_hist = nullptr; size_t cap_hist = 0;                /* histories, counts, query states, estimates (mpc_current.hip) */
    void *d_hist_ring = nullptr, *d_hist_state = nullptr;       /* device-resident histories: ring[7][600][n], per-vessel state */
    void *d_hist_io = nullptr, *h_hist_io = nullptr;            /* one sample (or query + estimate) per vessel, device / pinned */
    uint32_t hist_n = 0;

    /* tree search */
    std::vector<TreeLevelBuf> tree;
    uint32_t *d_scan_tmp = nullptr; size_t cap_scan_tmp = 0;
    uint32_t *d_bpar = nullptr; size_t cap_bpar = 0;              /* first parent of every 256-child workgroup */
    double *d_tree_seed = nullptr;                                /* tree search: [0..1] seed bounds of the last two levels (ghost_select_kernel) */
    int tree_precision = -1;                                     /* the model whose node records the stored levels hold */
    int tree_full_upto = 0;                                      /* levels 0..this are stored in full (no slice, no ghosts) by the last search */
    bool tree_stopped_by_width = false;                          /* the last tree search ended because the next level does not fit one expansion */
    uint32_t *d_tile_flags = nullptr; size_t cap_tile_flags = 0; /* tree search: ghost flag per tile of the level above the final one */
    uint32_t *d_tile_list = nullptr; size_t cap_tile_list = 0;   /* tree search, final level: [0] count, [1..2] seed tiles, [4..] the short list */
    LevelPtrs *d_level_ptrs = nullptr;                           /* [MPC_MAX_DEPTH + 2], fast-mode path walks */
    std::vector<LevelPtrs> h_level_ptrs;                         /* its host image (copies are asynchronous) */
    uint64_t *h_pair = nullptr;                                  /* pinned, 4 uint64 */
    uint64_t *d_tree_meta = nullptr, *h_tree_meta = nullptr;     /* per-level records of the blind shallow levels (pinned host copy) */
};

int mpc_use_device(const mpc_hip_handle *h);
int mpc_ensure_problem_capacity(mpc_hip_handle *h, uint32_t n);
int mpc_ensure_partials(mpc_hip_handle *h, size_t n);
int mpc_flush_pending(mpc_hip_handle *h);
int mpc_check_problem(const mpc_problem *p, int precision);
void mpc_pack_problem(const mpc_problem *p, uint64_t seed, const mpc::Model &model, DevProblem *d);
const mpc::Model &mpc_model_of(const mpc_hip_handle *h, uint32_t problem);
int mpc_check_fast_model(const mpc_hip_handle *h, uint32_t n_problems);
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

/* The strict scores with key bits a <= b lie within MPC_TIE_REL of each other (mpc_result.libm_tie): the window of
 * mpc_refine_limit at the validation mode's own tolerance -- relative to max(score, 1 % of the weights' sum). */
__host__ __device__ inline double mpc_tie_limit(double m, double wd, double wh, double ws) { return mpc_refine_limit(m, MPC_TIE_REL, wd, wh, ws); }
__device__ inline bool keys_tie(uint64_t a, uint64_t b, double wd, double wh, double ws)
{
    return a < KEY_INF && b < KEY_INF && __longlong_as_double((long long)b) <= mpc_tie_limit(__longlong_as_double((long long)a), wd, wh, ws);
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

/* Stage checkpoints and substep schedule of one problem into LDS.  DevProblem keeps cp[] and sub[] adjacent and
 * 16-byte aligned, and the kernels keep cpL / subL adjacent in LDS, so staging is ONE contiguous copy of
 * STAGE_VEC 16-byte vectors, done in two halves: stage_problem_load() only ISSUES the loads (into registers, all in
 * flight together and together with whatever the caller loads next), stage_problem_store() writes them to LDS.
 * The whole arrays are copied whatever the problem's depth: the copy then depends on nothing but the kernel
 * arguments (no scalar load of P.depth in front of it), and a per-depth loop was one full memory latency per
 * trip -- the first profile of K1 showed ~11 serialized latencies (~10 us of a 50-us kernel) in its prologue. */
#define STAGE_VEC ((MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE) / 2)
#define STAGE_TRIPS ((STAGE_VEC + 63) / 64)          /* per-thread vectors at the smallest block (one wave) */
static_assert(offsetof(DevProblem, cp) % 16 == 0 && sizeof(DevProblem) % 16 == 0, "16-byte staging vectors");
static_assert(offsetof(DevProblem, sub) == offsetof(DevProblem, cp) + sizeof(double) * MPC_MAX_DEPTH * CP_STRIDE, "cp, sub adjacent");

struct StagedProblem { double2 v[STAGE_TRIPS]; };

__device__ inline void stage_problem_load(const DevProblem &P, StagedProblem &r)
{
    const double2 *src = reinterpret_cast<const double2 *>(P.cp);
#pragma unroll
    for (int t = 0; t < STAGE_TRIPS; t++) {
        /* clamped, not predicated: a load under a branch is waited for at the branch's end */
        const uint32_t idx = min(threadIdx.x + t * blockDim.x, (uint32_t)STAGE_VEC - 1u);
        r.v[t] = src[idx];
    }
}

/* lds = cpL (subL follows it); ends with the workgroup barrier */
__device__ inline void stage_problem_store(const StagedProblem &r, double *lds)
{
    double2 *dst = reinterpret_cast<double2 *>(lds);
#pragma unroll
    for (int t = 0; t < STAGE_TRIPS; t++) {
        /* clamped like the loads (surplus lanes rewrite the last vector with its own value): a predicated store
         * makes hipcc sink each load into the store's branch, i.e. one exposed latency per trip again */
        const uint32_t idx = min(threadIdx.x + t * blockDim.x, (uint32_t)STAGE_VEC - 1u);
        dst[idx] = r.v[t];
    }
    __syncthreads();
}

__device__ inline void stage_problem(const DevProblem &P, double *cpL, double *subL)
{
    StagedProblem r;
    stage_problem_load(P, r);
    stage_problem_store(r, cpL);
    (void)subL;
}

#endif
