/*
 * mpc_hip.h -- C-ABI of libmpc_hip.so, the MI355X (gfx950) implementation of the afb2001/mpc
 * trajectory-optimisation hot path: candidate control search -> VehicleState forward simulation with the
 * estimated current applied -> tracking cost against the reference trajectory -> argmin.
 *
 * This is the drop-in boundary of SURVEY.md section 8b.  The reference has no FFI: its hot path is the body
 * of Controller::mpc (src/controller.cpp:26-205) calling VehicleState::simulate (src/VehicleState.h:43-55)
 * and Controller::compareStates (src/controller.cpp:231-241).  A maintainer keeps controller.h unchanged and
 * replaces the body of Controller::mpc by the marshalling shown in INTEGRATION.md, which calls the entry
 * points below.  Every entry point names the reference lines it replaces.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs; no C++ types, no exceptions, no torch types.
 *   - every function returns an mpc_status (0 = ok); mpc_hip_status_string() gives the text, including the
 *     three std::runtime_error texts of controller.cpp:37,196,202 so the C++ wrapper can re-throw them.
 *   - pointers named h_* are host memory, d_* are device memory on the handle's GPU.  The library never
 *     keeps a caller pointer after the call returns.
 *   - a handle owns its device buffers and may be used by one thread at a time; Controller holds two
 *     (SURVEY.md section 7 hard part 7: up to two mpc() calls overlap).
 *   - headings are compass headings in radians (0 = north = +y, clockwise), as path_planner_common State.
 *   - there is NO CPU fallback: without a usable HIP device mpc_hip_create fails with MPC_ERR_NO_DEVICE.
 */
#ifndef MPC_HIP_H
#define MPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_HIP_ABI_VERSION 2          /* 2: mpc_result.libm_tie, mpc_tree_level.libm_tie, precision/selection arguments of combine_shards_exact */
#define MPC_MAX_DEPTH 64        /* search depth D in 1-s scoring levels (BASELINE configs use 20 and 30)   */
#define MPC_MAX_SUBSTEPS 64     /* Euler substeps per level S (11 reference-native, 20 for 0.05 s steps)    */

typedef enum {
    MPC_OK = 0,
    MPC_ERR_INVALID = 1,            /* bad argument (message via mpc_hip_last_error)                         */
    MPC_ERR_NO_DEVICE = 2,          /* no HIP device / device index out of range                            */
    MPC_ERR_HIP = 3,                /* a HIP runtime call failed                                            */
    MPC_ERR_EMPTY_PLAN = 4,         /* controller.cpp:37  "Cannot run MPC on an empty plan"                 */
    MPC_ERR_NO_ITERATION = 5,       /* controller.cpp:196 "MPC failed to complete a single iteration"       */
    MPC_ERR_NON_FINITE = 6,         /* controller.cpp:202 "MPC selected non-finite controls"                */
    MPC_ERR_STALE_TRAJECTORY = 7,   /* controller.cpp:30-32,83-86: trajectory number no longer valid        */
    MPC_ERR_TOO_LARGE = 8           /* exceeds MPC_MAX_DEPTH / MPC_MAX_SUBSTEPS / memory limits             */
} mpc_status;

typedef enum {
    /* "FP32 mode" of BASELINE.json: the fast kernel.  State is integrated in FP64 (MI355X runs FP64 vector
     * math at half the FP32 rate) with the model algebraically streamlined: heading carried as (sin, cos)
     * and advanced by a polynomial rotation, divisions by reciprocals, reciprocal-sqrt + Newton, FP32
     * atan2 on an FP64 cross/dot pair.  Contract: per-candidate cost within 1e-5 relative of the reference (relative
     * to max(|cost|, 1 % of the sum of the weights): a cost that is one cancelling difference has no relative accuracy)
     * (measured: far tighter), and the SAME chosen candidate, guaranteed by re-evaluating every
     * candidate within MPC_REFINE_REL of the minimum with the strict kernel. */
    MPC_PREC_FAST = 0,
    /* "FP64 validation mode": the reference's operation order statement by statement, no FMA contraction,
     * IEEE sqrt, correctly rounded divisions, < 1-ulp sin/cos/atan2/exp.  Contract: cost
     * within 1e-12 relative of the reference (measured 1.4e-14; relative to max(|cost|, 1 % of the weights' sum)), same
     * argmin -- up to candidates whose scores differ by an ulp or two, which resolve as ONE libm would and not
     * necessarily as the reference's own did (its sin/cos/atan2/exp are glibc's, chosen by CPU at load time). */
    MPC_PREC_F64 = 1
} mpc_precision;

#define MPC_REFINE_REL 1.0e-6
/* "Same chosen control" has one limit that no implementation can remove: two candidates whose reference-arithmetic
 * scores are closer than the arithmetic is reproducible are ordered by the last bits of sin/cos/atan2/exp, i.e. by WHICH
 * libm evaluated them (the reference's own choice between them depends on its glibc build: glibc picks FMA or non-FMA
 * variants by CPU).  The device's functions are < 1 ulp but not glibc's bits, and an ulp of atan2 is several ulps of a
 * score once the heading weight has multiplied it: a measured case (tools/fuzz_tree.py 2000 11, problem 326) has two
 * leaves 3 ulps apart on the device and 1 ulp apart THE OTHER WAY ROUND under this container's glibc.  Results therefore
 * carry a flag, libm_tie: set when the strict (reference-arithmetic) score of the runner-up lies within MPC_TIE_REL of
 * the winner's -- relative to max(score, 1 % of the weights' sum), like every tolerance of this header; 1e-12 is the
 * validation mode's own contract, inside which no order is promised.  A choice that differs from the reference's is then
 * auditable as libm-dependent instead of silently different.  (Exactly equal scores also set it; those resolve by
 * ordinal, as the reference's first-strict-minimum rule does.) */
#define MPC_TIE_REL 1.0e-12

typedef enum {
    MPC_SELECT_LAST_STEP = 0,       /* the reference's rule: compare the LAST level's score, controller.cpp:155-166 */
    MPC_SELECT_CUMULATIVE = 1       /* opt-in: compare the cumulative Score of controller.cpp:156            */
} mpc_selection;

/* One reference-trajectory checkpoint: DubinsPlan::sample at start.time + (i+1)*1 s (controller.cpp:96-98). */
typedef struct { double x, y, heading, speed; } mpc_checkpoint;

/* State after one level, as MpcState (controller.h:28-34) + VehicleState members (VehicleState.h:68-80). */
typedef struct {
    double x, y, heading, speed;    /* State; time is start_time + level (candidate independent)             */
    double rpm, course_made_good, speed_over_ground;
    double rudder, throttle;        /* MpcState::LastRudder / LastThrottle: the control that led INTO this state */
    double score;                   /* MpcState::Score: cumulative                                           */
} mpc_level_state;

/* One MPC problem = everything Controller::mpc reads (controller.cpp:40-45, 59-61, 70-73, 96-98, 150, 231-241). */
typedef struct {
    double start_x, start_y, start_heading, start_speed;   /* startState (rpm derived, VehicleState.h:21)    */
    double last_rudder, last_throttle;                      /* m_LastRudder / m_LastThrottle, controller.cpp:71-72 */
    double current_x, current_y;                            /* m_DisturbanceEstimate, controller.cpp:150       */
    double w_distance, w_heading, w_speed;                  /* controller.cpp:240; must be finite and >= 0 (the
                                                             * ranges of cfg/mpc.cfg:11-15), else MPC_ERR_INVALID */
    double rudder_granularity, throttle_granularity;        /* controller.cpp:43-44 (tree search only)         */
    double min_rudder, mid_rudder, max_rudder;              /* controller.h:146                                */
    double min_throttle, max_throttle;                      /* controller.h:147                                */
    double tolerance;                                       /* c_Tolerance, controller.h:231                   */
    int32_t depth;                                          /* D: levels = checkpoints supplied                */
    int32_t n_substeps;                                     /* S: Euler substeps per level                     */
    mpc_checkpoint checkpoints[MPC_MAX_DEPTH];              /* [depth]                                         */
    double substeps[MPC_MAX_SUBSTEPS];                      /* [n_substeps] lengths from VehicleState.h:49-53  */
} mpc_problem;

typedef struct {
    int32_t  status;                /* mpc_status of this problem                                            */
    int32_t  levels;                /* D of the returned trajectory (tree search: deepest COMPLETED level)   */
    uint64_t best_ordinal;          /* winning candidate ordinal / DFS leaf ordinal                          */
    double   best_last;             /* its last-level score (controller.cpp:160 minScore)                    */
    double   best_cumulative;       /* its cumulative Score                                                  */
    double   first_rudder, first_throttle;   /* overallBestTrajectory[1].LastRudder/LastThrottle, :204        */
    uint64_t evaluated;             /* candidates (leaves) scored; tree search: nodes of the tree (see there)  */
    uint64_t rollout_steps;         /* estimate() calls of this solve (the unit of the steps/s metric); tree search:
                                     * those of the reference's walk over the same tree                         */
    uint32_t refined;               /* candidates re-evaluated by the strict kernel                          */
    uint32_t libm_tie;              /* 1: the strict runner-up is within MPC_TIE_REL of the winner (see above); tree
                                     * search: of the deepest completed level.  Within one GPU's candidates / leaves.   */
    float    kernel_ms;             /* HIP-event time of the dominant rollout kernel(s) of this call         */
    float    total_ms;              /* HIP-event time from first launch to last kernel end                   */
    mpc_level_state trajectory[MPC_MAX_DEPTH + 1];   /* [levels+1]; [0] = start (controller.cpp:185-188)      */
} mpc_result;


/* ---- library ------------------------------------------------------------------------------------------ */
int         mpc_hip_abi_version(void);
const char *mpc_hip_status_string(int status);
/* text of the last failure on this thread (HIP error string, argument that was rejected) */
const char *mpc_hip_last_error(void);

/* Fill the fields that have reference defaults (cfg/mpc.cfg:8-15, controller.h:146-147,231); zero the rest. */
void        mpc_hip_default_problem(mpc_problem *p);
/* VehicleState.h:49-53: the substep lengths simulate() uses for duration t, written to out[cap].
 * Returns the count (11 for t = 1.0: ten of 0.1 and one of 1.39e-16). */
int         mpc_hip_substep_schedule(double t, double *out, int cap);

typedef struct mpc_hip_handle mpc_hip_handle;

/* The boat: the 11 model parameters of VehicleState.h:86-96 (compile-time constants in the reference, with the note
 * "should allow setting of boat characteristics", :84).  What estimate() derives from them (:98-101: max prop speed,
 * max force, prop and drag coefficients) is computed by the library. */
typedef struct {
    double idle_rpm, max_rpm, max_rpm_change_rate;          /* VehicleState.h:86-88 */
    double prop_ratio, prop_pitch;                          /* :89-90 */
    double max_rudder_angle, rudder_coefficient, rudder_distance;   /* :91-93 (radians; the reference: pi/6) */
    double mass, max_power, max_speed;                      /* :94-96 */
} mpc_model;
/* the reference's own values */
void        mpc_hip_default_model(mpc_model *m);
/* Replaces the compile-time constants of VehicleState.h:86-96 for this handle: problem p of every later upload /
 * solve / fleet current estimate uses models[p % n_models] (n_models == 1: one hull for everything; n_models == the
 * fleet's size: a hull per vessel).  A new handle holds the reference's hull, with which every result is bit for bit
 * what it was before this call existed.  Resident problems are dropped (upload again).  Parameters must be finite,
 * max_rpm > idle_rpm, and change rate, prop ratio and pitch, rudder distance, mass, power and speed positive
 * (MPC_ERR_INVALID otherwise).  MPC_PREC_F64 takes any such hull.  MPC_PREC_FAST additionally needs what its
 * approximations assume -- max_rudder_angle <= pi/4 and at most 0.06 rad of yaw per 0.1-s substep (the reference's
 * hull: 0.049) -- and runs with a hull outside that fail with MPC_ERR_INVALID naming MPC_PREC_F64. */
int         mpc_hip_set_model(mpc_hip_handle *h, const mpc_model *models, uint32_t n_models);

/* Create a handle on HIP device `device`.  stream: a hipStream_t to launch on, or NULL for a stream the
 * handle creates.  Replaces nothing in the reference (it has no device). */
int         mpc_hip_create(int device, void *stream, mpc_hip_handle **out);
void        mpc_hip_destroy(mpc_hip_handle *h);
int         mpc_hip_synchronize(mpc_hip_handle *h);

/* ---- explicit candidate sets (BASELINE configs 2, 3) ----------------------------------------------------
 * Replaces, per candidate i, the chain of controller.cpp:149-156 over `depth` levels:
 *   state = state.simulate(r[l][i], t[l][i], level, current)   -- here as n_substeps estimate() calls
 *   score = compareStates(checkpoint[l], state);  Score += score
 * and the first-strict-minimum selection of controller.cpp:158-166.
 * d_rudders / d_throttles: device, FP32, level-major [depth][n] (SoA: lane i reads element l*n + i).
 * d_cost_last / d_cost_cum: device, FP64 [n], may be NULL.  h_result: host.
 * A set in which NO candidate has a finite score selects nothing, as nothing is ever below the reference's initial
 * minScore = DBL_MAX (controller.cpp:81,160): result status MPC_ERR_NO_ITERATION, best_ordinal = UINT64_MAX, packed
 * winner (+inf bits, UINT64_MAX).  The one-problem calls return that status; the batched call returns MPC_OK and
 * reports per problem in h_results[p].status. */
int         mpc_hip_eval_candidates(mpc_hip_handle *h, const mpc_problem *h_problem,
                                    const float *d_rudders, const float *d_throttles, uint64_t n,
                                    int precision, int selection,
                                    double *d_cost_last, double *d_cost_cum, mpc_result *h_result);

/* ---- seeded candidate sets generated on the device (BASELINE configs 4, 5) -------------------------------
 * Same computation; the controls of candidate `ordinal` at level l come from include/mpc_candidates.h
 * (seed, ordinal, l).  Evaluates ordinals first_ordinal .. first_ordinal + count - 1 (this GPU's shard of a
 * larger set: SURVEY.md section 8e); best_ordinal is the GLOBAL ordinal.  Costs are indexed by
 * ordinal - first_ordinal. */
int         mpc_hip_eval_seeded(mpc_hip_handle *h, const mpc_problem *h_problem, uint64_t seed,
                                uint64_t first_ordinal, uint64_t count, int precision, int selection,
                                double *d_cost_last, double *d_cost_cum, mpc_result *h_result);

/* Batched form (config 5): n_problems independent problems, each with its own seed and `count` candidates
 * (ordinals 0..count-1).  h_results[n_problems].  d_cost_last/d_cost_cum: [n_problems][count] or NULL. */
int         mpc_hip_eval_seeded_batch(mpc_hip_handle *h, const mpc_problem *h_problems, const uint64_t *h_seeds,
                                      uint32_t n_problems, uint64_t count, int precision, int selection,
                                      double *d_cost_last, double *d_cost_cum, mpc_result *h_results);

/* Split form of the batched call for callers that keep problems resident (bench, fleet simulation):
 * upload once, then run any number of solves with no host->device traffic but the launch itself. */
int         mpc_hip_upload_problems(mpc_hip_handle *h, const mpc_problem *h_problems, const uint64_t *h_seeds,
                                    uint32_t n_problems);
int         mpc_hip_run_seeded(mpc_hip_handle *h, uint64_t first_ordinal, uint64_t count, int precision,
                               int selection, double *d_cost_last, double *d_cost_cum);
int         mpc_hip_run_candidates(mpc_hip_handle *h, const float *d_rudders, const float *d_throttles,
                                   uint64_t n, int precision, int selection,
                                   double *d_cost_last, double *d_cost_cum);
/* mpc_hip_run_seeded for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE (it has no device): this entry
 * point exists for callers that must not reach hipMalloc / hipFree -- tight serving loops, harnesses that forbid
 * allocation, code that wants its device footprint fixed before it starts capturing HIP graphs (what the call guarantees
 * is the absence of allocation, asserted by tests/test_zz_no_alloc.py through hipMemGetInfo; launching it INSIDE an active
 * stream capture has not been exercised: it sets the device and attaches events to its dispatches when timing is on).  Same arguments, same kernels, same results in the same places
 * (mpc_hip_fetch_results, device winners, winner sink, d_cost_*) as mpc_hip_run_seeded, bit for bit; what differs is
 * where the working memory lives: the candidates' keys and the rollout workgroups' (key, ordinal, near-tie count)
 * triples are kept in a fixed device array inside the library (8.6 MB, held by every GPU the library has created a
 * handle on until the process ends) instead of grow-on-demand buffers of the handle, and no per-level snapshots are
 * written (the winner's trajectory is rolled out once more on one lane).  The call performs no hipMalloc, hipFree or
 * other allocating call; problems must have been uploaded before (mpc_hip_upload_problems allocates).
 * Limits: resident problems x count <= MPC_STATIC_MAX_CANDIDATES and problems x rollout workgroups (64 candidates
 * each below 2^20 candidates in total, 256 at it) <= MPC_STATIC_MAX_WORKGROUPS; beyond -> MPC_ERR_TOO_LARGE, nothing
 * launched.  All handles on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads)
 * race -- serialise them; nothing in the array needs clearing between calls (every word read is written first). */
#define MPC_STATIC_MAX_CANDIDATES (1u << 20)
#define MPC_STATIC_MAX_WORKGROUPS (1u << 15)
int         mpc_hip_run_seeded_static(mpc_hip_handle *h, uint64_t first_ordinal, uint64_t count, int precision,
                                      int selection, double *d_cost_last, double *d_cost_cum);
/* Reference trajectories sampled on the device.  Replaces, for batches, the host loop of
 * controller.cpp:96-98 -- checkpoint i = referenceTrajectory.sample(start time + (i+1) * 1 s) -- with one kernel
 * (a lane per problem and level) that writes the checkpoints into the resident problem records; the problems' own
 * checkpoints[] are ignored, everything else of mpc_upload_problems applies.  mpc_dubins = dubins.h DubinsPath +
 * DubinsWrapper::fill(path, speed, startTime) (mpc_node.cpp:245-248); DubinsPlan::sample as SURVEY.md 8c defines it
 * (external to the reference repository: parity unpinned).  Empty plan: MPC_ERR_EMPTY_PLAN (controller.cpp:37). */
#define MPC_MAX_PLAN_SEGMENTS 8
typedef struct {
    double qi[3], param[3], rho;      /* start pose (x, y, yaw), segment lengths in units of rho, turn radius     */
    int32_t type, reserved;           /* DubinsPathType: LSL, LSR, RSL, RSR, RLR, LRL = 0..5                        */
    double speed, start_time, end_time;
} mpc_dubins;
typedef struct { int32_t n_segments, reserved; mpc_dubins segments[MPC_MAX_PLAN_SEGMENTS]; } mpc_plan;
int         mpc_hip_upload_problems_from_plans(mpc_hip_handle *h, const mpc_problem *h_problems, const uint64_t *h_seeds,
                                               const mpc_plan *h_plans, const double *h_start_times,
                                               uint32_t n_problems);
/* the checkpoints of one resident problem as the kernels see them (h_out[depth]); for display and tests */
int         mpc_hip_fetch_checkpoints(mpc_hip_handle *h, uint32_t problem, mpc_checkpoint *h_out);
/* Batched CurrentEstimator::getCurrent (reference CurrentEstimator.cpp:15-41; its caller mpc_node.cpp:190-192 runs it
 * before every solve): vessel p's history is h_samples[p * stride .. + h_counts[p]), oldest first, each sample the
 * State at an updateEstimate call and the controls in effect (CurrentEstimator.cpp:11); h_actual[3p..] = x, y, time of
 * the state the estimate is asked for.  The replay runs on the device, a lane per vessel, in the strict FP64 model
 * with no current, and (actual - reckoned) / (query time - oldest sample time) is the estimate; (0, 0) for fewer
 * than MPC_CURRENT_MIN_SAMPLES samples or a query not after the oldest sample, as the reference returns.  (The
 * reference's 20-update skip and 600-sample cap are updateEstimate's, i.e. the caller's history keeping.)
 * h_currents_out (optional): [n][2] = cx, cy.  apply != 0: n must equal the resident problems and estimate p becomes
 * problem p's current for the following runs (what controller.cpp:398-400 stores per solve), rewritten in the
 * resident records on the device.  Synchronises the stream. */
#define MPC_CURRENT_MIN_SAMPLES 10        /* CurrentEstimator.cpp:18 */
#define MPC_CURRENT_MAX_SAMPLES 600       /* CurrentEstimator.h:82 c_BufferSize */
#define MPC_CURRENT_MAX_SPAN 3600.0       /* seconds of replay per vessel this library accepts (the reference keeps 60 s) */
typedef struct { double x, y, heading, speed, time, rudder, throttle; } mpc_history_sample;
int         mpc_hip_estimate_currents(mpc_hip_handle *h, const mpc_history_sample *h_samples, const uint32_t *h_counts,
                                      uint32_t stride, const double *h_actual, uint32_t n, int apply,
                                      double *h_currents_out);
/* The same estimate from DEVICE-RESIDENT histories: CurrentEstimator's own state for a fleet kept in HBM, so that a
 * tick uploads one sample per vessel (56 B) instead of every vessel's whole history (138 MB for 4096 x 600).
 *   mpc_hip_history_reset(h, n)        resetEstimate (CurrentEstimator.h:43-47) for n vessels: empty rings, 20 skips left
 *   mpc_hip_history_update(h, s, n)    updateEstimate (CurrentEstimator.cpp:5-13) for every vessel, one kernel:
 *                                      h_samples[p] = the State at this update + the controls in effect; the first 20
 *                                      updates after a reset are dropped, the ring keeps the newest 600
 *   mpc_hip_estimate_currents_resident getCurrent (:15-41) from the rings; arguments as mpc_hip_estimate_currents
 * The rings are field-major SoA [7][600][n]: the 64 lanes of a wave read 512 contiguous bytes per field (the uploaded
 * form is array-of-structs per vessel, lanes 33.6 KB apart).  Same arithmetic, same bits as the uploaded form. */
int         mpc_hip_history_reset(mpc_hip_handle *h, uint32_t n_vessels);
int         mpc_hip_history_update(mpc_hip_handle *h, const mpc_history_sample *h_samples, uint32_t n);
int         mpc_hip_estimate_currents_resident(mpc_hip_handle *h, const double *h_actual, uint32_t n, int apply,
                                               double *h_currents_out);
/* copy the results of the last run (n_problems of them) to the host; synchronises the stream */
int         mpc_hip_fetch_results(mpc_hip_handle *h, mpc_result *h_results, uint32_t n_problems);
/* device address of the per-problem packed winners of the last run: for problem p two uint64,
 * [2p] = bits of the winning (strict) score, [2p+1] = winning global ordinal.  Non-negative doubles order
 * like their bit patterns, so an RCCL min-allreduce over [2p] followed by one over the ordinals of the ranks
 * that hold the minimum combines shards. */
int         mpc_hip_device_winners(mpc_hip_handle *h, uint64_t **d_winners);

/* copy the packed winners (2 uint64 per problem) to caller-owned device memory, on the handle's stream: the
 * form a host that owns its communicator buffers (torch.distributed / RCCL) uses for the min-allreduce */
int         mpc_hip_copy_winners(mpc_hip_handle *h, uint64_t *d_dst);
/* The copy without the copy: from the next run on, the finish kernel ALSO writes the packed winners (2 uint64 per
 * problem) to d_dst -- caller-owned device memory that must stay valid until those runs have completed; NULL turns
 * it off.  A stream of solves that batches its winners for one collective (ShardCombiner) points this at the next row
 * of its batch buffer before every run instead of enqueuing a 16-byte device copy after it (~3 us per solve). */
int         mpc_hip_set_winner_sink(mpc_hip_handle *h, uint64_t *d_dst);
/* Combine the winners of `n_shards` shards after ONE all-gather of the copy_winners records:
 * d_gathered[shard][problem][2] -> d_out[problem][2] = (smallest score bits, lowest GLOBAL ordinal among the shards
 * that hold it), global ordinal = gathered ordinal + shard * ordinal_stride (0 if the shards were run with their
 * first_ordinal).  One tiny kernel on the handle's stream; the reference's first-strict-minimum rule
 * (controller.cpp:158-166) across GPUs.  No counterpart in the reference (it has one device: none). */
int         mpc_hip_combine_shards(mpc_hip_handle *h, const uint64_t *d_gathered, uint32_t n_shards,
                                   uint32_t n_problems, uint64_t ordinal_stride, uint64_t *d_out);
/* A winner record WITH its controls: [0] score bits, [1] ordinal, then (rudder, throttle) of the winner at every level as
 * doubles: MPC_WINNER_RECORD_WORDS(depth) uint64 per problem, depth >= the deepest resident problem.  The first form
 * packs the records of the last run into caller memory (on the stream); the second makes every later run's finish
 * kernel write its record there itself (NULL: off; replaces a plain winner sink and vice versa), as
 * mpc_hip_set_winner_sink does for the two-word form. */
#define MPC_WINNER_RECORD_WORDS(depth) (2 + 2 * (depth))
int         mpc_hip_copy_winner_records(mpc_hip_handle *h, uint64_t *d_dst, uint32_t depth);
int         mpc_hip_set_winner_record_sink(mpc_hip_handle *h, uint64_t *d_dst, uint32_t depth);
/* The EXACT cross-shard decision for fast-mode runs, after ONE all-gather of the winner records.  Inside a shard,
 * candidates within MPC_REFINE_REL of the shard's minimum are already decided by the strict model; ACROSS shards the
 * records carry fast scores (~1.2e-7 relative), which cannot order two shards that close.  d_gathered[shard][row]
 * [MPC_WINNER_RECORD_WORDS(depth)], n_rows = solves in the batch x resident problems (row % problems = problem).  For
 * every row in which more than one shard lies within MPC_REFINE_REL of the row's minimum, those shards' winners are
 * evaluated once more with the strict model FROM THE CONTROLS IN THEIR RECORDS -- so every rank can do it for every
 * shard, all ranks decide alike, and nothing but the gathered buffer and the resident problems is read -- and
 * d_out[row] = (score bits, global ordinal = gathered ordinal + shard * ordinal_stride) of the lexicographic minimum:
 * the reference's first strict minimum (controller.cpp:158-166) across GPUs, because the true minimum always lies
 * inside the near set and every record outside it is larger than every strict score inside it.  precision and
 * selection are those of the runs that PRODUCED the records (they say how to read them, whatever the handle ran last):
 * MPC_PREC_F64 records already carry the reference's arithmetic and reduce to mpc_hip_combine_shards; selection names
 * the score a re-scored winner is compared by.  One small kernel on the handle's stream (a wave per row, a lane per shard:
 * n_shards <= 64).  h_tied_rows (optional; synchronises): rows that had a cross-shard near-tie since the handle was
 * created (d_gathered == NULL: only read it).  No counterpart in the reference. */
int         mpc_hip_combine_shards_exact(mpc_hip_handle *h, const uint64_t *d_gathered, uint32_t n_shards, uint32_t n_rows,
                                         uint32_t depth, uint64_t ordinal_stride, int precision, int selection,
                                         uint64_t *d_out, uint32_t *h_tied_rows);
/* Streams of solves: the deferred finish.  A run is two kernels -- the rollouts with their workgroup argmins, then one
 * small workgroup per problem that reduces them, decides near-ties with the strict model and writes winners, result
 * and trajectory -- and for a 65,536-candidate solve the second one, with its dispatch boundary, is 15 % of the time.
 * With the pipeline on, mpc_hip_run_* enqueues only the rollouts and leaves that finish PENDING; the next run's rollout
 * launch carries it as one extra workgroup per problem, beside a thousand workgroups that do not wait for it (the
 * working memory between the two kernels is double-buffered), so a stream costs one dispatch per solve.  Results of
 * run n therefore appear on the stream with run n+1 -- or at once with any call that needs them, each of which first
 * launches a pending finish on its own: mpc_hip_fetch_results, mpc_hip_copy_winners, mpc_hip_device_winners,
 * mpc_hip_timing, mpc_hip_synchronize, mpc_hip_upload_problems*, mpc_hip_set_model, mpc_hip_estimate_currents,
 * mpc_hip_set_pipeline(h, 0), and mpc_hip_flush, for callers that consume the winner sink on the stream (flush before
 * the collective that reads it).  The winner sink in force when a run was ENQUEUED is the one its finish writes.
 * What a DEFERRED finish still reads of the caller's memory: the explicit control arrays of its run (d_rudders /
 * d_throttles: the strict decision among near-tied candidates and the winner's trajectory and record take the controls
 * from there) and the winner sink.  They must therefore stay UNCHANGED until the next run has been enqueued or
 * mpc_hip_flush has been called -- at level 1 already, not only at level 2: a sampling loop that rewrites its control
 * buffer in place between runs calls mpc_hip_flush first (with the pipeline off nothing is deferred and nothing is kept).
 * Results are bit for bit those of the two-kernel form.  The one-call forms (mpc_hip_eval_*) fetch and therefore
 * never defer; runs that have no fused variant (strict precision, non-uniform schedules, >= 2^20 candidates) finish
 * the pending solve with a launch of its own first.  Off by default.  No counterpart in the reference.
 * Levels 2, 3, 4 (on == K) add K RUNS PER LAUNCH: a 65,536-candidate set is one wave per SIMD, and a lone wave issues
 * an instruction every ~6.2 cycles where the SIMD could take one every 4 -- the second wave can only come from another
 * solve -- and every launch of a stream costs ~10 us whatever it holds (the boundary between dependent dispatches, the
 * workgroups' prologue, the tail), which a launch of four solves pays once.  A run that can share a launch (MPC_PREC_FAST,
 * uniform substeps, fewer than 2^20 candidates) is only PREPARED by its call; the K-th such run sends the whole group to
 * the GPU in one dispatch that also carries the finish workgroups of the group before (level 2: +10-15 % on a stream of
 * 65,536-candidate solves, level 4: a further +6 %).  What changes for the caller: the device arrays a run reads and
 * writes (controls, cost outputs, winner sink) are used from the stream position of the run that COMPLETES its group -- or
 * of the call that asks for results, which sends an incomplete group as it is -- so they must stay as they are until
 * then; everything else is as at level 1, results bit for bit. */
int         mpc_hip_set_pipeline(mpc_hip_handle *h, int on);
int         mpc_hip_flush(mpc_hip_handle *h);
/* Which runs carry HIP events: every `period`-th run of the handle (1 = all, the default; 0 = none).  The events
 * ride ON the run's two dispatches (the kernels' own begin/end timestamps), but a dispatch with events still
 * costs ~5 us more than one without, so throughput callers sample (bench.py: every 8th) or switch them off.
 * A run without events reports kernel_ms = total_ms = 0 in its mpc_result.  No counterpart in the reference. */
int         mpc_hip_set_timing(mpc_hip_handle *h, uint32_t period);
/* HIP-event times summed over the TIMED runs since the last reset (at most the last 1024 runs; *runs = how many
 * were timed): kernel_ms_sum covers the rollout kernel K1 alone (its begin to its end), total_ms_sum K1's begin
 * to the last kernel's end.  Synchronises the stream. */
int         mpc_hip_timing(mpc_hip_handle *h, int reset, uint64_t *runs, double *kernel_ms_sum, double *total_ms_sum);

/* ---- the reference's own candidate set: limited-branching tree (BASELINE config 1) ------------------------
 * Replaces controller.cpp:80-189: iterative-deepening DFS over the control sets of :121-146, evaluated here
 * level-synchronously (children of all nodes of a level in DFS order, so a leaf's index IS its DFS ordinal);
 * every deepening iteration's argmin over last-step scores falls out of one expansion.
 * max_levels = maxIterations of controller.cpp:61 (<= problem depth).  now(ctx) / end_time reproduce the
 * deadline of :101-103 between levels (now == NULL: frozen clock, run all levels).  Levels 1-4 (at most 30^4 nodes
 * whatever the problem) are expanded as one batch without a host round trip -- two launches per level, sizes read on the
 * device, one read-back at the end: depth 4 in 0.11 ms instead of 0.27 -- so the clock is polled once before and once
 * after that batch, then before and after every further level.  If the clock has passed end_time after the batch, the
 * levels whose reduction completed before it still count: their completion times are now(ctx) minus the difference
 * of the GPU's own 100-MHz clock stamps, i.e. the caller's clock is taken to run in seconds (the reference's does:
 * controller.cpp:19-24).  max_nodes bounds the
 * widest level (a level that does not fit ends the search like a deadline).  0 = no budget: levels are stored up to
 * the library's cap (2^28 nodes, 27 GB) and the first level beyond it -- depth 7 with the default control sets, 3e8
 * leaves -- is still expanded, without stores, as the final one.
 * h_level_best: optional [max_levels] per-level winners (ordinal + scores) as the deepening loop saw them.
 * precision: MPC_PREC_F64 expands with the reference's arithmetic.  MPC_PREC_FAST expands with the streamlined model
 * and lets the reference's arithmetic decide: every leaf within MPC_REFINE_REL of a level's minimum is re-evaluated
 * along its path from the root with the strict model, and the returned scores and trajectory are the strict
 * path's -- the same leaf and the same bits as MPC_PREC_F64, ~1.4x sooner at depth 7.
 * The LAST level expanded is searched, not enumerated (both precisions): all three terms of compareStates are >= 0, so
 * a leaf farther from its checkpoint than log(best score so far / w_distance) cannot win, and a node that cannot bring
 * any child that near within one level -- a bound on how far and in which directions the hull can travel in a level,
 * derived from the model's own recurrence (VehicleState.h:111-148) and the hull's parameters -- is not simulated at all
 * (97 % of the depth-7 level of the reference's semicircle scenario), and when max_levels makes a level the last one with
 * children, those of its nodes that can matter neither to its own record nor to the final level's get only their place in
 * it (their child count) -- two-level reach against two scores found by a greedy descent.  What is returned is what the full enumeration
 * returns: the level's first strict minimum in DFS order, its ordinal, scores, trajectory and libm_tie flag;
 * mpc_result.evaluated / rollout_steps and mpc_tree_level.n_leaves count the tree's nodes as the reference walks them,
 * whether or not this library had to simulate them. */
typedef double (*mpc_time_fn)(void *ctx);
typedef struct {
    uint64_t n_leaves, best_leaf; double best_score, best_cumulative;
    uint32_t libm_tie, reserved;    /* libm_tie: this level's strict runner-up is within MPC_TIE_REL of its winner */
} mpc_tree_level;
int         mpc_hip_solve_tree(mpc_hip_handle *h, const mpc_problem *h_problem, int max_levels,
                               int precision, mpc_time_fn now, void *now_ctx, double end_time,
                               uint64_t max_nodes, mpc_tree_level *h_level_best, mpc_result *h_result);

/* One GPU's share of a tree search split over n_shards GPUs.  Every shard expands levels
 * 1..split_level in full (<= 30^split_level nodes: redundant and cheap); from there shard s keeps the s-th contiguous
 * slice of the split level's nodes and expands only their subtrees, so DFS order is preserved within and across
 * shards.  h_level_best[l] for l < split_level is the full level's winner (the same on every shard); for deeper levels
 * n_leaves counts this shard's leaves and best_leaf is LOCAL to them: the global DFS ordinal is best_leaf + the leaves
 * of the lower shards at that level, and the level's winner is the (score, global ordinal) minimum over the shards
 * (mpc_amd/distributed.py: combine_tree_levels).  h_result is this shard's own best path.  split_level must be below
 * max_levels and that level must have at least n_shards nodes.  No counterpart in the reference (one device: none). */
int         mpc_hip_solve_tree_shard(mpc_hip_handle *h, const mpc_problem *h_problem, int max_levels,
                                     int precision, mpc_time_fn now, void *now_ctx, double end_time,
                                     uint64_t max_nodes, int split_level, uint32_t shard, uint32_t n_shards,
                                     mpc_tree_level *h_level_best, mpc_result *h_result);

/* The reference's deepening loop (controller.cpp:80) has no depth limit but its clock; one expansion has two: 32-bit leaf
 * ordinals and the storage cap.  With the default control sets they end mpc_hip_solve_tree at depth 7 (3.0e8 leaves) --
 * in ~1.5 ms of the 100 ms the planner grants.  This entry point goes on from there: each further level is searched in
 * pieces -- slices of the deepest level that is stored in full, searched one after the other on this handle without
 * expanding anything above them again, as many as make a piece's leaves fit -- and the pieces' winners are combined as
 * the shards' are: the smallest (strict score, global DFS ordinal).  Depth 8 (5.2e9 leaves) takes 7-11 ms
 * (MPC_PREC_FAST) on one MI355X, depth 9 (8.8e10) 87-126 ms.  A level counts only if all its pieces finished before
 * end_time (controller.cpp:101-103: an unfinished iteration is discarded); h_level_best / h_result then stay the
 * previous level's -- and the level is given up as soon as its remaining pieces, at the pace of those done, would end
 * after it.  now == NULL runs to max_levels whatever that takes.  Results are those of one search of the same depth:
 * same leaf, same scores, same trajectory; best_ordinal and h_level_best[].best_leaf are 64-bit DFS ordinals of the
 * whole level.  Replaces the same lines of the reference as mpc_hip_solve_tree (controller.cpp:80-189). */
int         mpc_hip_solve_tree_deep(mpc_hip_handle *h, const mpc_problem *h_problem, int max_levels,
                                    int precision, mpc_time_fn now, void *now_ctx, double end_time,
                                    mpc_tree_level *h_level_best, mpc_result *h_result);

/* ---- single rollout (VehicleState::simulate itself, VehicleState.h:43-55) --------------------------------
 * Runs ONE control sequence (host arrays [depth]) through the strict kernel and returns the state after every
 * level.  Used by the host wrapper for getStateAfterCurrentControl-style calls and by the parity tests. */
int         mpc_hip_rollout(mpc_hip_handle *h, const mpc_problem *h_problem, const double *h_rudders,
                            const double *h_throttles, int precision, mpc_result *h_result);

#ifdef __cplusplus
}
#endif
#endif /* MPC_HIP_H */
