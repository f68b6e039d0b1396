/*
 * mpc_hip.hip -- candidate-set kernels and C-ABI (include/mpc_hip.h) of the MI355X implementation of the
 * afb2001/mpc hot path.  (The reference's own limited-branching tree is in mpc_tree.hip.)
 *
 * Kernels: mpc_kernels.h (K1 rollout_kernel, K2 finish_kernel).  All launches go on the handle's stream and
 * nothing is allocated per call once the handle is warm.
 * Selection follows controller.cpp:158-166: strictly smaller LAST-level score wins, so the lowest ordinal wins
 * ties; MPC_SELECT_CUMULATIVE compares the cumulative Score instead.
 *
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (strict code must not contract; the fast code
 * spells its FMAs out).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include <hip/hip_ext.h>

#include "mpc_kernels.h"

static thread_local char g_err[512] = "";
void mpc_set_err(const char *fmt, const char *a, const char *b) { snprintf(g_err, sizeof g_err, fmt, a, b); }

/* ---- working memory of mpc_hip_run_seeded_static (the entry point that allocates nothing) -----------------------
 * What run_common() otherwise keeps in hipMalloc'ed, grow-on-demand buffers of the handle -- the candidates' keys
 * and the K1 workgroup triples -- laid out in ONE file-scope device array of fixed size (8.6 MB, held by every GPU
 * the library runs on until the process ends).  `scratch` is a host-side stand-in for the symbol, not a device
 * address: host code only ever uses the per-device address hipGetSymbolAddress returned (static_scratch_ptr(),
 * resolved in mpc_hip_create with that device current; each GPU holds its own copy at its own address). */
struct StaticScratch {
    double   keys[MPC_STATIC_MAX_CANDIDATES];
    uint64_t pkey[MPC_STATIC_MAX_WORKGROUPS];
    uint64_t pord[MPC_STATIC_MAX_WORKGROUPS];
    uint32_t pcnt[MPC_STATIC_MAX_WORKGROUPS];
};
static __device__ __align__(256) unsigned char scratch[sizeof(StaticScratch)];

#define MPC_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[MPC_MAX_DEVICES];          /* per device: the address of ITS copy of scratch[] */

/* device must be current */
static int static_scratch_ptr(int device, void **out)
{
    if (device < 0 || device >= MPC_MAX_DEVICES) { mpc_set_err("device index beyond MPC_MAX_DEVICES"); return MPC_ERR_NO_DEVICE; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) { mpc_set_err("hipGetSymbolAddress(scratch) returned a null address"); return MPC_ERR_HIP; }
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return MPC_OK;
}

/* ------------------------------------------------------------------------------------------------------ */
/* host side                                                                                               */
/* ------------------------------------------------------------------------------------------------------ */
static int flush_pending(mpc_hip_handle *h);
int mpc_use_device(const mpc_hip_handle *h)
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
    if (!out) { mpc_set_err("mpc_hip_create: out is NULL"); return MPC_ERR_INVALID; }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { mpc_set_err("no HIP device visible"); return MPC_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { mpc_set_err("device index out of range"); return MPC_ERR_NO_DEVICE; }
    mpc_hip_handle *h = new mpc_hip_handle();
    h->device = device;
    h->ev.assign(4 * TIMING_RING, nullptr);
    h->timed.assign(TIMING_RING, 0);
    h->total_from_k1.assign(TIMING_RING, 0);
    const int s = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        if (stream) h->stream = (hipStream_t)stream;
        else { HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
        for (auto &e : h->ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipMalloc(&h->d_controls, sizeof(double) * 2 * MPC_MAX_DEPTH));
        HIP_TRY(hipHostMalloc(&h->h_pair, sizeof(uint64_t) * 8));
        if (int s2 = static_scratch_ptr(device, &h->d_static_scratch)) return s2;
        h->models.assign(1, default_model());              /* the reference's hull until mpc_hip_set_model */
        h->models_fast_ok.assign(1, 1);
        HIP_TRY(hipMalloc(&h->d_models, sizeof(Model)));
        h->cap_models = 1;
        HIP_TRY(hipMemcpy(h->d_models, h->models.data(), sizeof(Model), hipMemcpyHostToDevice));
        return MPC_OK;
    }();
    if (s != MPC_OK) { mpc_hip_destroy(h); return s; }       /* nothing of a half-built handle leaks */
    *out = h;
    return MPC_OK;
}

extern "C" void mpc_hip_destroy(mpc_hip_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_probs); (void)hipHostFree(h->h_probs);
    (void)hipFree(h->d_results); (void)hipHostFree(h->h_results); (void)hipFree(h->d_results_alt); (void)hipFree(h->d_winners_alt);
    (void)hipFree(h->d_winners); (void)hipFree(h->d_list_count); (void)hipFree(h->d_pcnt);
    (void)hipFree(h->d_pkey); (void)hipFree(h->d_pord); (void)hipFree(h->d_pkey2);
    for (WorkMem &w : h->wm) { (void)hipFree(w.d_keys); (void)hipFree(w.d_snap); (void)hipFree(w.d_pkey); (void)hipFree(w.d_pord); (void)hipFree(w.d_pcnt); }
    (void)hipFree(h->d_controls); (void)hipFree(h->d_models); (void)hipFree(h->d_tied_rows); (void)hipHostFree(h->h_pair); (void)hipFree(h->d_plans); (void)hipFree(h->d_hist);
    (void)hipFree(h->d_hist_ring); (void)hipFree(h->d_hist_state); (void)hipFree(h->d_hist_io); (void)hipHostFree(h->h_hist_io);
    mpc_tree_free(h);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int mpc_hip_synchronize(mpc_hip_handle *h)
{
    if (!h) { mpc_set_err("handle is NULL"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (int s = flush_pending(h)) return s;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MPC_OK;
}

int mpc_ensure_problem_capacity(mpc_hip_handle *h, uint32_t n)
{
    if (n <= h->cap_probs) return MPC_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_probs); (void)hipHostFree(h->h_probs); (void)hipFree(h->d_results);
    (void)hipHostFree(h->h_results); (void)hipFree(h->d_winners);
    (void)hipFree(h->d_list_count); (void)hipFree(h->d_results_alt); (void)hipFree(h->d_winners_alt);
    h->d_probs = nullptr; h->h_probs = nullptr; h->d_results = nullptr; h->h_results = nullptr;
    h->d_winners = nullptr; h->d_list_count = nullptr; h->d_results_alt = nullptr; h->d_winners_alt = nullptr;
    h->cap_probs = 0;
    HIP_TRY(hipMalloc(&h->d_probs, sizeof(DevProblem) * n));
    HIP_TRY(hipHostMalloc(&h->h_probs, sizeof(DevProblem) * n));
    HIP_TRY(hipMalloc(&h->d_results, sizeof(DevResult) * n));
    HIP_TRY(hipHostMalloc(&h->h_results, sizeof(DevResult) * n));
    HIP_TRY(hipMalloc(&h->d_winners, sizeof(uint64_t) * (2 * (size_t)n + 8)));   /* the tree search uses 6 words */
    HIP_TRY(hipMalloc(&h->d_list_count, sizeof(uint32_t) * n));
    HIP_TRY(hipMalloc(&h->d_results_alt, sizeof(DevResult) * n));
    HIP_TRY(hipMalloc(&h->d_winners_alt, sizeof(uint64_t) * 2 * (size_t)n));
    h->cap_probs = n;
    return MPC_OK;
}

int mpc_ensure_partials(mpc_hip_handle *h, size_t n)
{
    if (n <= h->cap_partials) return MPC_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_pkey); (void)hipFree(h->d_pord); (void)hipFree(h->d_pcnt); (void)hipFree(h->d_pkey2);
    h->d_pkey = h->d_pord = h->d_pkey2 = nullptr; h->d_pcnt = nullptr; h->cap_partials = 0;
    HIP_TRY(hipMalloc(&h->d_pkey, sizeof(uint64_t) * n));
    HIP_TRY(hipMalloc(&h->d_pkey2, sizeof(uint64_t) * n));
    HIP_TRY(hipMalloc(&h->d_pord, sizeof(uint64_t) * n));
    HIP_TRY(hipMalloc(&h->d_pcnt, sizeof(uint32_t) * n));
    h->cap_partials = n;
    return MPC_OK;
}

int mpc_check_problem(const mpc_problem *p, int precision)
{
    if (p->depth < 1 || p->depth > MPC_MAX_DEPTH) { mpc_set_err("problem depth out of range [1, MPC_MAX_DEPTH]"); return MPC_ERR_TOO_LARGE; }
    if (p->n_substeps < 1 || p->n_substeps > MPC_MAX_SUBSTEPS) { mpc_set_err("problem n_substeps out of range [1, MPC_MAX_SUBSTEPS]"); return MPC_ERR_TOO_LARGE; }
    /* scores are compared through their bit patterns (key_of), which orders non-negative doubles only: the cost of
     * controller.cpp:240 is >= 0 exactly when its weights are (cfg/mpc.cfg:11-15 allows [0, 10], [0, 60], [0, 10]) */
    const double w[3] = {p->w_distance, p->w_heading, p->w_speed};
    for (double v : w)
        if (!(v >= 0.0) || !isfinite(v)) { mpc_set_err("weights must be finite and >= 0 (cfg/mpc.cfg:11-15)"); return MPC_ERR_INVALID; }
    for (int s = 0; s < p->n_substeps; s++) {
        if (!(p->substeps[s] > 0)) { mpc_set_err("substep lengths must be > 0"); return MPC_ERR_INVALID; }
        if (precision == MPC_PREC_FAST && p->substeps[s] > 0.1000001) {
            mpc_set_err("MPC_PREC_FAST needs substeps <= 0.1 s (VehicleState.h:50); use MPC_PREC_F64"); return MPC_ERR_INVALID;
        }
    }
    return MPC_OK;
}

const Model &mpc_model_of(const mpc_hip_handle *h, uint32_t problem) { return h->models[problem % h->models.size()]; }

/* What the FAST model's approximations assume of a hull (the strict model takes any): rudder angles small enough for
 * its unreduced sin/cos polynomials (<= pi/4: truncation 2e-14), a yaw of at most 0.06 rad per 0.1-s substep for its
 * small-angle rotation (the reference's hull: 0.049), and a positive prop speed per rpm (reciprocal square root). */
static bool model_fits_fast(const Model &m)
{
    const double top = fmax(sqrt(fmax((m.max_rpm * m.prop_ratio) / m.prop_pitch, 0.0)), 1.5 * m.max_speed);
    const double yaw = fabs(m.kyaw) * sin(fmin(fabs(m.max_rudder_angle), PI_2)) * top * 0.1;
    return fabs(m.max_rudder_angle) <= PI / 4 && yaw <= 0.06 && m.c_prop > 0 && m.idle_rpm >= 0 && m.rpm_span > 0;
}

int mpc_check_fast_model(const mpc_hip_handle *h, uint32_t n_problems)
{
    const uint32_t n = (uint32_t)std::min<size_t>(h->models.size(), n_problems);
    for (uint32_t i = 0; i < n; i++)
        if (!h->models_fast_ok[i]) {
            mpc_set_err("MPC_PREC_FAST does not cover this hull (rudder angle > pi/4 or yaw > 0.06 rad per substep); use MPC_PREC_F64");
            return MPC_ERR_INVALID;
        }
    return MPC_OK;
}

extern "C" void mpc_hip_default_model(mpc_model *m)
{
    if (!m) return;
    const Model d = default_model();
    m->idle_rpm = d.idle_rpm; m->max_rpm = d.max_rpm; m->max_rpm_change_rate = d.max_rpm_change_rate;
    m->prop_ratio = d.prop_ratio; m->prop_pitch = d.prop_pitch; m->max_rudder_angle = d.max_rudder_angle;
    m->rudder_coefficient = d.rudder_coefficient; m->rudder_distance = d.rudder_distance; m->mass = d.mass;
    m->max_power = d.max_power; m->max_speed = d.max_speed;
}

extern "C" int mpc_hip_set_model(mpc_hip_handle *h, const mpc_model *models, uint32_t n_models)
{
    if (!h || !models || n_models == 0) { mpc_set_err("set_model: NULL handle/models or zero models"); return MPC_ERR_INVALID; }
    std::vector<Model> packed(n_models);
    std::vector<uint8_t> ok(n_models);
    for (uint32_t i = 0; i < n_models; i++) {
        const mpc_model &m = models[i];
        const double v[11] = {m.idle_rpm, m.max_rpm, m.max_rpm_change_rate, m.prop_ratio, m.prop_pitch, m.max_rudder_angle,
                              m.rudder_coefficient, m.rudder_distance, m.mass, m.max_power, m.max_speed};
        for (double x : v) if (!isfinite(x)) { mpc_set_err("set_model: parameters must be finite"); return MPC_ERR_INVALID; }
        if (!(m.max_rpm > m.idle_rpm) || !(m.max_rpm_change_rate > 0) || !(m.prop_pitch > 0) || !(m.rudder_distance > 0) ||
            !(m.mass > 0) || !(m.max_speed > 0) || !(m.prop_ratio > 0) || !(m.max_power > 0)) {
            mpc_set_err("set_model: max_rpm > idle_rpm and positive change rate, prop ratio/pitch, rudder distance, mass, power, speed are required");
            return MPC_ERR_INVALID;
        }
        packed[i] = make_model(m.idle_rpm, m.max_rpm, m.max_rpm_change_rate, m.prop_ratio, m.prop_pitch, m.max_rudder_angle,
                               m.rudder_coefficient, m.rudder_distance, m.mass, m.max_power, m.max_speed);
        if (!isfinite(packed[i].prop_coefficient) || !isfinite(packed[i].drag_coefficient)) {
            mpc_set_err("set_model: max prop speed equals max speed (VehicleState.h:100 divides by their squares' difference)");
            return MPC_ERR_INVALID;
        }
        ok[i] = model_fits_fast(packed[i]);
    }
    if (int s = mpc_use_device(h)) return s;
    if (int s = flush_pending(h)) return s;
    if (n_models > h->cap_models) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_models); h->d_models = nullptr; h->cap_models = 0;
        HIP_TRY(hipMalloc(&h->d_models, sizeof(Model) * n_models));
        h->cap_models = n_models;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));              /* runs in flight read the old device copy */
    HIP_TRY(hipMemcpy(h->d_models, packed.data(), sizeof(Model) * n_models, hipMemcpyHostToDevice));
    h->models.swap(packed); h->models_fast_ok.swap(ok);
    h->n_probs = 0; h->ran = false;                        /* resident problems carry the old hull: upload again */
    return MPC_OK;
}

void mpc_pack_problem(const mpc_problem *p, uint64_t seed, const Model &model, DevProblem *d)
{
    d->model = model;
    d->sx = p->start_x; d->sy = p->start_y; d->sh = p->start_heading; d->sv = p->start_speed;
    d->srpm = rpm_from_speed(p->start_speed, model);
    d->last_r = p->last_rudder; d->last_t = p->last_throttle;
    d->cx = p->current_x; d->cy = p->current_y;
    d->wd = p->w_distance; d->wh = p->w_heading; d->ws = p->w_speed;
    d->gr = p->rudder_granularity; d->gt = p->throttle_granularity;
    d->minr = p->min_rudder; d->midr = p->mid_rudder; d->maxr = p->max_rudder;
    d->mint = p->min_throttle; d->maxt = p->max_throttle; d->tol = p->tolerance;
    d->seed = seed; d->depth = p->depth; d->nsub = p->n_substeps;
    d->head_uniform = p->n_substeps >= 2; d->pad0 = 0; d->pad1 = 0;
    for (int s = 1; s + 1 < p->n_substeps; s++)
        if (p->substeps[s] != p->substeps[0]) d->head_uniform = 0;
    /* 2: ... and the last substep is a negligible remainder (the 1.4e-16 s that VehicleState.h:49-53's floating-point loop
     * leaves of a 1-s level): the fast model forms its displacement and leaves the state alone (mpc_model.h) */
    if (d->head_uniform && p->substeps[p->n_substeps - 1] <= p->substeps[0] * 0x1p-40) d->head_uniform = 2;
    double level_time = 0;
    for (int s = 0; s < p->n_substeps; s++) level_time += p->substeps[s];
    for (int l = 0; l < p->depth; l++) {
        double *c = d->cp + l * CP_STRIDE;
        c[0] = p->checkpoints[l].x; c[1] = p->checkpoints[l].y;
        c[2] = p->checkpoints[l].heading; c[3] = p->checkpoints[l].speed;
        c[4] = sin(p->checkpoints[l].heading); c[5] = cos(p->checkpoints[l].heading);
        c[6] = c[0] - p->current_x * (level_time * (l + 1)); c[7] = c[1] - p->current_y * (level_time * (l + 1));
    }
    /* Reach of one level.  Every substep (VehicleState.h:111-148) moves the vessel by v' dt through the water, with
     *     v' = v + dt/m (pc (ps^2 - v^2) cos(rudder angle) - D v^3),   pc, D >= 0,  0 <= cos <= 1 (|max rudder angle| <= pi/2),
     * and the rpm never leaves [min(rpm, idle), max(rpm, max_rpm)] (it moves toward a target inside [idle, max_rpm]).  So for
     * a level entered with 0 <= v and |rpm| <= R = max(|idle|, |max_rpm|):  v' <= v + dt pc (C R)^2 / m  (the other terms
     * only slow a vessel moving forward), hence v <= V = v0 + T pc (C R)^2 / m throughout, and  v' >= v (1 - dt (pc v + D v^2) / m)
     * >= 0 as long as dt_max (pc V + D V^2) / m <= 1 (asked for with 1/2).  After the substep that ends at time t the speed
     * is at most v0 + a t (a = pc (C R)^2 / m), so the level's path, sum v' dt, is no longer than v0 T + a T (T + dt_max) / 2.
     * The margins (1e-9 on the speed, 1e-6 on the time, and 1e-6 m where the bound is used) are orders above what rounding
     * and the fast model's 1e-9 approximations of the same recurrence can add. */
    {
        const double R = fmax(fabs(model.idle_rpm), fabs(model.max_rpm));
        const double ps = model.c_prop * R;
        double dt_max = 0;
        bool ok = model.prop_coefficient >= 0 && model.drag_coefficient >= 0 && model.mass > 0 &&
                  fabs(model.max_rudder_angle) <= 1.5707963267948966 && isfinite(level_time) && p->n_substeps > 0;
        for (int s = 0; s < p->n_substeps; s++) {
            if (!(p->substeps[s] >= 0)) ok = false;
            dt_max = fmax(dt_max, p->substeps[s]);
        }
        const double pc = model.prop_coefficient, D = model.drag_coefficient;
        double vcap = -1.0;
        if (ok && dt_max > 0) {
            const double rhs = 0.5 * model.mass / dt_max;                    /* pc V + D V^2 <= rhs */
            vcap = D > 0 ? (sqrt(pc * pc + 4.0 * D * rhs) - pc) / (2.0 * D) : (pc > 0 ? rhs / pc : 1.0e300);
            vcap *= 1.0 - 1.0e-9;
        }
        d->reach_dv = level_time * pc * ps * ps / model.mass * (1.0 + 1.0e-9);
        d->reach_t = level_time * (1.0 + 1.0e-6);
        d->reach_acc = 0.5 * d->reach_dv * (level_time + dt_max) * (1.0 + 1.0e-6);
        d->reach_acc2 = d->reach_dv * (2.0 * level_time + dt_max) * (1.0 + 1.0e-6);      /* a 2T (2T + dt) / 2 */
        d->reach_pad = 0;
        /* the heading turns at rudder_coefficient * max(sqrt(prop speed), v) * sin(rudder angle) / rudder_distance
         * (VehicleState.h:122-128), |rudder angle| <= |max_rudder_angle| <= pi/2: while v stays below sqrt(C R), by at most
         * D = that rate at sqrt(C R) times the level's length */
        d->reach_sqrtps = sqrt(fabs(ps));
        const double turn = fabs(model.rudder_coefficient / model.rudder_distance) * sin(fabs(model.max_rudder_angle)) * d->reach_sqrtps *
                            level_time * (1.0 + 1.0e-6) + 1.0e-6;
        d->reach_sd = turn < 1.5 ? sin(turn) : 1.0;
        d->reach_cd = turn < 1.5 ? cos(turn) : -1.0;
        d->reach_vcap = ok && isfinite(d->reach_dv) && isfinite(vcap) ? vcap : -1.0;
        d->reach_rpm = R;
        d->reach_cx = p->current_x * level_time; d->reach_cy = p->current_y * level_time;
    }
    for (int s = 0; s < p->n_substeps; s++) {
        double *q = d->sub + s * SUB_STRIDE;
        const double dt = p->substeps[s];
        q[0] = dt; q[1] = model.max_rpm_change_rate * dt; q[2] = dt / model.mass;
        q[3] = p->current_x * dt; q[4] = p->current_y * dt; q[5] = 1.0 / dt;
    }
}

extern "C" int mpc_hip_upload_problems(mpc_hip_handle *h, const mpc_problem *h_problems, const uint64_t *h_seeds,
                                       uint32_t n_problems)
{
    if (!h || !h_problems || n_problems == 0) { mpc_set_err("upload_problems: NULL handle/problems or zero problems"); return MPC_ERR_INVALID; }
    if (n_problems > 65535) { mpc_set_err("more than 65535 problems in one batch"); return MPC_ERR_TOO_LARGE; }
    if (int s = mpc_use_device(h)) return s;
    if (int s = flush_pending(h)) return s;          /* a pending finish reads the resident problems */
    for (uint32_t p = 0; p < n_problems; p++)
        if (int s = mpc_check_problem(&h_problems[p], MPC_PREC_F64)) return s;
    if (int s = mpc_ensure_problem_capacity(h, n_problems)) return s;
    HIP_TRY(hipStreamSynchronize(h->stream));     /* the pinned staging buffer may still be in flight */
    h->depth.resize(n_problems); h->nsub.resize(n_problems);
    for (uint32_t p = 0; p < n_problems; p++) {
        mpc_pack_problem(&h_problems[p], h_seeds ? h_seeds[p] : 0, mpc_model_of(h, p), &h->h_probs[p]);
        h->depth[p] = h_problems[p].depth; h->nsub[p] = h_problems[p].n_substeps;
    }
    HIP_TRY(hipMemcpyAsync(h->d_probs, h->h_probs, sizeof(DevProblem) * n_problems, hipMemcpyHostToDevice, h->stream));
    h->n_probs = n_problems;
    h->ran = false;
    return MPC_OK;
}

/* dynamic LDS of K1 without staged controls (checkpoints + schedule + argmin words), and what one launch may ask for */
#define K1_LDS_BASE (sizeof(double) * (MPC_MAX_DEPTH * CP_STRIDE + MPC_MAX_SUBSTEPS * SUB_STRIDE) + sizeof(uint64_t) * 12)
#define K1_LDS_LIMIT ((size_t)64 * 1024)

/* ---- the deferred finish (mpc_hip_set_pipeline) ----------------------------------------------------------------
 * A run is K1 (rollouts) + K2 (finish).  With the pipeline on, a run enqueues only K1 and leaves its K2 PENDING; the
 * next run's K1 launch carries it as one extra workgroup per problem (rollout_kernel<.., FUSE>), and any call that
 * needs the results (fetch, winners, upload, synchronize, ...) launches it on its own first.  Working memory that K1
 * writes and K2 reads is double-buffered (h->wm[0..1]; level 2 rotates over all four sets) so that K1 of solve n+1 never
 * overwrites what K2 of solve n reads. */
static int launch_group(mpc_hip_handle *h);

/* K2's workgroup: 256 lanes reduce up to 1024 workgroup triples per trip; a problem with few of them (a fleet: 32 per
 * vessel) gets ONE wave -- its finish is one lane rolling the winner out again, and 4096 such workgroups of four waves
 * took 0.25 ms where 4096 single waves share the SIMDs four at a time */
static inline uint32_t finish_block(uint32_t nblk) { return nblk <= 256 ? 64u : 256u; }

static int finish_standalone(mpc_hip_handle *h, const StashedRun &r)
{
    const dim3 grid(r.n_probs), blk(finish_block(r.fin.nblk));
    if (r.seeded) hipLaunchKernelGGL((finish_kernel<true>), grid, blk, 0, h->stream, r.fin);
    else hipLaunchKernelGGL((finish_kernel<false>), grid, blk, 0, h->stream, r.fin);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

/* level >= 2's part: the runs waiting for their group to fill and the finishes of the last launch's runs */
static int flush_pairs(mpc_hip_handle *h)
{
    if (h->n_stash) { if (int s = launch_group(h)) return s; }              /* an incomplete group goes as it is */
    if (h->n_pendg > 1) {                                                   /* the runs of one launch: one launch for all their finishes */
        FinishGroup G{};
        uint32_t nblk = 0;
        for (int i = 0; i < h->n_pendg; i++) { G.f[i] = h->pendg[i].fin; nblk = std::max(nblk, h->pendg[i].fin.nblk); }
        const dim3 grid(h->pendg[0].n_probs, (uint32_t)h->n_pendg), blk(finish_block(nblk));   /* (only the newest writes the handle's result buffers) */
        if (h->pendg[0].seeded) hipLaunchKernelGGL((finish_group_kernel<true>), grid, blk, 0, h->stream, G);
        else hipLaunchKernelGGL((finish_group_kernel<false>), grid, blk, 0, h->stream, G);
        HIP_TRY(hipGetLastError());
    } else if (h->n_pendg == 1) {
        if (int s = finish_standalone(h, h->pendg[0])) return s;
    }
    h->n_pendg = 0;
    return MPC_OK;
}

static int flush_pending(mpc_hip_handle *h)
{
    if (int s = flush_pairs(h)) return s;
    if (!h->pending) return MPC_OK;
    hipEvent_t *ev = &h->ev[4 * h->pend_slot];
    hipEvent_t k2a = h->pend_timed ? ev[2] : nullptr, k2b = h->pend_timed ? ev[3] : nullptr;
    const dim3 grid(h->pend_n_probs), blk(finish_block(h->pend.nblk));
    if (h->pend_seeded) hipExtLaunchKernelGGL((finish_kernel<true>), grid, blk, 0, h->stream, k2a, k2b, 0, h->pend);
    else hipExtLaunchKernelGGL((finish_kernel<false>), grid, blk, 0, h->stream, k2a, k2b, 0, h->pend);
    h->pending = false;
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

int mpc_flush_pending(mpc_hip_handle *h) { return flush_pending(h); }

static int ensure_buffer(mpc_hip_handle *h, void **buf, size_t *cap, size_t need_bytes)
{
    if (need_bytes <= *cap) return MPC_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(*buf); *buf = nullptr; *cap = 0;
    HIP_TRY(hipMalloc(buf, need_bytes));
    *cap = need_bytes;
    return MPC_OK;
}

/* ---- several runs per launch (mpc_hip_set_pipeline level K = 2 .. MPC_GROUP_MAX) ----------------------------------
 * A 65,536-candidate set is one wave per SIMD; the second wave a SIMD needs to issue a VALU instruction every 4 cycles
 * instead of every ~6.2 can only come from another solve, and the ~10 us a launch costs whatever it holds (dispatch
 * boundary, prologue, tail) are paid once per launch.  At level K a run that can share a launch (fast, uniform substeps,
 * one-wave workgroups) is only PREPARED by its call; the K-th such run launches the whole group in one dispatch that also
 * carries the finish workgroups of the launch before (rollout_group_kernel); an incomplete group goes as it is when
 * anything asks for results (flush_pending).  Working memory rotates over 2 K sets: the runs being launched and the
 * runs whose finishes ride along. */
static int launch_group(mpc_hip_handle *h)
{
    const int n = h->n_stash;
    StashedRun run[MPC_GROUP_MAX];
    for (int i = 0; i < n; i++) run[i] = h->stash[i];
    h->n_stash = 0;
    /* finishes that cannot ride along (another candidate source or batch size) go first, on their own */
    if (h->n_pendg && (h->pendg[0].seeded != run[0].seeded || h->pendg[0].n_probs != run[0].n_probs)) {
        for (int i = 0; i < h->n_pendg; i++)
            if (int s = finish_standalone(h, h->pendg[i])) return s;
        h->n_pendg = 0;
    }
    GroupArgs A{};
    uint32_t end = 0;
    size_t lds = 0;
    const StashedRun *t = nullptr;                       /* one pair of events per dispatch: the first timed run's */
    for (int i = 0; i < MPC_GROUP_MAX; i++) {
        if (i < n) {
            A.run[i] = run[i].run;
            end += run[i].run.nblk;
            lds = std::max(lds, run[i].lds);
            if (!t && run[i].timed) t = &run[i];
        }
        A.end[i] = end;                                  /* unused runs: no workgroups */
    }
    for (int i = 0; i < h->n_pendg; i++) A.prev[i] = h->pendg[i].fin;
    const uint32_t n_prev = (uint32_t)h->n_pendg;
    const dim3 grid(end + n_prev, run[0].n_probs), blk(64);
    hipEvent_t k1a = t ? h->ev[4 * t->slot] : nullptr, k1b = t ? h->ev[4 * t->slot + 1] : nullptr;
    for (int i = 0; i < n; i++) h->timed[run[i].slot] = t == &run[i];
    if (t) h->total_from_k1[t->slot] = 1;                /* the finishes have no dispatch of their own */
    if (run[0].seeded) hipExtLaunchKernelGGL((rollout_group_kernel<true>), grid, blk, lds, h->stream, k1a, k1b, 0, h->d_probs, A);
    else hipExtLaunchKernelGGL((rollout_group_kernel<false>), grid, blk, lds, h->stream, k1a, k1b, 0, h->d_probs, A);
    HIP_TRY(hipGetLastError());
    /* several finishes in one launch: only the newest run's writes the handle's results */
    for (int i = 0; i < n; i++) {
        h->pendg[i] = run[i];
        if (i + 1 < n) { h->pendg[i].fin.results = h->d_results_alt; h->pendg[i].fin.winners = h->d_winners_alt; }
    }
    h->n_pendg = n;
    return MPC_OK;
}

/* use_static: working memory = the device's scratch[] instead of the handle's own buffers, no snapshots; the call
 * then allocates (and frees) nothing and rejects sets beyond the array's fixed capacity */
static int run_common(mpc_hip_handle *h, bool seeded, const float *d_rud, const float *d_thr, uint64_t first,
                      uint64_t count, int precision, int selection, double *d_cost_last, double *d_cost_cum,
                      bool use_static = false)
{
    if (!h) { mpc_set_err("handle is NULL"); return MPC_ERR_INVALID; }
    if (h->n_probs == 0) { mpc_set_err("no problems uploaded"); return MPC_ERR_INVALID; }
    if (count == 0) { mpc_set_err("candidate count is zero"); return MPC_ERR_INVALID; }
    if (precision != MPC_PREC_FAST && precision != MPC_PREC_F64) { mpc_set_err("unknown precision"); return MPC_ERR_INVALID; }
    if (selection != MPC_SELECT_LAST_STEP && selection != MPC_SELECT_CUMULATIVE) { mpc_set_err("unknown selection"); return MPC_ERR_INVALID; }
    if (!seeded && (!d_rud || !d_thr)) { mpc_set_err("control arrays are NULL"); return MPC_ERR_INVALID; }
    if (!seeded && h->n_probs != 1) { mpc_set_err("explicit control arrays take exactly one problem"); return MPC_ERR_INVALID; }
    if (count > (1ull << 40)) { mpc_set_err("candidate count too large"); return MPC_ERR_TOO_LARGE; }
    if (int s = mpc_use_device(h)) return s;
    if (precision == MPC_PREC_FAST)
        if (int s = mpc_check_fast_model(h, h->n_probs)) return s;
    if (precision == MPC_PREC_FAST)
        for (uint32_t p = 0; p < h->n_probs; p++)
            for (int s = 0; s < h->nsub[p]; s++)
                if (h->h_probs[p].sub[s * SUB_STRIDE] > 0.1000001) {
                    mpc_set_err("MPC_PREC_FAST needs substeps <= 0.1 s (VehicleState.h:50); use MPC_PREC_F64");
                    return MPC_ERR_INVALID;
                }
    /* one wave per workgroup while the set is small (spreads the waves over all SIMDs), 256 lanes otherwise */
    uint32_t block = (count * h->n_probs >= (1u << 20)) ? 256u : 64u;
    /* explicit sets stage 2 * depth * block FP32 controls in LDS next to the 7.3 KB of problem data: the workgroup
     * shrinks until that fits the 64 KiB a launch gets without opting in to more (256 lanes up to depth 28, 128 up to
     * depth 56, 64 beyond: 39.3 KiB at MPC_MAX_DEPTH).  A candidate's cost does not depend on its workgroup's size. */
    while (!seeded && block > 64 && K1_LDS_BASE + sizeof(float) * 2 * (size_t)h->depth[0] * block > K1_LDS_LIMIT) block >>= 1;
    const uint32_t cpb = block;                   /* candidates per workgroup */
    const uint64_t nblk64 = (count + cpb - 1) / cpb;
    if (nblk64 > 0x7FFFFFFEull) { mpc_set_err("candidate count too large for one launch"); return MPC_ERR_TOO_LARGE; }
    const uint32_t nblk = (uint32_t)nblk64;

    if (h->d_winner_sink && h->winner_sink_words > 2)
        for (uint32_t p = 0; p < h->n_probs; p++)
            if (2u + 2u * (uint32_t)h->depth[p] > h->winner_sink_words) {
                mpc_set_err("the winner record sink is too shallow for a resident problem (mpc_hip_set_winner_record_sink depth)");
                return MPC_ERR_INVALID;
            }
    const int fast = precision == MPC_PREC_FAST;
    /* all substeps of a level equally long (configs 2-5): the kernels keep the schedule row in registers */
    bool uniform = true;
    for (uint32_t p = 0; p < h->n_probs && uniform; p++)
        for (int s = 1; s < h->nsub[p]; s++)
            if (h->h_probs[p].sub[s * SUB_STRIDE] != h->h_probs[p].sub[0]) { uniform = false; break; }
    /* level 2 pairs up what the pair kernel has a variant for; any other run is treated as at level 1 */
    const bool pipelined = h->pipeline && !use_static;
    const bool pairable = pipelined && h->pipeline >= 2 && fast && uniform && block == 64;
    if (!pipelined || (pairable && h->pending)) { if (int s = flush_pending(h)) return s; }
    else if (!pairable) { if (int s = flush_pairs(h)) return s; }
    /* which set of the working memory this run writes: the others may still be read by pending finishes */
    const int half = pairable ? (int)(h->wm_rr++ % (2u * MPC_GROUP_MAX)) : pipelined ? (h->pending ? 1 - h->pend_half : 0) : 0;
    WorkMem &wm = h->wm[half];
    double *d_keys; uint64_t *d_pkey, *d_pord; uint32_t *d_pcnt;
    if (use_static) {
        if ((uint64_t)h->n_probs * count > MPC_STATIC_MAX_CANDIDATES) {
            mpc_set_err("run_seeded_static: problems x candidates exceeds MPC_STATIC_MAX_CANDIDATES (2^20); use mpc_hip_run_seeded");
            return MPC_ERR_TOO_LARGE;
        }
        if ((uint64_t)h->n_probs * nblk > MPC_STATIC_MAX_WORKGROUPS) {
            mpc_set_err("run_seeded_static: problems x workgroups exceeds MPC_STATIC_MAX_WORKGROUPS (2^15); use mpc_hip_run_seeded");
            return MPC_ERR_TOO_LARGE;
        }
        if (!h->d_static_scratch) { mpc_set_err("run_seeded_static: the device's scratch address is not resolved"); return MPC_ERR_HIP; }
        StaticScratch *S = (StaticScratch *)h->d_static_scratch;       /* the device address, never the symbol's name */
        d_keys = S->keys; d_pkey = S->pkey; d_pord = S->pord; d_pcnt = S->pcnt;
    } else {
        const size_t n_part = (size_t)h->n_probs * nblk;
        if (half == 0) {                       /* half 0's triples are the tree search's too (h->d_pkey ...) */
            if (int s = mpc_ensure_partials(h, n_part)) return s;
            d_pkey = h->d_pkey; d_pord = h->d_pord; d_pcnt = h->d_pcnt;
        } else {
            if (int s = ensure_buffer(h, (void **)&wm.d_pkey, &wm.cap_pkey, sizeof(uint64_t) * n_part)) return s;
            if (int s = ensure_buffer(h, (void **)&wm.d_pord, &wm.cap_pord, sizeof(uint64_t) * n_part)) return s;
            if (int s = ensure_buffer(h, (void **)&wm.d_pcnt, &wm.cap_pcnt, sizeof(uint32_t) * n_part)) return s;
            d_pkey = wm.d_pkey; d_pord = wm.d_pord; d_pcnt = wm.d_pcnt;
        }
        if (int s = ensure_buffer(h, (void **)&wm.d_keys, &wm.cap_keys, sizeof(double) * (size_t)h->n_probs * count)) return s;
        d_keys = wm.d_keys;
    }

    /* per-level snapshots of every candidate (so the winner's trajectory needs no second rollout) while they
     * stay small; beyond that the winner is rolled out again by one lane, which is then noise */
    int snap_depth = 0;
    for (uint32_t p = 0; p < h->n_probs; p++) snap_depth = std::max(snap_depth, (int)h->depth[p]);
    /* every level for a run whose K2 is a launch of its own (it then only reads), every 4th where the finish rides under
     * the next rollout launch (a stream of small fast solves) and where every level would not fit */
    const bool hidden_finish = pipelined && fast && uniform && block == 64;    /* the runs whose K2 can ride in a later K1 launch */
    int snap_shift = hidden_finish ? SNAP_SHIFT_STREAM : 0;
    size_t snap_need = (size_t)h->n_probs * count * snap_slots(snap_depth, snap_shift) * SNAP_FIELDS;
    if (snap_shift == 0 && snap_need * sizeof(double) > MPC_SNAPSHOT_BYTES_MAX) {
        snap_shift = SNAP_SHIFT_STREAM;
        snap_need = (size_t)h->n_probs * count * snap_slots(snap_depth, snap_shift) * SNAP_FIELDS;
    }
    double *d_snap = nullptr;
    if (!use_static && snap_need > 0 && snap_need * sizeof(double) <= MPC_SNAPSHOT_BYTES_MAX) {
        if (int s = ensure_buffer(h, (void **)&wm.d_snap, &wm.cap_snap, sizeof(double) * snap_need)) return s;
        d_snap = wm.d_snap;
    }
    const double rel = fast ? MPC_REFINE_REL : 0.0;
    const uint32_t slot = (uint32_t)(h->runs % TIMING_RING);
    hipEvent_t *ev = &h->ev[4 * slot];
    /* the pending finish rides along when this launch has a variant for it: fast uniform rollouts in one-wave
     * workgroups (every throughput stream of small sets: config 2), the same kind of candidate source, the same batch.
     * (Dense launches gain nothing from it: a finish workgroup among 256-lane rollout workgroups holds four wave slots
     * for as long as a rollout workgroup does -- config 5 measured 15.11 ms fused against 15.07 ms with its own launch.) */
    const bool fuse = pipelined && !pairable && h->pending && fast && uniform && block == 64 && h->pend_seeded == seeded &&
                      h->pend_n_probs == h->n_probs;
    if (h->pending && !fuse) { if (int s = flush_pending(h)) return s; }
    const dim3 grid(nblk + (fuse ? 1u : 0u), h->n_probs), blk(block);
    /* explicit sets: all levels' controls of a workgroup are staged in LDS as FP32 (<= 64 KiB by the block choice);
     * a fused finish keeps its near-tie list (+ 16 words) in the same place */
    /* small seeded sets (fast, one-wave workgroups, depth <= 32): every level's controls are generated into LDS by the
     * prologue (PREGEN, mpc_kernels.h): 2 * depth * 64 doubles */
    const bool pregen = seeded && fast && block == 64 && snap_depth <= 32 && !pairable;   /* (two waves per SIMD: 2 x 4 x 28 KB would not fit a CU's LDS) */
    size_t ctl_bytes = seeded ? (pregen ? sizeof(double) * 2 * (size_t)snap_depth * block : 0) : sizeof(float) * 2 * (size_t)snap_depth * block;
    if (fuse || pairable) ctl_bytes = std::max(ctl_bytes, sizeof(uint64_t) * (LIST_CAP + FINISH_LDS_WORDS));
    const size_t lds = K1_LDS_BASE + ctl_bytes;
    if (lds > K1_LDS_LIMIT) { mpc_set_err("rollout kernel: staged controls do not fit the workgroup's LDS"); return MPC_ERR_TOO_LARGE; }
    /* timed runs carry their events ON the dispatches (the kernels' own begin/end, as a profiler reads them):
     * recording events between the launches instead costs ~4 barrier packets = 15 us per 59-us solve */
    const bool timed = h->timing_period && (h->runs % h->timing_period) == 0;
    h->timed[slot] = timed;
    h->total_from_k1[slot] = 0;
    if (fuse) h->total_from_k1[h->pend_slot] = 1;          /* that run's finish has no dispatch of its own: hidden in this K1 */
    const FinishArgs fin{h->d_probs, d_rud, d_thr, first, count, selection, fast, (int)uniform, rel, nblk, cpb,
                         d_pkey, d_pord, d_pcnt, d_keys, d_snap, snap_depth, h->d_winners, h->d_winner_sink, h->winner_sink_words, snap_shift,
                         h->d_results};
    if (pairable) {
        const StashedRun rec{RunArgs{d_rud, d_thr, first, count, selection, snap_depth, rel, d_cost_last, d_cost_cum, d_keys, d_snap,
                                     d_pkey, d_pord, d_pcnt, nblk, snap_shift},
                             fin, seeded, timed, lds, h->n_probs, slot};
        h->timed[slot] = false;                                         /* set by the launch that carries the events */
        int s = MPC_OK;
        /* a run of another kind (candidate source, batch size) does not share a launch with the waiting ones */
        if (h->n_stash && (h->stash[0].seeded != seeded || h->stash[0].n_probs != h->n_probs)) s = launch_group(h);
        if (s) return s;
        h->stash[h->n_stash++] = rec;
        if (h->n_stash >= std::min(h->pipeline, MPC_GROUP_MAX)) s = launch_group(h);
        if (s) return s;
        h->runs++;
        h->last_count = count; h->last_first = first; h->last_precision = precision; h->last_selection = selection;
        h->last_seeded = seeded; h->ran = true;
        h->last_rud = d_rud; h->last_thr = d_thr;
        return MPC_OK;
    }
    hipEvent_t k1a = timed ? ev[0] : nullptr, k1b = timed ? ev[1] : nullptr;
    const FinishArgs prev = fuse ? h->pend : FinishArgs{};
#define LAUNCH_K1(PREC, SEEDED)                                                                               \
    if (uniform) LAUNCH_K1U(PREC, SEEDED, true, false); else LAUNCH_K1U(PREC, SEEDED, false, false)
#define LAUNCH_K1U(PREC, SEEDED, UNI, FUSE)                                                                   \
    hipExtLaunchKernelGGL((rollout_kernel<PREC, SEEDED, UNI, FUSE>), grid, blk, lds, h->stream, k1a, k1b, 0, h->d_probs, d_rud, d_thr, \
                       first, count, selection, rel, d_cost_last, d_cost_cum, d_keys, d_snap, snap_depth, snap_shift, \
                       d_pkey, d_pord, d_pcnt, prev)
#define LAUNCH_K1P(UNI, FUSE)                                                                                 \
    hipExtLaunchKernelGGL((rollout_kernel<MPC_PREC_FAST, true, UNI, FUSE, true>), grid, blk, lds, h->stream, k1a, k1b, 0, h->d_probs, d_rud, d_thr, \
                       first, count, selection, rel, d_cost_last, d_cost_cum, d_keys, d_snap, snap_depth, snap_shift, \
                       d_pkey, d_pord, d_pcnt, prev)
    if (fuse && pregen) LAUNCH_K1P(true, true);
    else if (fuse) { if (seeded) LAUNCH_K1U(MPC_PREC_FAST, true, true, true); else LAUNCH_K1U(MPC_PREC_FAST, false, true, true); }
    else if (pregen) { if (uniform) LAUNCH_K1P(true, false); else LAUNCH_K1P(false, false); }
    else if (precision == MPC_PREC_F64) { if (seeded) LAUNCH_K1(MPC_PREC_F64, true); else LAUNCH_K1(MPC_PREC_F64, false); }
    else { if (seeded) LAUNCH_K1(MPC_PREC_FAST, true); else LAUNCH_K1(MPC_PREC_FAST, false); }
#undef LAUNCH_K1
#undef LAUNCH_K1U
#undef LAUNCH_K1P
    HIP_TRY(hipGetLastError());
    /* this run's finish: pending (pipeline) or launched right away */
    h->pend = fin;
    h->pending = true; h->pend_seeded = seeded; h->pend_n_probs = h->n_probs; h->pend_half = half;
    h->pend_slot = slot; h->pend_timed = timed;
    if (!pipelined) { if (int s = flush_pending(h)) return s; }
    h->runs++;
    h->last_count = count; h->last_first = first; h->last_precision = precision; h->last_selection = selection;
    h->last_seeded = seeded; h->ran = true;
    h->last_rud = d_rud; h->last_thr = d_thr;
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

/* mpc_hip_run_seeded without any allocation: K1's keys and workgroup triples live in the device's scratch[] (nothing in
 * it has to start at zero: K1 writes every word K2 reads), there are no per-level snapshots (K2 rolls the winner out
 * again on one lane -- the same arithmetic, the same bits), results go where mpc_hip_run_seeded puts them. */
extern "C" int mpc_hip_run_seeded_static(mpc_hip_handle *h, uint64_t first_ordinal, uint64_t count, int precision,
                                         int selection, double *d_cost_last, double *d_cost_cum)
{
    return run_common(h, true, nullptr, nullptr, first_ordinal, count, precision, selection, d_cost_last, d_cost_cum, true);
}

extern "C" int mpc_hip_set_pipeline(mpc_hip_handle *h, int on)
{
    if (!h) { mpc_set_err("handle is NULL"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (on < 0 || on > MPC_GROUP_MAX) { mpc_set_err("set_pipeline: level out of range [0, 4]"); return MPC_ERR_INVALID; }
    if (on != h->pipeline) { if (int s = flush_pending(h)) return s; }
    h->pipeline = on;
    return MPC_OK;
}

extern "C" int mpc_hip_flush(mpc_hip_handle *h)
{
    if (!h) { mpc_set_err("handle is NULL"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    return flush_pending(h);
}

extern "C" int mpc_hip_set_timing(mpc_hip_handle *h, uint32_t period)
{
    if (!h) { mpc_set_err("handle is NULL"); return MPC_ERR_INVALID; }
    h->timing_period = period;
    return MPC_OK;
}

extern "C" int mpc_hip_timing(mpc_hip_handle *h, int reset, uint64_t *runs, double *kernel_ms_sum, double *total_ms_sum)
{
    if (!h) { mpc_set_err("handle is NULL"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (int s = flush_pending(h)) return s;
    HIP_TRY(hipStreamSynchronize(h->stream));
    uint64_t from = h->timing_base;
    if (h->runs - from > TIMING_RING) from = h->runs - TIMING_RING;
    double k = 0, t = 0;
    uint64_t n_timed = 0;
    for (uint64_t r = from; r < h->runs; r++) {
        if (!h->timed[r % TIMING_RING]) continue;
        n_timed++;
        hipEvent_t *ev = &h->ev[4 * (r % TIMING_RING)];
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&b, ev[0], h->total_from_k1[r % TIMING_RING] ? ev[1] : ev[3]));
        k += a; t += b;
    }
    if (runs) *runs = n_timed;
    if (kernel_ms_sum) *kernel_ms_sum = k;
    if (total_ms_sum) *total_ms_sum = t;
    if (reset) h->timing_base = h->runs;
    return MPC_OK;
}

static void fill_result(const mpc_hip_handle *h, uint32_t p, const DevResult &R, float k1_ms, float total_ms,
                        mpc_result *out)
{
    memset(out, 0, sizeof *out);
    if (R.ordinal == ~0ull) {            /* no candidate with a finite score: nothing selected */
        out->best_ordinal = R.ordinal;
        out->best_last = out->best_cumulative = INFINITY;
        out->evaluated = h->last_count;
        out->kernel_ms = k1_ms; out->total_ms = total_ms;
        out->status = MPC_ERR_NO_ITERATION;
        return;
    }
    const int D = h->depth[p];
    out->levels = D;
    out->best_ordinal = R.ordinal;
    out->best_last = R.best_last; out->best_cumulative = R.best_cum;
    out->first_rudder = R.traj[1].rudder; out->first_throttle = R.traj[1].throttle;
    out->evaluated = h->last_count;
    out->rollout_steps = h->last_count * (uint64_t)D * (uint64_t)h->nsub[p];
    out->refined = R.refined;
    out->libm_tie = R.tie;
    out->kernel_ms = k1_ms; out->total_ms = total_ms;
    memcpy(out->trajectory, R.traj, sizeof(mpc_level_state) * (size_t)(D + 1));
    out->status = MPC_OK;
    if (!isfinite(out->first_rudder) || !isfinite(out->first_throttle)) out->status = MPC_ERR_NON_FINITE;   /* controller.cpp:201-202 */
}

static int fetch_internal(mpc_hip_handle *h, mpc_result *h_results, uint32_t n_problems)
{
    if (!h || !h_results) { mpc_set_err("fetch_results: NULL argument"); return MPC_ERR_INVALID; }
    if (!h->ran) { mpc_set_err("fetch_results: nothing has run since the last upload"); return MPC_ERR_INVALID; }
    if (n_problems != h->n_probs) { mpc_set_err("fetch_results: n_problems differs from the uploaded batch"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (int s = flush_pending(h)) return s;
    HIP_TRY(hipMemcpyAsync(h->h_results, h->d_results, sizeof(DevResult) * n_problems, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float k1 = 0, total = 0;
    hipEvent_t *ev = &h->ev[4 * ((h->runs - 1) % TIMING_RING)];
    if (h->timed[(h->runs - 1) % TIMING_RING]) {
        (void)hipEventElapsedTime(&k1, ev[0], ev[1]);
        (void)hipEventElapsedTime(&total, ev[0], h->total_from_k1[(h->runs - 1) % TIMING_RING] ? ev[1] : ev[3]);
    }
    bool overflow = false;
    for (uint32_t p = 0; p < n_problems; p++) overflow |= h->h_results[p].overflow != 0;
    for (uint32_t p = 0; p < n_problems; p++) fill_result(h, p, h->h_results[p], k1, total, &h_results[p]);
    /* more than LIST_CAP candidates within MPC_REFINE_REL of the minimum: the caller decides the whole set with
     * the strict kernel (whose argmin resolves ties by ordinal exactly) */
    return (overflow && h->last_precision == MPC_PREC_FAST) ? MPC_INTERNAL_RERUN_STRICT : MPC_OK;
}

extern "C" int mpc_hip_fetch_results(mpc_hip_handle *h, mpc_result *h_results, uint32_t n_problems)
{
    const int s = fetch_internal(h, h_results, n_problems);
    if (s != MPC_INTERNAL_RERUN_STRICT) return s;
    mpc_set_err("more than 1024 candidates tie with the minimum; rerun with MPC_PREC_F64");
    return MPC_ERR_TOO_LARGE;
}

extern "C" int mpc_hip_device_winners(mpc_hip_handle *h, uint64_t **d_winners)
{
    if (!h || !d_winners) { mpc_set_err("device_winners: NULL argument"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (int s = flush_pending(h)) return s;
    *d_winners = h->d_winners;
    return MPC_OK;
}

extern "C" int mpc_hip_set_winner_sink(mpc_hip_handle *h, uint64_t *d_dst)
{
    if (!h) { mpc_set_err("handle is NULL"); return MPC_ERR_INVALID; }
    h->d_winner_sink = d_dst;
    h->winner_sink_words = 2;
    return MPC_OK;
}

extern "C" int mpc_hip_set_winner_record_sink(mpc_hip_handle *h, uint64_t *d_dst, uint32_t depth)
{
    if (!h) { mpc_set_err("handle is NULL"); return MPC_ERR_INVALID; }
    if (depth < 1 || depth > MPC_MAX_DEPTH) { mpc_set_err("winner record depth out of range [1, MPC_MAX_DEPTH]"); return MPC_ERR_INVALID; }
    h->d_winner_sink = d_dst;
    h->winner_sink_words = MPC_WINNER_RECORD_WORDS(depth);
    return MPC_OK;
}

extern "C" int mpc_hip_copy_winner_records(mpc_hip_handle *h, uint64_t *d_dst, uint32_t depth)
{
    if (!h || !d_dst || h->n_probs == 0) { mpc_set_err("copy_winner_records: NULL argument or nothing uploaded"); return MPC_ERR_INVALID; }
    if (depth < 1 || depth > MPC_MAX_DEPTH) { mpc_set_err("winner record depth out of range [1, MPC_MAX_DEPTH]"); return MPC_ERR_INVALID; }
    if (!h->ran) { mpc_set_err("copy_winner_records: nothing has run since the last upload"); return MPC_ERR_INVALID; }
    for (uint32_t p = 0; p < h->n_probs; p++)
        if ((uint32_t)h->depth[p] > depth) { mpc_set_err("copy_winner_records: a resident problem is deeper than the record"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (int s = flush_pending(h)) return s;
    hipLaunchKernelGGL(pack_records_kernel, dim3(h->n_probs), dim3(64), 0, h->stream, h->d_results, h->d_winners, h->d_probs,
                       (uint32_t)MPC_WINNER_RECORD_WORDS(depth), d_dst);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_hip_copy_winners(mpc_hip_handle *h, uint64_t *d_dst)
{
    if (!h || !d_dst || h->n_probs == 0) { mpc_set_err("copy_winners: NULL argument or nothing uploaded"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (int s = flush_pending(h)) return s;
    HIP_TRY(hipMemcpyAsync(d_dst, h->d_winners, sizeof(uint64_t) * 2 * h->n_probs, hipMemcpyDeviceToDevice, h->stream));
    return MPC_OK;
}

extern "C" int mpc_hip_combine_shards(mpc_hip_handle *h, const uint64_t *d_gathered, uint32_t n_shards,
                                      uint32_t n_problems, uint64_t ordinal_stride, uint64_t *d_out)
{
    if (!h || !d_gathered || !d_out || n_shards == 0 || n_problems == 0) { mpc_set_err("combine_shards: NULL or empty argument"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    hipLaunchKernelGGL(combine_shards_kernel, dim3((n_problems + 255) / 256), dim3(256), 0, h->stream, d_gathered, n_shards,
                       n_problems, ordinal_stride, d_out);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_hip_combine_shards_exact(mpc_hip_handle *h, const uint64_t *d_gathered, uint32_t n_shards, uint32_t n_rows,
                                            uint32_t depth, uint64_t ordinal_stride, int precision, int selection,
                                            uint64_t *d_out, uint32_t *h_tied_rows)
{
    if (h && !d_gathered && h_tied_rows) {           /* read the counter only */
        if (int s = mpc_use_device(h)) return s;
        *h_tied_rows = 0;
        if (!h->d_tied_rows) return MPC_OK;
        HIP_TRY(hipMemcpyAsync(h->h_pair, h->d_tied_rows, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(h_tied_rows, h->h_pair, sizeof(uint32_t));
        return MPC_OK;
    }
    if (!h || !d_gathered || !d_out || n_shards == 0 || n_rows == 0) { mpc_set_err("combine_shards_exact: NULL or empty argument"); return MPC_ERR_INVALID; }
    if (n_shards > 64) { mpc_set_err("combine_shards_exact: more than 64 shards"); return MPC_ERR_TOO_LARGE; }
    if (precision != MPC_PREC_FAST && precision != MPC_PREC_F64) { mpc_set_err("combine_shards_exact: unknown precision"); return MPC_ERR_INVALID; }
    if (selection != MPC_SELECT_LAST_STEP && selection != MPC_SELECT_CUMULATIVE) { mpc_set_err("combine_shards_exact: unknown selection"); return MPC_ERR_INVALID; }
    if (depth < 1 || depth > MPC_MAX_DEPTH) { mpc_set_err("winner record depth out of range [1, MPC_MAX_DEPTH]"); return MPC_ERR_INVALID; }
    if (h->n_probs == 0) { mpc_set_err("combine_shards_exact: no resident problems (the re-scoring reads them)"); return MPC_ERR_INVALID; }
    if (n_rows % h->n_probs) { mpc_set_err("combine_shards_exact: n_rows is not a multiple of the resident problems"); return MPC_ERR_INVALID; }
    for (uint32_t p = 0; p < h->n_probs; p++)
        if ((uint32_t)h->depth[p] > depth) { mpc_set_err("combine_shards_exact: a resident problem is deeper than the records"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (!h->d_tied_rows) {
        HIP_TRY(hipMalloc(&h->d_tied_rows, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(h->d_tied_rows, 0, sizeof(uint32_t), h->stream));
    }
    /* only fast-mode winners need re-scoring: strict-mode records already carry the reference's arithmetic.  The
     * records' precision and selection are the CALLER'S to say (a batch is combined long after the handle's last run,
     * which may have been of another kind) */
    const double rel = precision == MPC_PREC_FAST ? MPC_REFINE_REL : -1.0;
    hipLaunchKernelGGL(combine_exact_kernel, dim3(n_rows), dim3(64), 0, h->stream, h->d_probs, h->n_probs, selection, d_gathered,
                       n_shards, n_rows, (uint32_t)MPC_WINNER_RECORD_WORDS(depth), ordinal_stride, rel, d_out, h->d_tied_rows);
    HIP_TRY(hipGetLastError());
    if (h_tied_rows) {                       /* optional read-back (synchronises): rows with a cross-shard near-tie so far */
        HIP_TRY(hipMemcpyAsync(h->h_pair, h->d_tied_rows, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(h_tied_rows, h->h_pair, sizeof(uint32_t));
    }
    return MPC_OK;
}

static int solve_common(mpc_hip_handle *h, const mpc_problem *probs, const uint64_t *seeds, uint32_t n_problems,
                        bool seeded, const float *d_rud, const float *d_thr, uint64_t first, uint64_t count,
                        int precision, int selection, double *d_cost_last, double *d_cost_cum, mpc_result *results)
{
    if (!results) { mpc_set_err("result pointer is NULL"); return MPC_ERR_INVALID; }
    if (int s = mpc_hip_upload_problems(h, probs, seeds, n_problems)) return s;
    if (int s = run_common(h, seeded, d_rud, d_thr, first, count, precision, selection, d_cost_last, d_cost_cum)) return s;
    int s = fetch_internal(h, results, n_problems);
    if (s == MPC_INTERNAL_RERUN_STRICT) {
        /* the reported costs stay the fast ones; only the decision is redone (into the library's own buffers) */
        if (int s2 = run_common(h, seeded, d_rud, d_thr, first, count, MPC_PREC_F64, selection, nullptr, nullptr)) return s2;
        s = fetch_internal(h, results, n_problems);
    }
    if (s == MPC_OK && n_problems == 1 && results[0].status != MPC_OK) {
        /* one problem: its status is the call's status (a batch reports per problem in results[p].status) */
        if (results[0].status == MPC_ERR_NO_ITERATION) mpc_set_err("no candidate has a finite score: nothing is below DBL_MAX (controller.cpp:81,160)");
        else mpc_set_err("the chosen control is not finite (controller.cpp:201-202)");
        return results[0].status;
    }
    return s;
}

extern "C" int mpc_hip_eval_candidates(mpc_hip_handle *h, const mpc_problem *h_problem, const float *d_rudders,
                                       const float *d_throttles, uint64_t n, int precision, int selection,
                                       double *d_cost_last, double *d_cost_cum, mpc_result *h_result)
{
    if (!h_problem) { mpc_set_err("problem is NULL"); return MPC_ERR_INVALID; }
    if (int s = mpc_check_problem(h_problem, precision)) return s;
    return solve_common(h, h_problem, nullptr, 1, false, d_rudders, d_throttles, 0, n, precision, selection,
                        d_cost_last, d_cost_cum, h_result);
}

extern "C" int mpc_hip_eval_seeded(mpc_hip_handle *h, const mpc_problem *h_problem, uint64_t seed,
                                   uint64_t first_ordinal, uint64_t count, int precision, int selection,
                                   double *d_cost_last, double *d_cost_cum, mpc_result *h_result)
{
    if (!h_problem) { mpc_set_err("problem is NULL"); return MPC_ERR_INVALID; }
    if (int s = mpc_check_problem(h_problem, precision)) return s;
    return solve_common(h, h_problem, &seed, 1, true, nullptr, nullptr, first_ordinal, count, precision, selection,
                        d_cost_last, d_cost_cum, h_result);
}

extern "C" int mpc_hip_eval_seeded_batch(mpc_hip_handle *h, const mpc_problem *h_problems, const uint64_t *h_seeds,
                                         uint32_t n_problems, uint64_t count, int precision, int selection,
                                         double *d_cost_last, double *d_cost_cum, mpc_result *h_results)
{
    if (!h_problems || !h_seeds) { mpc_set_err("problems/seeds are NULL"); return MPC_ERR_INVALID; }
    for (uint32_t p = 0; p < n_problems; p++)
        if (int s = mpc_check_problem(&h_problems[p], precision)) return s;
    return solve_common(h, h_problems, h_seeds, n_problems, true, nullptr, nullptr, 0, count, precision, selection,
                        d_cost_last, d_cost_cum, h_results);
}

extern "C" int mpc_hip_rollout(mpc_hip_handle *h, const mpc_problem *h_problem, const double *h_rudders,
                               const double *h_throttles, int precision, mpc_result *h_result)
{
    if (!h || !h_problem || !h_rudders || !h_throttles || !h_result) { mpc_set_err("rollout: NULL argument"); return MPC_ERR_INVALID; }
    if (precision != MPC_PREC_FAST && precision != MPC_PREC_F64) { mpc_set_err("unknown precision"); return MPC_ERR_INVALID; }
    if (int s = mpc_check_problem(h_problem, precision)) return s;
    if (precision == MPC_PREC_FAST) if (int s = mpc_check_fast_model(h, 1)) return s;
    if (int s = mpc_hip_upload_problems(h, h_problem, nullptr, 1)) return s;
    double ctl[2 * MPC_MAX_DEPTH] = {0};
    for (int l = 0; l < h_problem->depth; l++) { ctl[l] = h_rudders[l]; ctl[MPC_MAX_DEPTH + l] = h_throttles[l]; }
    HIP_TRY(hipMemcpyAsync(h->d_controls, ctl, sizeof ctl, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));      /* ctl is a stack buffer */
    hipEvent_t *ev = &h->ev[4 * (h->runs % TIMING_RING)];
    h->timed[h->runs % TIMING_RING] = 1;
    h->total_from_k1[h->runs % TIMING_RING] = 0;
    HIP_TRY(hipEventRecord(ev[0], h->stream));
    hipLaunchKernelGGL(rollout_one_kernel, dim3(1), dim3(64), 0, h->stream, h->d_probs, h->d_controls, precision, h->d_results);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[1], h->stream));
    HIP_TRY(hipEventRecord(ev[3], h->stream));
    h->runs++;
    h->last_count = 1; h->last_first = 0; h->last_precision = MPC_PREC_F64; h->last_selection = 0; h->ran = true;
    return fetch_internal(h, h_result, 1);
}
