/*
 * mpc_plan.hip -- reference trajectories sampled on the device.
 *
 * Replaces, for batches of problems, the host loop of controller.cpp:96-98
 *     for (i < maxIterations) { s.time() = startTime + (i + 1) * c_ScoringTimeStep; referenceTrajectory.sample(s); }
 * i.e. path_planner_common's DubinsPlan::sample (external to the reference repository: defined as SURVEY.md 8c
 * records it -- A. Walker's Dubins-Curves segment forms at arclength (t - startTime) * speed, compass heading =
 * wrap(pi/2 - yaw), wrapper speed; times outside the plan clamp to its ends, a gap holds the earlier end).
 * One lane per (problem, level) writes the checkpoint row straight into the device problem record: x, y, heading,
 * speed, sin, cos and the current-corrected position the fast kernels use.  FP64, ocml sin/cos.
 */
#include "mpc_internal.h"

using namespace mpc;

__device__ static inline double mod2pi(double a) { return a - TWO_PI * floor(a / TWO_PI); }

enum { SEG_L = 0, SEG_S = 1, SEG_R = 2 };

/* dubins.h DubinsPathType order LSL, LSR, RSL, RSR, RLR, LRL -> the three segment kinds */
__device__ static inline int segment_kind(int word, int k)
{
    const int first = (word == 0 || word == 1 || word == 5) ? SEG_L : SEG_R;
    const int last = (word == 0 || word == 2 || word == 5) ? SEG_L : SEG_R;
    const int mid = word < 4 ? SEG_S : (word == 4 ? SEG_L : SEG_R);
    return k == 0 ? first : (k == 1 ? mid : last);
}

/* unit-radius closed forms: a left arc, a straight, a right arc of length t from pose qi */
__device__ static inline void dubins_segment(double t, const double qi[3], double qt[3], int kind)
{
    const double st = sin(qi[2]), ct = cos(qi[2]);
    if (kind == SEG_L) { qt[0] = sin(qi[2] + t) - st; qt[1] = -cos(qi[2] + t) + ct; qt[2] = t; }
    else if (kind == SEG_R) { qt[0] = -sin(qi[2] - t) + st; qt[1] = cos(qi[2] - t) - ct; qt[2] = -t; }
    else { qt[0] = ct * t; qt[1] = st * t; qt[2] = 0.0; }
    qt[0] += qi[0]; qt[1] += qi[1]; qt[2] += qi[2];
}

__device__ static void sample_wrapper(const mpc_dubins &w, double arclen, double q[3])
{
    const double tp = arclen / w.rho;
    const double qi[3] = {0.0, 0.0, w.qi[2]};
    double q1[3], q2[3];
    const double p1 = w.param[0], p2 = w.param[1];
    dubins_segment(p1, qi, q1, segment_kind(w.type, 0));
    dubins_segment(p2, q1, q2, segment_kind(w.type, 1));
    if (tp < p1) dubins_segment(tp, qi, q, segment_kind(w.type, 0));
    else if (tp < p1 + p2) dubins_segment(tp - p1, q1, q, segment_kind(w.type, 1));
    else dubins_segment(tp - p1 - p2, q2, q, segment_kind(w.type, 2));
    q[0] = q[0] * w.rho + w.qi[0];
    q[1] = q[1] * w.rho + w.qi[1];
    q[2] = mod2pi(q[2]);
}

__global__ __launch_bounds__(64) void sample_plans_kernel(DevProblem *__restrict__ probs, const mpc_plan *__restrict__ plans,
                                                          const double *__restrict__ start_times, uint32_t n_problems)
{
    const uint32_t p = blockIdx.x;
    const int l = threadIdx.x;
    if (p >= n_problems) return;
    DevProblem &P = probs[p];
    if (l >= P.depth) return;
    const mpc_plan &plan = plans[p];
    const int n = plan.n_segments;
    const double time = start_times[p] + (double)(l + 1) * 1.0;           /* controller.cpp:97, c_ScoringTimeStep = 1 */
    int k = -1;
    for (int i = 0; i < n; i++)
        if (plan.segments[i].start_time <= time && time <= plan.segments[i].end_time) { k = i; break; }
    double t = time;
    if (k < 0) {
        if (t < plan.segments[0].start_time) { k = 0; t = plan.segments[0].start_time; }
        else {
            k = n - 1; t = plan.segments[n - 1].end_time;
            for (int i = 0; i + 1 < n; i++)
                if (time > plan.segments[i].end_time && time < plan.segments[i + 1].start_time) { k = i; t = plan.segments[i].end_time; break; }
        }
    }
    const mpc_dubins &w = plan.segments[k];
    double q[3];
    sample_wrapper(w, (t - w.start_time) * w.speed, q);
    const double heading = mod2pi(PI_2 - q[2]);
    double level_time = 0;
    for (int s = 0; s < P.nsub; s++) level_time += P.sub[s * SUB_STRIDE];
    double *c = P.cp + l * CP_STRIDE;
    c[0] = q[0]; c[1] = q[1]; c[2] = heading; c[3] = w.speed;
    c[4] = sin(heading); c[5] = cos(heading);
    c[6] = q[0] - P.cx * (level_time * (l + 1)); c[7] = q[1] - P.cy * (level_time * (l + 1));
}

extern "C" int mpc_hip_upload_problems_from_plans(mpc_hip_handle *h, const mpc_problem *h_problems, const uint64_t *h_seeds,
                                                  const mpc_plan *h_plans, const double *h_start_times, uint32_t n_problems)
{
    if (!h_plans || !h_start_times) { mpc_set_err("upload_problems_from_plans: plans/start_times are NULL"); return MPC_ERR_INVALID; }
    for (uint32_t p = 0; p < n_problems; p++) {
        const mpc_plan &pl = h_plans[p];
        if (pl.n_segments < 1) { mpc_set_err("Cannot run MPC on an empty plan"); return MPC_ERR_EMPTY_PLAN; }   /* controller.cpp:37 */
        if (pl.n_segments > MPC_MAX_PLAN_SEGMENTS) { mpc_set_err("plan has more than MPC_MAX_PLAN_SEGMENTS segments"); return MPC_ERR_TOO_LARGE; }
        for (int i = 0; i < pl.n_segments; i++) {
            const mpc_dubins &w = pl.segments[i];
            if (w.type < 0 || w.type > 5 || !(w.rho > 0) || !(w.speed > 0) || !(w.end_time >= w.start_time)) {
                mpc_set_err("plan segment: unknown Dubins word, or rho/speed/end time not positive");
                return MPC_ERR_INVALID;
            }
        }
    }
    if (int s = mpc_hip_upload_problems(h, h_problems, h_seeds, n_problems)) return s;
    const size_t need = sizeof(mpc_plan) * n_problems + sizeof(double) * n_problems;
    if (need > h->cap_plans) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_plans); h->d_plans = nullptr; h->cap_plans = 0;
        HIP_TRY(hipMalloc(&h->d_plans, need));
        h->cap_plans = need;
    }
    mpc_plan *d_plans = (mpc_plan *)h->d_plans;
    double *d_t0 = (double *)((char *)h->d_plans + sizeof(mpc_plan) * n_problems);
    /* pageable host memory: the runtime stages the copy before hipMemcpyAsync returns, the caller keeps no obligation */
    HIP_TRY(hipMemcpyAsync(d_plans, h_plans, sizeof(mpc_plan) * n_problems, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_t0, h_start_times, sizeof(double) * n_problems, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(sample_plans_kernel, dim3(n_problems), dim3(64), 0, h->stream, h->d_probs, d_plans, d_t0, n_problems);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));      /* h_plans / h_start_times are the caller's again */
    return MPC_OK;
}

extern "C" int mpc_hip_fetch_checkpoints(mpc_hip_handle *h, uint32_t problem, mpc_checkpoint *h_out)
{
    if (!h || !h_out) { mpc_set_err("fetch_checkpoints: NULL argument"); return MPC_ERR_INVALID; }
    if (problem >= h->n_probs) { mpc_set_err("fetch_checkpoints: no such problem"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    double cp[MPC_MAX_DEPTH * CP_STRIDE];
    HIP_TRY(hipMemcpyAsync(cp, h->d_probs[problem].cp, sizeof cp, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int l = 0; l < h->depth[problem]; l++)
        h_out[l] = mpc_checkpoint{cp[l * CP_STRIDE], cp[l * CP_STRIDE + 1], cp[l * CP_STRIDE + 2], cp[l * CP_STRIDE + 3]};
    return MPC_OK;
}
