/*
 * fp32_probe.hip -- evidence for DESIGN.md section 4 ("why the fast mode is not FP32"), not part of the product.
 *
 * The fast model's algebra (mpc_model.h: fast_substep / fast_level_coef / fast_score, default hull, uniform schedule)
 * written once over two types -- R for the dynamics (speed, rpm slew, heading vector, cost terms), P for the position
 * accumulators -- and run on BASELINE config 2 from tools/fp32_error.py, which compares every candidate's cost with
 * the CPU oracle and times the kernel:
 *     <double, double>          the shape of the shipped fast kernel (sanity row: must land at ~1e-7)
 *     <float,  float>           all-FP32 state, BASELINE.json's "FP32 mode" taken literally
 *     <float,  float + Kahan>   FP32 dynamics, compensated FP32 position
 *     <float,  double>          FP32 dynamics, FP64 position and distance/exp term
 * One lane per candidate, checkpoints in LDS, controls read coalesced per level, cost written; no snapshots and no
 * argmin (which only flatters the FP32 variants' times).
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

struct ProbeProblem {
    double sx, sy, sh, sv, srpm, wd, wh, ws, dt, cx, cy;
    int32_t depth, nsub;
    double cp[64 * 8];           /* per level: x, y, heading, speed, sin h, cos h, x - cx T, y - cy T */
};

template <typename R> __device__ inline R rsq_(R x);
template <> __device__ inline float rsq_<float>(float x) { return __builtin_amdgcn_rsqf(x); }
template <> __device__ inline double rsq_<double>(double x) { return __builtin_amdgcn_rsq(x); }
template <typename R> __device__ inline R fma_(R a, R b, R c);
template <> __device__ inline float fma_<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ inline double fma_<double>(double a, double b, double c) { return fma(a, b, c); }
template <typename R> __device__ inline R exp_(R x);
template <> __device__ inline float exp_<float>(float x) { return expf(x); }
template <> __device__ inline double exp_<double>(double x) { return exp(x); }

/* sqrt by rsq + one coupled Newton step, as fast_sqrt */
template <typename R> __device__ inline R sqrt_(R a)
{
    a = a > (R)1e-30 ? a : (R)1e-30;
    const R y = rsq_<R>(a);
    R g = a * y;
    const R hh = (R)0.5 * y;
    const R r = fma_<R>(-hh, g, (R)0.5);
    g = fma_<R>(g, r, g);
    const R e = fma_<R>(-g, g, a);
    return fma_<R>(e, hh, g);
}

/* position accumulator: plain (P = float or double) or Kahan-compensated float */
template <typename P, bool KAHAN> struct Acc {
    P s, c;
    __device__ inline void init(double v) { s = (P)v; c = 0; }
    template <typename R> __device__ inline void add(R step, R dir)
    {
        if (KAHAN) { const P y = (P)(step * dir) - c; const P t = s + y; c = (t - s) - y; s = t; }
        else s = (P)fma_<P>((P)step, (P)dir, s);
    }
    __device__ inline double value() const { return (double)s; }
};

template <typename R, typename P, bool KAHAN>
__global__ __launch_bounds__(64) void probe_kernel(const ProbeProblem *__restrict__ prob, const float *__restrict__ rud,
                                                   const float *__restrict__ thr, uint64_t n, double *__restrict__ cost_last)
{
    __shared__ double cpL[64 * 8];
    const ProbeProblem &Q = *prob;
    for (int i = threadIdx.x; i < Q.depth * 8; i += blockDim.x) cpL[i] = Q.cp[i];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    /* the reference's hull (VehicleState.h:86-101) */
    const double MAX_RPM = 3200.0, C_PROP = 0.389105058 / 20.0, MAX_SPEED = 2.75, MASS = 2000.0;
    const double MAX_PROP_SPEED = 3200.0 * 0.389105058 / 20.0, MAX_FORCE = 8948.4 / MAX_SPEED;
    const double KP = MAX_FORCE / (MAX_PROP_SPEED * MAX_PROP_SPEED - MAX_SPEED * MAX_SPEED);
    const double KD = MAX_FORCE / (MAX_SPEED * MAX_SPEED * MAX_SPEED);
    const R dt = (R)Q.dt, max_step = (R)(1000.0 * Q.dt), dtm = (R)(Q.dt / MASS), inv_dt = (R)(1.0 / Q.dt);
    R v = (R)Q.sv, rpm = (R)Q.srpm, sh = (R)sin(Q.sh), ch = (R)cos(Q.sh);
    /* positions relative to the start (what an FP32 implementation would do first: SURVEY.md section 7 hard part 2) and in
     * units of dt while the substeps are equally long, as the shipped kernel carries them */
    Acc<P, KAHAN> x, y;
    x.init(0.0); y.init(0.0);
    double last = 0;
    for (int l = 0; l < Q.depth; l++) {
        R r = (R)rud[(uint64_t)l * n + i], t = (R)thr[(uint64_t)l * n + i];
        r = r < (R)-1 ? (R)-1 : (r > (R)1 ? (R)1 : r);
        t = t < (R)0 ? (R)0 : (t > (R)1 ? (R)1 : t);
        const R ang = r * (R)(M_PI / 6), a2r = ang * ang;
        /* Taylor sin/cos of the rudder angle (|x| <= pi/6), enough terms for the type */
        R ps_ = (R)(-1.0 / 6227020800.0), pc_ = (R)(-1.0 / 87178291200.0);
        pc_ = fma_<R>(pc_, a2r, (R)(1.0 / 479001600.0)); ps_ = fma_<R>(ps_, a2r, (R)(1.0 / 39916800.0));
        pc_ = fma_<R>(pc_, a2r, (R)(-1.0 / 3628800.0)); ps_ = fma_<R>(ps_, a2r, (R)(-1.0 / 362880.0));
        pc_ = fma_<R>(pc_, a2r, (R)(1.0 / 40320.0)); ps_ = fma_<R>(ps_, a2r, (R)(1.0 / 5040.0));
        pc_ = fma_<R>(pc_, a2r, (R)(-1.0 / 720.0)); ps_ = fma_<R>(ps_, a2r, (R)(-1.0 / 120.0));
        pc_ = fma_<R>(pc_, a2r, (R)(1.0 / 24.0)); ps_ = fma_<R>(ps_, a2r, (R)(1.0 / 6.0));
        pc_ = fma_<R>(pc_, a2r, (R)-0.5);
        const R sin_ra = fma_<R>(-(ang * a2r), ps_, ang), cos_ra = fma_<R>(pc_, a2r, (R)1);
        const R target = t * (R)MAX_RPM;
        const R kpcm = (R)KP * cos_ra * dtm, dm = (R)KD * dtm, kyaw_dt = (R)0.125 * sin_ra * dt;
        const R e = rpm - target;
        R m = e < 0 ? -e : e;
        const R sc = e < 0 ? (R)-C_PROP : (R)C_PROP, sgn = e < 0 ? (R)-1 : (R)1;
        const R pst = fma_<R>((R)C_PROP * target, (R)1 + (sizeof(R) == 4 ? (R)1e-6 : (R)0x1p-50), (R)1e-30);
#pragma unroll 10
        for (int k = 0; k < Q.nsub; k++) {
            m = m - max_step; m = m > 0 ? m : (R)0;
            const R ps = fma_<R>(sc, m, pst);
            const R y0 = rsq_<R>(ps);
            const R g = ps * y0, hh = (R)0.5 * y0;
            const R rr = fma_<R>(-hh, g, (R)0.5);
            R rs = fma_<R>(g, rr, g);
            rs = rs > v ? rs : v;
            const R a = kyaw_dt * rs;
            const R v2 = v * v, w = fma_<R>(dm, v, kpcm), kps = kpcm * ps;
            v = fma_<R>(kps, ps, fma_<R>(-v2, w, v));
            const R a2 = a * a;
            const R sa = fma_<R>(a * a2, fma_<R>(a2, (R)(1.0 / 120), (R)(-1.0 / 6)), a);
            const R cm = a2 * fma_<R>(a2, (R)(1.0 / 24), (R)-0.5);
            const R nsh = fma_<R>(sh, cm, fma_<R>(ch, sa, sh)), nch = fma_<R>(ch, cm, fma_<R>(-sh, sa, ch));
            sh = nsh; ch = nch;
            x.template add<R>(v, nsh);
            y.template add<R>(v, nch);
        }
        rpm = fma_<R>(sgn, m, target);
        /* tracking cost (fast_score): heading and speed terms in R; the distance term in the position's type */
        const double *cp = cpL + l * 8;
        const R delta = v * dt;
        const R gdx = fma_<R>(delta, sh, (R)(Q.cx * Q.dt)), gdy = fma_<R>(delta, ch, (R)(Q.cy * Q.dt));
        R ddx = gdx, ddy = gdy;
        if (ddx == 0 && ddy == 0) ddx = 1;
        const R cross = ddx * (R)cp[5] - ddy * (R)cp[4], dot = fma_<R>(ddy, (R)cp[5], ddx * (R)cp[4]);
        const R hd = (R)fabsf(atan2f((float)cross, (float)dot));
        const R sog = sqrt_<R>(fma_<R>(gdx, gdx, gdy * gdy)) * inv_dt;
        const R sd = (R)cp[3] - sog;
        if (sizeof(P) == 8) {
            const double ex = (cp[6] - Q.sx) - x.value() * Q.dt, ey = (cp[7] - Q.sy) - y.value() * Q.dt;
            const double dist = sqrt(ex * ex + ey * ey);
            last = Q.wd * exp(dist) + (double)((R)Q.wh * hd + (R)Q.ws * (sd < 0 ? -sd : sd));
        } else {
            const R ex = (R)(cp[6] - Q.sx) - (R)x.value() * dt, ey = (R)(cp[7] - Q.sy) - (R)y.value() * dt;
            const R dist = sqrt_<R>(fma_<R>(ex, ex, ey * ey));
            last = (double)fma_<R>((R)Q.wd, exp_<R>(dist), fma_<R>((R)Q.wh, hd, (R)Q.ws * (sd < 0 ? -sd : sd)));
        }
    }
    cost_last[i] = last;
}

/* variant: 0 = <double,double>, 1 = <float,float>, 2 = <float,float+Kahan>, 3 = <float,double>.  Returns the mean
 * kernel time in microseconds over `reps` launches (HIP events around the whole batch, stream 0), or < 0 on error. */
extern "C" double fp32_probe_run(int variant, const ProbeProblem *d_prob, const float *d_rud, const float *d_thr,
                                 uint64_t n, double *d_cost_last, int reps)
{
    const dim3 grid((unsigned)((n + 63) / 64)), blk(64);
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    for (int rep = -3; rep < reps; rep++) {
        if (rep == 0) (void)hipEventRecord(a, 0);
        switch (variant) {
        case 0: hipLaunchKernelGGL((probe_kernel<double, double, false>), grid, blk, 0, 0, d_prob, d_rud, d_thr, n, d_cost_last); break;
        case 1: hipLaunchKernelGGL((probe_kernel<float, float, false>), grid, blk, 0, 0, d_prob, d_rud, d_thr, n, d_cost_last); break;
        case 2: hipLaunchKernelGGL((probe_kernel<float, float, true>), grid, blk, 0, 0, d_prob, d_rud, d_thr, n, d_cost_last); break;
        case 3: hipLaunchKernelGGL((probe_kernel<float, double, false>), grid, blk, 0, 0, d_prob, d_rud, d_thr, n, d_cost_last); break;
        default: return -2;
        }
    }
    (void)hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess || hipGetLastError() != hipSuccess) return -3;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 1e3 * ms / reps;
}
