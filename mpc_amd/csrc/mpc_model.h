/*
 * mpc_model.h -- device functions of the vehicle model and tracking cost, in two precisions.
 *
 *   strict_*  FP64, the reference's statements in the reference's order (VehicleState.h:111-148,
 *             controller.cpp:231-241).  Built with -ffp-contract=off, IEEE sqrt, correctly rounded divisions
 *             (div_known), ocml libm.
 *   fast_*    FP64 state, algebraically streamlined for the CDNA4 vector ALU (FP64 at half the FP32 rate,
 *             transcendentals at quarter rate): no sin/cos/fmod/divide per substep, ~36 FP64 instructions.
 *
 * Shared per-problem data lives in LDS (staged once per workgroup by the kernels):
 *   cp[l*CP_STRIDE + k]   k: 0 x, 1 y, 2 heading, 3 speed, 4 sin(heading), 5 cos(heading),
 *                            6 x - cx*T_l, 7 y - cy*T_l   (T_l = time from the start to checkpoint l)
 *   sub[s*SUB_STRIDE + k] k: 0 dt, 1 1000*dt, 2 dt/2000, 3 cx*dt, 4 cy*dt, 5 1/dt
 */
#ifndef MPC_MODEL_H
#define MPC_MODEL_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MPC_MODEL_FN __host__ __device__
#else
#define MPC_MODEL_FN                      /* host-only build (g++): the strict model is shared with mpc_amd/host */
#endif
#include <math.h>
#include <stdint.h>

#define CP_STRIDE 8
#define SUB_STRIDE 6

/* Host builds of the strict model keep IEEE semantics whatever the including project's flags are: the reference builds
 * with -funsafe-math-optimizations (CMakeLists.txt:4), which lets g++ reassociate sums and turn divisions into
 * multiplications by reciprocals -- the host VehicleState of mpc_amd/host would then differ from the device's strict
 * kernels (and from the oracle) in the last bits.  Functions compiled under these options are not inlined into
 * callers compiled under others, so the arithmetic below stays as written. */
#if !defined(__HIPCC__) && defined(__GNUC__) && !defined(__clang__)
#pragma GCC push_options
#pragma GCC optimize("no-unsafe-math-optimizations", "no-fast-math", "fp-contract=off")
#define MPC_MODEL_HOST_STRICT_OPTIONS 1
#endif

namespace mpc {

constexpr double PI = 3.14159265358979323846;
constexpr double PI_2 = 1.57079632679489661923;
constexpr double TWO_PI = PI * 2;
constexpr double THREE_PI = 3 * PI;

/* The boat's characteristics: the 11 model parameters of VehicleState.h:86-96 (compile-time constants in the reference,
 * marked "should allow setting of boat characteristics", :84) and what estimate() derives from them (:98-101), plus
 * the reciprocals and products the device code multiplies by.  Per problem, in the problem record (a fleet may mix
 * hulls); every access is wave-uniform.  make_model() is the ONLY place the derived fields are computed, on the host,
 * in IEEE double with the reference's expressions: the default hull reproduces the reference's constants bit for bit. */
struct Model {
    double idle_rpm, max_rpm, max_rpm_change_rate, prop_ratio, prop_pitch, max_rudder_angle, rudder_coefficient,
           rudder_distance, mass, max_power, max_speed;
    double prop_coefficient, drag_coefficient;          /* VehicleState.h:100-101 */
    double rpm_span;                                    /* max_rpm - idle_rpm */
    double inv_prop_pitch, inv_rudder_distance, inv_mass;   /* RN(1/x): div_known */
    double c_prop;                                      /* prop_ratio / prop_pitch: prop speed per rpm */
    double kyaw;                                        /* rudder_coefficient / rudder_distance */
    double pad0;                                        /* even number of doubles: the staging vectors behind it stay 16-byte aligned */
};

MPC_MODEL_FN inline Model make_model(double idle_rpm, double max_rpm, double max_rpm_change_rate, double prop_ratio,
                                     double prop_pitch, double max_rudder_angle, double rudder_coefficient,
                                     double rudder_distance, double mass, double max_power, double max_speed)
{
    Model m{};
    m.idle_rpm = idle_rpm; m.max_rpm = max_rpm; m.max_rpm_change_rate = max_rpm_change_rate; m.prop_ratio = prop_ratio;
    m.prop_pitch = prop_pitch; m.max_rudder_angle = max_rudder_angle; m.rudder_coefficient = rudder_coefficient;
    m.rudder_distance = rudder_distance; m.mass = mass; m.max_power = max_power; m.max_speed = max_speed;
    const double max_prop_speed = (max_rpm * prop_ratio) / prop_pitch;                     /* :98 */
    const double max_force = max_power / max_speed;                                        /* :99 */
    m.prop_coefficient = max_force / (max_prop_speed * max_prop_speed - max_speed * max_speed);   /* :100 */
    m.drag_coefficient = max_force / (max_speed * max_speed * max_speed);                  /* :101 */
    m.rpm_span = max_rpm - idle_rpm;
    m.inv_prop_pitch = 1.0 / prop_pitch; m.inv_rudder_distance = 1.0 / rudder_distance; m.inv_mass = 1.0 / mass;
    m.c_prop = prop_ratio / prop_pitch;
    m.kyaw = rudder_coefficient / rudder_distance;
    return m;
}

/* the reference's hull, VehicleState.h:86-96 */
MPC_MODEL_FN inline Model default_model()
{
    return make_model(0.0, 3200.0, 1000.0, 0.389105058, 20.0, PI / 6, 0.25, 2.0, 2000.0, 8948.4, 2.75);
}

struct Weights { double wd, wh, ws; };

/* ======================================= strict FP64 ================================================== */
struct Veh { double x, y, h, v, rpm, cmg, sog; };

/* VehicleState.h:20-22 */
MPC_MODEL_FN inline double rpm_from_speed(double speed, const Model &M)
{
    return M.idle_rpm + (speed / M.max_speed) * (M.max_rpm - M.idle_rpm);
}

/* a / b on the device as q = a*y, q + (a - b*q)*y with y = RN(1/b): three instructions instead of the ~25 of the IEEE
 * divide sequence.  With a correctly rounded reciprocal the result is the correctly rounded quotient for every
 * dividend except at most one significand per divisor (Brisebarre, Muller, Raina, "Accelerating correctly rounded
 * floating-point division when the divisor is known in advance", IEEE TC 2004); 2.4e9 random dividends against each of
 * the divisors used here (20, 2000, the BASELINE and reference-native substep lengths) showed no difference.
 * Non-finite dividends give NaN instead of +-inf (the next substep makes both NaN anyway).  The host build divides. */
MPC_MODEL_FN inline double div_known(double a, double b, double y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
#else
    (void)y;
    return a / b;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
/* sin and cos of a heading in [0, 2 pi) (what estimate()'s fmod leaves, VehicleState.h:129-132), branch-free: quadrant
 * by Cody-Waite reduction with a two-part pi/2 (exact to ~2^-106 for k <= 4), then the fdlibm kernels on |r| <= pi/4
 * (Sun's __kernel_sin / __kernel_cos minimax coefficients, < 1 ulp).  The library sincos carries a large-argument
 * path behind a branch, which keeps the substep loop from being scheduled as one block; this is ~35 instructions and
 * none.  Outside [0, 2 pi) (never produced by the model) the result is still sin/cos to ~1e-16 up to |h| ~ 1e5. */
__device__ inline void heading_sincos(double h, double &sin_h, double &cos_h)
{
    const double kd = __builtin_rint(h * 0.63661977236758138243);           /* 2/pi */
    double r = fma(-kd, 1.57079632679489655800e+00, h);                     /* RN(pi/2) */
    r = fma(-kd, 6.12323399573676603587e-17, r);                            /* pi/2 - RN(pi/2) */
    const int k = (int)kd;
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double s = fma(r * z, ps, r);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const double a = (k & 1) ? c : s, b = (k & 1) ? s : c;
    sin_h = (k & 2) ? -a : a;
    cos_h = ((k + 1) & 2) ? -b : b;
}
#endif

/* One level: simulate()'s clamps (VehicleState.h:44-47) and n_sub estimate() calls (:111-148).
 * sin/cos of the rudder angle are loop invariant and cmg/sog are overwritten by every substep, so they are
 * evaluated once / for the last substep only: the values are the ones the reference ends the level with. */
/* STRIDE: distance between schedule rows; 0 = every substep uses row 0 (a uniform schedule held in registers). */
template <int STRIDE = SUB_STRIDE>
MPC_MODEL_FN inline void strict_level(Veh &s, double rudder, double throttle, const double *sub, int n_sub,
                                    double cx, double cy, const Model &M)
{
    if (rudder < -1) rudder = -1; else if (rudder > 1) rudder = 1;
    if (throttle < 0) throttle = 0; else if (throttle > 1) throttle = 1;
    const double target_rpm = M.idle_rpm + throttle * (M.max_rpm - M.idle_rpm);
    const double rudder_angle = rudder * M.max_rudder_angle;
    const double cos_ra = cos(rudder_angle), sin_ra = sin(rudder_angle);
    double dx = 0, dy = 0, d_time = 1;
    for (int k = 0; k < n_sub; k++) {
        d_time = sub[k * STRIDE];
        /* 1/dt: the schedule rows carry it (host-computed, correctly rounded); a bare dt (STRIDE 0) is divided once */
        const double inv_dt = STRIDE == SUB_STRIDE ? sub[k * STRIDE + 5] : 1.0 / d_time;
        double rcr = div_known(target_rpm - s.rpm, d_time, inv_dt);
        if (rcr > M.max_rpm_change_rate) rcr = M.max_rpm_change_rate;
        else if (rcr < -M.max_rpm_change_rate) rcr = -M.max_rpm_change_rate;
        s.rpm += rcr * d_time;

        const double prop_rpm = M.prop_ratio * s.rpm;
        const double prop_speed = div_known(prop_rpm, M.prop_pitch, M.inv_prop_pitch);
        const double rudder_speed = fmax(sqrt(prop_speed), s.v);
        const double thrust = M.prop_coefficient * (prop_speed * prop_speed - s.v * s.v);
        const double thrust_fwd = thrust * cos_ra;
        const double rudder_speed_yaw = rudder_speed * sin_ra;
        /* (coefficient * x) / distance; by the reference's 2.0 the reciprocal form is exact */
        const double yaw_rate = div_known(M.rudder_coefficient * rudder_speed_yaw, M.rudder_distance, M.inv_rudder_distance);

        /* heading = fmod(heading + yaw_rate*dt, 2pi); if (< 0) += 2pi.  For an argument in (-2pi, 4pi) -- every
         * substep of a real schedule -- fmod is an exact conditional subtraction (Sterbenz) or the identity, written as
         * selects; anything else (and NaN) takes the one rare branch to fmod itself first. */
        double h = s.h + yaw_rate * d_time;
        if (!(h > -TWO_PI && h < 2 * TWO_PI)) h = fmod(h, TWO_PI);
        h = h >= TWO_PI ? h - TWO_PI : h;
        h = h < 0 ? h + TWO_PI : h;
        s.h = h;

        const double speed = s.v;
        const double drag = speed * speed * speed * M.drag_coefficient;
        s.v += div_known(thrust_fwd - drag, M.mass, M.inv_mass) * d_time;
        const double delta = s.v * d_time;
#if defined(__HIP_DEVICE_COMPILE__)
        double sin_h, cos_h;
        heading_sincos(h, sin_h, cos_h);
#else
        const double sin_h = sin(h), cos_h = cos(h);
#endif
        dx = delta * sin_h + cx * d_time;
        dy = delta * cos_h + cy * d_time;
        s.x += dx;
        s.y += dy;
    }
    if (n_sub > 0) {
        s.cmg = PI_2 - atan2(dy, dx);
        s.sog = sqrt(dx * dx + dy * dy) / d_time;
    }
}

/* path_planner_common State::headingDifference as SURVEY.md section 8c defines it (external; unpinned) */
MPC_MODEL_FN inline double heading_difference(double heading, double other)
{
    return fmod(fmod(other - heading, TWO_PI) + THREE_PI, TWO_PI) - PI;
}

/* controller.cpp:231-241 */
MPC_MODEL_FN inline double strict_score(const double *cp, const Veh &s, const Weights &w)
{
    const double heading_diff = fabs(heading_difference(cp[2], s.cmg));
    const double speed_diff = fabs(cp[3] - s.sog);
    const double dx = cp[0] - s.x;
    const double dy = cp[1] - s.y;
    const double d = exp(sqrt(dx * dx + dy * dy));
    return w.wd * d + w.wh * heading_diff + w.ws * speed_diff;
}

#if defined(__HIPCC__)
/* ========================================= fast (device only) ========================================== */
/* Heading is carried as (sh, ch) = (sin h, cos h).  x, y integrate the vessel's motion THROUGH THE WATER only:
 * the current's contribution sum(cx*dt) is the same for every candidate, so it is taken off the checkpoint on
 * the host (cp[6], cp[7]) instead of being added to every lane in every substep.  delta is the last substep's
 * v*dt, from which the last displacement over ground (for course made good / speed over ground) follows. */
struct FastVeh { double x, y, sh, ch, v, rpm, delta; };

/* sqrt(a), a >= 0: hardware reciprocal-sqrt estimate + one coupled Newton (Goldschmidt) step
 * with a residual correction (~1 ulp), for the scoring terms; a == 0 gives ~1e-154 instead of 0 */
__device__ inline double fast_sqrt(double a)
{
    a = fmax(a, 1.0e-300);
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y;
    const double hh = 0.5 * y;
    const double r = fma(-hh, g, 0.5);
    g = fma(g, r, g);
    const double e = fma(-g, g, a);
    return fma(e, hh, g);
}

/* a*b + c as ONE three-address v_fma_f64.  For a Horner step with a constant addend hipcc emits v_mov_b64 (copy the
 * constant) + v_fmac_f64 (accumulate into the copy).  On a lone wave those copies hide in dependent-issue bubbles; a
 * full SIMD is VALU-bound and pays for every one of them.  Not volatile: the scheduler may still move them. */
__device__ inline double fma3(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

/* The same with the addend in SCALAR registers: for a Horner step whose addend is a compile-time constant.  fma3 makes
 * hipcc materialise the constant in a VGPR pair with two v_mov_b32 per step -- THREE vector instructions per step, one
 * more than the compiler's own mov + fmac; a 64-bit literal cannot be encoded in VOP3, but one SGPR-pair source can, and
 * s_mov_b32 issues on the scalar unit beside the other waves' vector instructions: ONE vector instruction per step.
 * Same operation, same operands: the bits are fma3's. */
__device__ inline double fma3c(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

/* LEAN chooses between the two spellings of a Horner step: the compiler's (mov + fmac) or fma3.  While config 2 ran one
 * wave per SIMD the compiler's was faster there (36.1 us vs 36.7 us with fma3) and fma3 only where the SIMDs were full
 * (4 M candidates: -2 %, tree expansion: -3 %); now that a stream of solves runs two waves per SIMD, every device
 * kernel uses the lean one (MPC_FAST_LEAN in mpc_kernels.h) -- one spelling, so that a candidate's cost does not depend
 * on the size of the set it came in. */
template <bool LEAN>
__device__ inline double horner(double p, double x, double c) { return LEAN ? fma3c(p, x, c) : fma(p, x, c); }   /* c: always a literal */

/* exp(x) for the distance term, x >= 0: x = n ln2 + r, |r| <= ln2/2, Taylor to r^11 (3e-15 relative), scaled by 2^n.
 * ldexp overflows to +inf like exp does (exp(5000) = inf keeps "no finite score" what it is); NaN stays NaN.
 * 17 instructions against the library's ~38 (which also serve negative and subnormal results). */
__device__ inline double fast_exp_nonneg(double x)
{
    x = fmin(x, 1024.0);                                   /* keeps n in int range; exp(1024) = inf already */
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = fma(n, -6.93147180559945286227e-01, x);     /* RN(ln 2) */
    r = fma(n, -2.31904681384629955842e-17, r);            /* ln 2 - RN(ln 2) */
    double p = 1.0 / 39916800.0;
    p = fma3c(p, r, 1.0 / 3628800.0);
    p = fma3c(p, r, 1.0 / 362880.0);
    p = fma3c(p, r, 1.0 / 40320.0);
    p = fma3c(p, r, 1.0 / 5040.0);
    p = fma3c(p, r, 1.0 / 720.0);
    p = fma3c(p, r, 1.0 / 120.0);
    p = fma3c(p, r, 1.0 / 24.0);
    p = fma3c(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

/* Per-level constants of the fast model.  The rudder angle lies in [-pi/6, pi/6] (clamped rudder), so sin/cos need
 * no range reduction: Taylor polynomials to x^13 / x^14 (truncation < 1e-16 relative there) instead of the
 * library sincos with its reduction paths -- ~18 FMAs instead of ~70 instructions per level. */
struct FastLevelCoef { double target_rpm, kpc, kyaw; };   /* thrust_fwd = kpc * (ps^2 - v^2);  yaw_rate = kyaw * rudder_speed */

template <bool LEAN = false>
__device__ inline FastLevelCoef fast_level_coef(double rudder, double throttle, const Model &M)
{
    rudder = fmin(fmax(rudder, -1.0), 1.0);
    throttle = fmin(fmax(throttle, 0.0), 1.0);
    const double x = rudder * M.max_rudder_angle, x2 = x * x;
    /* the two chains alternate in the source: inline asm (LEAN) keeps its source order */
    double ps = -1.0 / 6227020800.0;                                   /* -1/13! */
    double pc = -1.0 / 87178291200.0;                                  /* -1/14! */
    pc = horner<LEAN>(pc, x2, 1.0 / 479001600.0);
    ps = horner<LEAN>(ps, x2, 1.0 / 39916800.0);
    pc = horner<LEAN>(pc, x2, -1.0 / 3628800.0);
    ps = horner<LEAN>(ps, x2, -1.0 / 362880.0);
    pc = horner<LEAN>(pc, x2, 1.0 / 40320.0);
    ps = horner<LEAN>(ps, x2, 1.0 / 5040.0);
    pc = horner<LEAN>(pc, x2, -1.0 / 720.0);
    ps = horner<LEAN>(ps, x2, -1.0 / 120.0);
    pc = horner<LEAN>(pc, x2, 1.0 / 24.0);
    ps = horner<LEAN>(ps, x2, 1.0 / 6.0);
    pc = fma(pc, x2, -0.5);
    const double sin_ra = fma(-(x * x2), ps, x);
    const double cos_ra = fma(pc, x2, 1.0);
    return FastLevelCoef{fma(throttle, M.rpm_span, M.idle_rpm), M.prop_coefficient * cos_ra, M.kyaw * sin_ra};
}

/* One Euler substep, 27 FP64 instructions (VehicleState.h:111-148 restated for throughput):
 *   rpm slew   rpm += clamp((target - rpm)/dt, +-1000) * dt, carried as the distance to the target (FastSlew below)
 *   sqrt       rsq estimate + one coupled Newton (Goldschmidt) step
 *   speed      v += dt/m * (kpc (ps^2 - v^2) - D v^3)  as  v - v^2 (kpcm + dm v) + (kpcm ps) ps   (kpcm, dm carry dt/m)
 *   heading    rotation of (sin h, cos h) by a = yaw_rate*dt, |a| <= 0.05 rad for dt <= 0.1 s: sin to a^5 (2.6e-9 a
 *              relative), cos - 1 to a^4 (2e-11 absolute), written as corrections to the old vector (no cancellation)
 *   position   x += step * sin h', y += step * cos h': step = v*dt, or v alone when the caller carries x/dt, y/dt
 *              through a level of equal substeps (SCALED) */
template <bool SCALED>
__device__ inline void fast_substep(double &m, double &v, double &sh, double &ch, double &x, double &y,
                                    double sc, double pst, double max_step, double kpcm, double dm, double kyaw_dt,
                                    double dt)
{
    m = fmax(m - max_step, 0.0);
    const double ps = fma(sc, m, pst);
    const double y0 = __builtin_amdgcn_rsq(ps);
    const double g = ps * y0;
    const double hh = 0.5 * y0;
    const double r = fma(-hh, g, 0.5);
    const double rudder_speed = fmax(fma(g, r, g), v);
    const double a = kyaw_dt * rudder_speed;
    const double v2 = v * v;
    const double w = fma(dm, v, kpcm);
    const double kps = kpcm * ps;
    v = fma(kps, ps, fma(-v2, w, v));
    const double a2 = a * a;
    const double sa = fma(a * a2, fma(a2, 1.0 / 120, -1.0 / 6), a);
    const double cm = a2 * fma(a2, 1.0 / 24, -0.5);                   /* cos a - 1 */
    const double nsh = fma(sh, cm, fma(ch, sa, sh));
    const double nch = fma(ch, cm, fma(-sh, sa, ch));
    sh = nsh; ch = nch;
    const double step = SCALED ? v : v * dt;
    x = fma(step, nsh, x);
    y = fma(step, nch, y);
}

/* The rpm slew of a level in the substeps' own terms: the rpm moves toward its target by 1000 dt per substep until it
 * is there, so what the substeps carry is m = |rpm - target| (m' = max(m - 1000 dt, 0): two instructions instead of
 * the four of a clamp) and prop_speed = C rpm = pst + sc m with sc = +-C.  pst = C target is biased by 2^-50 relative
 * and +1e-300 so that the rounded difference stays positive when the rpm starts at 0 (the reciprocal square root
 * must not see 0 or a negative rounding residue); 1e-15 relative on prop_speed is far below the model's needs. */
struct FastSlew { double m, sc, pst, sgn; };

__device__ inline FastSlew fast_slew_begin(double rpm, double target_rpm, double C_PROP)
{
    const double e = rpm - target_rpm;
    FastSlew z;
    z.m = fabs(e);
    z.sgn = copysign(1.0, e);
    z.sc = copysign(C_PROP, e);
    z.pst = fma(C_PROP * target_rpm, 1.0 + 0x1p-50, 1.0e-300);
    return z;
}

/* STRIDE 0: every substep of the level uses schedule row 0 (held in registers by the caller) */
template <int STRIDE = SUB_STRIDE, bool LEAN = false>
__device__ inline void fast_level(FastVeh &s, double rudder, double throttle, const double *sub, int n_sub, const Model &M)
{
    const FastLevelCoef c = fast_level_coef<LEAN>(rudder, throttle, M);
    FastSlew z = fast_slew_begin(s.rpm, c.target_rpm, M.c_prop);
    if (STRIDE == 0) {
        const double dt = sub[0], max_step = sub[1], dtm = sub[2], inv_dt = sub[5];
        const double kpcm = c.kpc * dtm, dm = M.drag_coefficient * dtm, kyaw_dt = c.kyaw * dt;
        double xs = s.x * inv_dt, ys = s.y * inv_dt;
#ifdef MPC_SUBSTEP_UNROLL
#pragma unroll MPC_SUBSTEP_UNROLL
#endif
        for (int k = 0; k < n_sub; k++)
            fast_substep<true>(z.m, s.v, s.sh, s.ch, xs, ys, z.sc, z.pst, max_step, kpcm, dm, kyaw_dt, dt);
        s.x = xs * dt; s.y = ys * dt;
        s.delta = s.v * dt;
    } else {
        double dt = 0;
#ifdef MPC_SUBSTEP_UNROLL
#pragma unroll MPC_SUBSTEP_UNROLL
#endif
        for (int k = 0; k < n_sub; k++) {
            const double *q = sub + k * STRIDE;
            dt = q[0];
            fast_substep<false>(z.m, s.v, s.sh, s.ch, s.x, s.y, z.sc, z.pst, q[1], c.kpc * q[2],
                                M.drag_coefficient * q[2], c.kyaw * dt, dt);
        }
        if (n_sub > 0) s.delta = s.v * dt;
    }
    s.rpm = fma(z.sgn, z.m, c.target_rpm);
}

/* Substeps 0 .. n_sub-2 equally long, the last one not (n_sub >= 2): what VehicleState::simulate itself produces
 * (:48-54: steps of 0.1 s and a remainder; the 1-s level of controller.cpp:149 is ten of 0.1 s and one of 1.4e-16 s).
 * The head runs like a uniform level (schedule row in registers, position in units of dt), the tail as one plain substep. */
/* tail_negligible (wave-uniform; the host sets it when the last substep is shorter than 2^-40 of the others, as the
 * 1.4e-16-s remainder of the reference's 1-s level is): such a substep moves the rpm by 1e-13, the speed, heading and
 * position by < 1e-15 of themselves -- ten orders below this model's tolerance -- so the state is left as it is and only
 * what the cost reads of that substep is formed: its displacement delta = v * dl, from which course made good and speed
 * over ground follow (VehicleState.h:139-144 -- the reference's cmg / sog ARE those of this last tiny step).  One
 * substep in eleven less for every child of the reference-native tree. */
template <bool LEAN = false>
__device__ inline void fast_level_head_uniform(FastVeh &s, double rudder, double throttle, const double *sub, int n_sub,
                                               const Model &M, bool tail_negligible)
{
    const FastLevelCoef c = fast_level_coef<LEAN>(rudder, throttle, M);
    FastSlew z = fast_slew_begin(s.rpm, c.target_rpm, M.c_prop);
    const double dt = sub[0], max_step = sub[1], dtm = sub[2], inv_dt = sub[5];
    const double kpcm = c.kpc * dtm, dm = M.drag_coefficient * dtm, kyaw_dt = c.kyaw * dt;
    double xs = s.x * inv_dt, ys = s.y * inv_dt;
#ifdef MPC_SUBSTEP_UNROLL
#pragma unroll MPC_SUBSTEP_UNROLL
#endif
    for (int k = 0; k < n_sub - 1; k++)
        fast_substep<true>(z.m, s.v, s.sh, s.ch, xs, ys, z.sc, z.pst, max_step, kpcm, dm, kyaw_dt, dt);
    s.x = xs * dt; s.y = ys * dt;
    const double *q = sub + (n_sub - 1) * SUB_STRIDE;
    const double dl = q[0];
    if (!tail_negligible)
        fast_substep<false>(z.m, s.v, s.sh, s.ch, s.x, s.y, z.sc, z.pst, q[1], c.kpc * q[2], M.drag_coefficient * q[2],
                            c.kyaw * dl, dl);
    s.delta = s.v * dl;
    s.rpm = fma(z.sgn, z.m, c.target_rpm);
}

/* |atan2(y, x)| in [0, pi] for the heading term, in FP32 like the atan2f it replaces (the term's accuracy is FP32's
 * either way: ~1e-7 relative, tools/fp32_error.py) but without what the library function carries for arguments this
 * caller never passes -- infinities, NaN tests, a frexp/ldexp-scaled division: min/max, one v_rcp_f32, a degree-7
 * minimax polynomial in t^2 for atan(t) / t on [0, 1] (1.7e-8 relative in exact arithmetic, 1.5 ulp evaluated in FP32;
 * fitted by tools/fit_atan.py), two selects.  ~24 instructions instead of ~40.  (0, 0) gives 0 like atan2f. */
__device__ inline float fast_abs_atan2f(float y, float x)
{
    const float ay = fabsf(y), ax = fabsf(x);
    const float mx = fmaxf(fmaxf(ax, ay), 1.0e-37f), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float s = t * t;
    float p = 2.922662999e-03f;
    p = fmaf(p, s, -1.637326181e-02f);
    p = fmaf(p, s, 4.321546480e-02f);
    p = fmaf(p, s, -7.551984489e-02f);
    p = fmaf(p, s, 1.066557691e-01f);
    p = fmaf(p, s, -1.421084106e-01f);
    p = fmaf(p, s, 1.999372840e-01f);
    p = fmaf(p, s, -3.333314955e-01f);
    float a = fmaf(t, s * p, t);
    a = ay > ax ? 1.57079637f - a : a;
    a = x < 0.0f ? 3.14159274f - a : a;
    return a;
}

/* tracking cost from the fast state; q_last = the schedule row of the level's last substep */
template <bool LEAN = false>
__device__ inline double fast_score(const double *cp, const FastVeh &s, const Weights &w, const double *q_last)
{
    /* the last substep's displacement over ground, VehicleState.h:139-140 */
    const double gdx = fma(s.delta, s.sh, q_last[3]), gdy = fma(s.delta, s.ch, q_last[4]);
    double dx = gdx, dy = gdy;
    if (dx == 0.0 && dy == 0.0) dx = 1.0;          /* atan2(0, 0) = 0: course made good = pi/2 (east)    */
    /* |headingDifference(ref, cmg)| = |angle between the travel direction and the reference heading| */
    const double cross = dx * cp[5] - dy * cp[4];
    const double dot = fma(dy, cp[5], dx * cp[4]);
#ifdef MPC_LIBRARY_ATAN2F
    const double heading_diff = (double)fabsf(atan2f((float)cross, (float)dot));
#else
    const double heading_diff = (double)fast_abs_atan2f((float)cross, (float)dot);
#endif
    const double sog = fast_sqrt(fma(gdx, gdx, gdy * gdy)) * q_last[5];
    const double speed_diff = fabs(cp[3] - sog);
    const double ex = cp[6] - s.x, ey = cp[7] - s.y;
    const double dist = fast_sqrt(fma(ex, ex, ey * ey));
    const double d = LEAN ? fast_exp_nonneg(dist) : exp(dist);
    return fma(w.wd, d, fma(w.wh, heading_diff, w.ws * speed_diff));
}

#endif /* __HIPCC__ */

}  // namespace mpc

#if defined(MPC_MODEL_HOST_STRICT_OPTIONS)
#pragma GCC pop_options
#endif
#endif
