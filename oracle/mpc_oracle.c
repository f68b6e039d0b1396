/*
 * mpc_oracle.c -- CPU ORACLE (test infrastructure, never part of the product path).
 * See mpc_oracle.h for scope, reference citations and pinning status.
 *
 * Build: plain IEEE double arithmetic, no FMA contraction, glibc libm (oracle/Makefile).
 * Every function names the reference lines it restates; arithmetic is written in the reference's
 * operation order so that results are bit-comparable with the reference built by g++ on x86-64.
 */
#include "mpc_oracle.h"
#include "../include/mpc_candidates.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- model constants, VehicleState.h:86-101 --------------------------------------------------- */
/* The reference compiles these in ("TODO! -- should allow setting of boat characteristics", :84); the oracle keeps
 * them in ONE process-wide record so that the product's mpc_hip_set_model can be checked against the same
 * arithmetic with another hull (orc_set_model; not thread-safe against concurrent evaluation: set it between
 * runs).  The default record holds the reference's literals; the derived values follow :98-101 expression by
 * expression, in double, so the default hull gives the bits the compile-time constants gave. */
static orc_model g_model = {0.0, 3200.0, 1000.0, 0.389105058, 20.0, M_PI / 6, 0.25, 2.0, 2000.0, 8948.4, 2.75};
#define IDLE_RPM (g_model.idle_rpm)
#define MAX_RPM (g_model.max_rpm)
#define MAX_RPM_CHANGE_RATE (g_model.max_rpm_change_rate)
#define PROP_RATIO (g_model.prop_ratio)
#define PROP_PITCH (g_model.prop_pitch)
#define MAX_RUDDER_ANGLE (g_model.max_rudder_angle)
#define RUDDER_COEFFICIENT (g_model.rudder_coefficient)
#define RUDDER_DISTANCE (g_model.rudder_distance)
#define MASS (g_model.mass)
#define MAX_POWER (g_model.max_power)
#define MAX_SPEED (g_model.max_speed)
#define k_max_prop_speed ((MAX_RPM * PROP_RATIO) / PROP_PITCH)
#define k_max_force (MAX_POWER / MAX_SPEED)
#define PROP_COEFFICIENT (k_max_force / (k_max_prop_speed * k_max_prop_speed - MAX_SPEED * MAX_SPEED))
#define DRAG_COEFFICIENT (k_max_force / (MAX_SPEED * MAX_SPEED * MAX_SPEED))

void orc_default_model(orc_model *m)
{
    const orc_model d = {0.0, 3200.0, 1000.0, 0.389105058, 20.0, M_PI / 6, 0.25, 2.0, 2000.0, 8948.4, 2.75};
    *m = d;
}

void orc_set_model(const orc_model *m)
{
    if (m) g_model = *m; else orc_default_model(&g_model);
}

void orc_default_config(orc_config *c)
{
    /* cfg/mpc.cfg:8-15 defaults; controller.h:146-147, 231, 186 */
    c->rudder_granularity = 0.0625; c->throttle_granularity = 0.125;
    c->w_distance = 1.0; c->w_heading = 20.0; c->w_speed = 5.0;
    c->min_rudder = -1.0; c->mid_rudder = 0.0; c->max_rudder = 1.0;
    c->min_throttle = 0.0; c->max_throttle = 1.0;
    c->tolerance = 1.0e-5;
    c->last_rudder = 0.0; c->last_throttle = 0.0;
    c->current_x = 0.0; c->current_y = 0.0;
}

/* VehicleState.h:20-22.  The reference leaves courseMadeGood/speedOverGround uninitialised here; they
 * are only read after an estimate() has written them, so zero is as good as anything. */
orc_vehicle orc_vehicle_from_state(orc_state s)
{
    orc_vehicle v;
    v.s = s;
    v.rpm = IDLE_RPM + (s.speed / MAX_SPEED) * (MAX_RPM - IDLE_RPM);
    v.cmg = 0.0; v.sog = 0.0;
    return v;
}

/* VehicleState.h:49-53: the substep lengths simulate() uses for a duration t (floating-point defined:
 * 11 entries for t = 1, the last one 1.39e-16). Returns the count (entries beyond cap are not stored). */
int orc_schedule(double t, double *out, int cap)
{
    int n = 0;
    while (t > 0) {
        double d = t < 0.1 ? t : 0.1;
        t -= 0.1;
        if (n < cap) out[n] = d;
        n++;
    }
    return n;
}

/* VehicleState.h:111-148, one Euler substep */
void orc_estimate(orc_vehicle *v, double rudder, double throttle, double d_time, double cx, double cy)
{
    const double prop_coefficient = PROP_COEFFICIENT;
    const double drag_coefficient = DRAG_COEFFICIENT;

    double target_rpm = IDLE_RPM + throttle * (MAX_RPM - IDLE_RPM);
    double rcr = (target_rpm - v->rpm) / d_time;
    if (rcr > MAX_RPM_CHANGE_RATE) rcr = MAX_RPM_CHANGE_RATE;
    else if (rcr < -MAX_RPM_CHANGE_RATE) rcr = -MAX_RPM_CHANGE_RATE;
    v->rpm += rcr * d_time;

    double prop_rpm = PROP_RATIO * v->rpm;
    double prop_speed = prop_rpm / PROP_PITCH;
    double rudder_speed = fmax(sqrt(prop_speed), v->s.speed);
    double thrust = prop_coefficient * (prop_speed * prop_speed - v->s.speed * v->s.speed);
    double rudder_angle = rudder * MAX_RUDDER_ANGLE;
    double thrust_fwd = thrust * cos(rudder_angle);
    double rudder_speed_yaw = rudder_speed * sin(rudder_angle);
    double yaw_rate = RUDDER_COEFFICIENT * rudder_speed_yaw / RUDDER_DISTANCE;

    v->s.heading += yaw_rate * d_time;
    v->s.heading = fmod(v->s.heading, M_PI * 2);
    if (v->s.heading < 0) v->s.heading += M_PI * 2;

    double speed = v->s.speed;
    double drag = speed * speed * speed * drag_coefficient;
    v->s.speed += ((thrust_fwd - drag) / MASS) * d_time;
    double delta = v->s.speed * d_time;

    double dx = delta * sin(v->s.heading) + cx * d_time;
    double dy = delta * cos(v->s.heading) + cy * d_time;
    v->s.x += dx;
    v->s.y += dy;
    v->cmg = M_PI_2 - atan2(dy, dx);
    v->sog = sqrt(dx * dx + dy * dy) / d_time;
    v->s.time += d_time;
}

/* VehicleState.h:43-55 */
orc_vehicle orc_simulate(const orc_vehicle *in, double rudder, double throttle, double t, double cx, double cy)
{
    if (rudder < -1) rudder = -1; else if (rudder > 1) rudder = 1;
    if (throttle < 0) throttle = 0; else if (throttle > 1) throttle = 1;
    orc_vehicle v = *in;
    while (t > 0) {
        double d = t < 0.1 ? t : 0.1;
        t -= 0.1;
        orc_estimate(&v, rudder, throttle, d, cx, cy);
    }
    return v;
}

/* path_planner_common State::headingDifference(double) -- external; definition of SURVEY.md 8c:
 * signed minimal angle from this heading to `other`, in [-pi, pi). Parity unpinned. */
double orc_heading_difference(double heading, double other)
{
    return fmod(fmod(other - heading, 2 * M_PI) + 3 * M_PI, 2 * M_PI) - M_PI;
}

/* controller.cpp:231-241 */
double orc_compare_states(const orc_state *ref, const orc_vehicle *sim, double wd, double wh, double ws)
{
    double heading_diff = fabs(orc_heading_difference(ref->heading, sim->cmg));
    double speed_diff = fabs(ref->speed - sim->sog);
    double dx = ref->x - sim->s.x;
    double dy = ref->y - sim->s.y;
    double d = exp(sqrt(dx * dx + dy * dy));
    return wd * d + wh * heading_diff + ws * speed_diff;
}

/* ---- Dubins reference trajectory -------------------------------------------------------------- */
enum { SEG_L = 0, SEG_S = 1, SEG_R = 2 };
static const int k_dirdata[6][3] = {
    {SEG_L, SEG_S, SEG_L}, {SEG_L, SEG_S, SEG_R}, {SEG_R, SEG_S, SEG_L},
    {SEG_R, SEG_S, SEG_R}, {SEG_R, SEG_L, SEG_R}, {SEG_L, SEG_R, SEG_L}};

static double mod2pi(double theta) { return theta - 2 * M_PI * floor(theta / (2 * M_PI)); }

/* dubins_curves dubins_segment: unit-radius closed forms for a left arc, straight, right arc */
static void dubins_segment(double t, const double qi[3], double qt[3], int type)
{
    double st = sin(qi[2]), ct = cos(qi[2]);
    if (type == SEG_L) {
        qt[0] = +sin(qi[2] + t) - st; qt[1] = -cos(qi[2] + t) + ct; qt[2] = t;
    } else if (type == SEG_R) {
        qt[0] = -sin(qi[2] - t) + st; qt[1] = +cos(qi[2] - t) - ct; qt[2] = -t;
    } else {
        qt[0] = ct * t; qt[1] = st * t; qt[2] = 0.0;
    }
    qt[0] += qi[0]; qt[1] += qi[1]; qt[2] += qi[2];
}

void orc_dubins_fill(orc_dubins *w, const double qi[3], const double param[3], double rho, int type,
                     double speed, double start_time)
{
    memcpy(w->qi, qi, sizeof w->qi); memcpy(w->param, param, sizeof w->param);
    w->rho = rho; w->type = type; w->speed = speed; w->start_time = start_time;
    /* test/test_controller.cpp:467 pins getEndTime() = start + (sum param) * rho / speed */
    w->end_time = start_time + (param[0] + param[1] + param[2]) * rho / speed;
}

/* dubins_curves dubins_path_sample at arclength `t` (metres) -> q = (x, y, yaw) */
void orc_dubins_sample(const orc_dubins *w, double t, double q[3])
{
    double tprime = t / w->rho;
    double qi[3] = {0.0, 0.0, w->qi[2]}, q1[3], q2[3];
    const int *types = k_dirdata[w->type];
    double p1 = w->param[0], p2 = w->param[1];
    dubins_segment(p1, qi, q1, types[0]);
    dubins_segment(p2, q1, q2, types[1]);
    if (tprime < p1) dubins_segment(tprime, qi, q, types[0]);
    else if (tprime < (p1 + p2)) dubins_segment(tprime - p1, q1, q, types[1]);
    else dubins_segment(tprime - p1 - p2, q2, q, types[2]);
    q[0] = q[0] * w->rho + w->qi[0];
    q[1] = q[1] * w->rho + w->qi[1];
    q[2] = mod2pi(q[2]);
}

/* DubinsPlan::sample(State&): fill x, y, heading, speed of `s` from the plan at s->time (SURVEY.md 8c:
 * heading = wrap(pi/2 - yaw), speed = wrapper speed). Times outside the plan clamp to its ends and
 * return 1 (the reference's behaviour there is unknown: external code). */
int orc_plan_sample(const orc_dubins *plan, int n, orc_state *s)
{
    if (n <= 0) return -1;
    int k = -1, clamped = 0;
    for (int i = 0; i < n; i++)
        if (plan[i].start_time <= s->time && s->time <= plan[i].end_time) { k = i; break; }
    double t = s->time;
    if (k < 0) {
        clamped = 1;
        if (t < plan[0].start_time) { k = 0; t = plan[0].start_time; }
        else {
            k = n - 1; t = plan[n - 1].end_time;
            for (int i = 0; i + 1 < n; i++)           /* a gap between wrappers: hold the earlier end */
                if (s->time > plan[i].end_time && s->time < plan[i + 1].start_time) { k = i; t = plan[i].end_time; break; }
        }
    }
    double q[3];
    orc_dubins_sample(&plan[k], (t - plan[k].start_time) * plan[k].speed, q);
    s->x = q[0]; s->y = q[1];
    s->heading = mod2pi(M_PI_2 - q[2]);
    s->speed = plan[k].speed;
    return clamped;
}

double orc_plan_total_time(const orc_dubins *plan, int n)
{
    double t = 0;
    for (int i = 0; i < n; i++) t += plan[i].end_time - plan[i].start_time;
    return t;
}

int orc_plan_contains_time(const orc_dubins *plan, int n, double t)
{
    for (int i = 0; i < n; i++) if (plan[i].start_time <= t && t <= plan[i].end_time) return 1;
    return 0;
}

/* controller.cpp:59-61 (c_ScoringTimeStep = 1; the cast truncates) */
int orc_max_iterations(const orc_dubins *plan, int n, double start_time)
{
    double trajectory_start = plan[0].start_time;
    return (int)((trajectory_start - start_time + orc_plan_total_time(plan, n)) / 1.0 - 1);
}

/* ---- limited-branching iterative-deepening search, controller.cpp:26-205 ------------------------- */
typedef struct {
    const orc_config *cfg;
    const orc_state *checkpoints;
    int iterations;                 /* current maximum depth index (0-based) */
    int64_t budget;                 /* simulate() calls left; <0 = unlimited */
    int timed_out;
    uint64_t nodes;
    double min_score;
    uint64_t leaf_counter, best_leaf;
    orc_mpc_state *current;         /* [max_iterations + 1] */
    orc_mpc_state *best;            /* [max_iterations + 1] */
} search_ctx;

/* controller.cpp:121-146: control set of a frame, derived from the control that led into it */
static void control_set(const orc_config *c, double last_r, double last_t,
                        double rudders[6], int *n_r, double throttles[5], int *n_t)
{
    const double tol = c->tolerance;
    int nr = 0, nt = 0;
    rudders[nr++] = c->min_rudder; rudders[nr++] = c->mid_rudder; rudders[nr++] = c->max_rudder;
    throttles[nt++] = c->min_throttle; throttles[nt++] = c->max_throttle;
    const double low_r = last_r - c->rudder_granularity, high_r = last_r + c->rudder_granularity;
    if (c->min_rudder + tol < low_r && fabs(c->mid_rudder - low_r) > tol) rudders[nr++] = low_r;
    if (fabs(c->mid_rudder - high_r) > tol && c->max_rudder > high_r + tol) rudders[nr++] = high_r;
    if (fabs(c->min_rudder - last_r) > tol && fabs(c->mid_rudder - last_r) > tol &&
        fabs(c->max_rudder - last_r) > tol) rudders[nr++] = last_r;
    const double low_t = last_t - c->throttle_granularity, high_t = last_t + c->throttle_granularity;
    if (c->min_throttle + tol < low_t) throttles[nt++] = low_t;
    if (c->max_throttle > high_t + tol) throttles[nt++] = high_t;
    if (fabs(c->min_throttle - last_t) > tol && fabs(c->max_throttle - last_t) > tol) throttles[nt++] = last_t;
    *n_r = nr; *n_t = nt;
}

/* The reference walks an explicit stack (controller.cpp:100-183); the same visit order written
 * recursively: throttle outer, rudder inner (controller.cpp:174-182), depth first. */
static void search(search_ctx *c, int depth)
{
    const orc_mpc_state *cur = &c->current[depth];
    double rudders[6], throttles[5]; int n_r, n_t;
    control_set(c->cfg, cur->last_rudder, cur->last_throttle, rudders, &n_r, throttles, &n_t);
    for (int ti = 0; ti < n_t; ti++) {
        for (int ri = 0; ri < n_r; ri++) {
            if (c->budget == 0) { c->timed_out = 1; return; }       /* controller.cpp:101-103 */
            if (c->budget > 0) c->budget--;
            c->nodes++;
            orc_mpc_state *next = &c->current[depth + 1];
            next->v = orc_simulate(&cur->v, rudders[ri], throttles[ti], 1.0,
                                   c->cfg->current_x, c->cfg->current_y);       /* :149-150 */
            next->last_rudder = rudders[ri]; next->last_throttle = throttles[ti];
            double score = orc_compare_states(&c->checkpoints[depth], &next->v, c->cfg->w_distance,
                                              c->cfg->w_heading, c->cfg->w_speed);   /* :155 */
            next->score = cur->score + score;                                       /* :156 */
            if (depth == c->iterations) {                                           /* :158-166 */
                if (score < c->min_score) {
                    memcpy(c->best, c->current, sizeof(orc_mpc_state) * (size_t)(c->iterations + 2));
                    c->min_score = score;
                    c->best_leaf = c->leaf_counter;
                }
                c->leaf_counter++;
            } else {
                search(c, depth + 1);
                if (c->timed_out) return;
            }
        }
    }
}

int orc_mpc_tree(const orc_config *cfg, orc_state start, const orc_state *checkpoints, int max_iterations,
                 int64_t node_budget, orc_mpc_state *trajectory, orc_level_result *levels,
                 uint64_t *nodes_simulated)
{
    if (max_iterations <= 0) { if (nodes_simulated) *nodes_simulated = 0; return 0; }
    search_ctx c;
    memset(&c, 0, sizeof c);
    c.cfg = cfg; c.checkpoints = checkpoints; c.budget = node_budget;
    c.current = calloc((size_t)max_iterations + 1, sizeof(orc_mpc_state));
    c.best = calloc((size_t)max_iterations + 1, sizeof(orc_mpc_state));
    c.current[0].v = orc_vehicle_from_state(start);            /* controller.cpp:70 via MpcState(State) */
    c.current[0].last_rudder = cfg->last_rudder;               /* :71-72 */
    c.current[0].last_throttle = cfg->last_throttle;
    c.current[0].score = 0;
    int iterations = 0;
    while (iterations < max_iterations) {                       /* :80 */
        c.iterations = iterations; c.min_score = DBL_MAX;       /* :81 */
        c.leaf_counter = 0; c.best_leaf = 0;
        search(&c, 0);
        if (c.timed_out) break;                                 /* goto mpcCleanup */
        if (levels) {
            levels[iterations].n_leaves = c.leaf_counter;
            levels[iterations].best_leaf = c.best_leaf;
            levels[iterations].best_score = c.min_score;
            levels[iterations].best_cumulative = c.best[iterations + 1].score;
        }
        iterations++;                                           /* :185-188 */
        memcpy(trajectory, c.best, sizeof(orc_mpc_state) * (size_t)(iterations + 1));
    }
    if (nodes_simulated) *nodes_simulated = c.nodes;
    free(c.current); free(c.best);
    return iterations;
}

/* ---- explicit candidate sets ------------------------------------------------------------------ */
static void roll_one(const orc_config *cfg, orc_vehicle v, const orc_state *checkpoints, int depth,
                     const double *substeps, int n_substeps, const double *rudders, const double *throttles,
                     uint64_t stride, double *last, double *cumulative, orc_mpc_state *trajectory)
{
    double cum = 0, score = 0;
    if (trajectory) { trajectory[0].v = v; trajectory[0].last_rudder = cfg->last_rudder;
                      trajectory[0].last_throttle = cfg->last_throttle; trajectory[0].score = 0; }
    for (int l = 0; l < depth; l++) {
        double r = rudders[(uint64_t)l * stride], t = throttles[(uint64_t)l * stride];
        if (r < -1) r = -1; else if (r > 1) r = 1;              /* VehicleState.h:44-47 */
        if (t < 0) t = 0; else if (t > 1) t = 1;
        for (int s = 0; s < n_substeps; s++)
            orc_estimate(&v, r, t, substeps[s], cfg->current_x, cfg->current_y);
        score = orc_compare_states(&checkpoints[l], &v, cfg->w_distance, cfg->w_heading, cfg->w_speed);
        cum += score;
        if (trajectory) { trajectory[l + 1].v = v; trajectory[l + 1].last_rudder = r;
                          trajectory[l + 1].last_throttle = t; trajectory[l + 1].score = cum; }
    }
    *last = score; *cumulative = cum;
}

void orc_eval_candidates(const orc_config *cfg, orc_vehicle start, const orc_state *checkpoints, int depth,
                         const double *substeps, int n_substeps, const double *rudders,
                         const double *throttles, uint64_t n, int cumulative, double *cost_last,
                         double *cost_cumulative, orc_best *best)
{
    double min_key = DBL_MAX;
    if (best) { best->best_index = 0; best->best_last = DBL_MAX; best->best_cumulative = DBL_MAX; }
    for (uint64_t i = 0; i < n; i++) {
        double last, cum;
        roll_one(cfg, start, checkpoints, depth, substeps, n_substeps, rudders + i, throttles + i, n,
                 &last, &cum, NULL);
        if (cost_last) cost_last[i] = last;
        if (cost_cumulative) cost_cumulative[i] = cum;
        double key = cumulative ? cum : last;
        if (best && key < min_key) { min_key = key; best->best_index = i; best->best_last = last; best->best_cumulative = cum; }
    }
}

void orc_eval_seeded(const orc_config *cfg, orc_vehicle start, const orc_state *checkpoints, int depth,
                     const double *substeps, int n_substeps, uint64_t seed, uint64_t first, uint64_t stride,
                     uint64_t count, int cumulative, double *cost_last, double *cost_cumulative, orc_best *best)
{
    double *r = malloc(sizeof(double) * (size_t)depth), *t = malloc(sizeof(double) * (size_t)depth);
    double min_key = DBL_MAX;
    if (best) { best->best_index = 0; best->best_last = DBL_MAX; best->best_cumulative = DBL_MAX; }
    for (uint64_t k = 0; k < count; k++) {
        uint64_t ordinal = first + k * stride;
        for (int l = 0; l < depth; l++)
            mpc_candidate_control(seed, ordinal, (uint32_t)l, cfg->last_rudder, cfg->last_throttle, &r[l], &t[l]);
        double last, cum;
        roll_one(cfg, start, checkpoints, depth, substeps, n_substeps, r, t, 1, &last, &cum, NULL);
        if (cost_last) cost_last[k] = last;
        if (cost_cumulative) cost_cumulative[k] = cum;
        double key = cumulative ? cum : last;
        if (best && key < min_key) { min_key = key; best->best_index = ordinal; best->best_last = last; best->best_cumulative = cum; }
    }
    free(r); free(t);
}

void orc_rollout(const orc_config *cfg, orc_vehicle start, const orc_state *checkpoints, int depth,
                 const double *substeps, int n_substeps, const double *rudders, const double *throttles,
                 orc_mpc_state *trajectory)
{
    double last, cum;
    roll_one(cfg, start, checkpoints, depth, substeps, n_substeps, rudders, throttles, 1, &last, &cum, trajectory);
}

/* the seeded workload definition of include/mpc_candidates.h, exported for the tests */
void orc_seeded_control(uint64_t seed, uint64_t ordinal, uint32_t level, double last_rudder, double last_throttle,
                        double *rudder, double *throttle)
{
    mpc_candidate_control(seed, ordinal, level, last_rudder, last_throttle, rudder, throttle);
}

uint64_t orc_seeded_hash(uint64_t seed, uint64_t ordinal, uint32_t level)
{
    return mpc_candidate_hash(seed, ordinal, level);
}

/* ---- CurrentEstimator::getCurrent, CurrentEstimator.cpp:15-41 ------------------------------------- */
void orc_current_estimate(const orc_history_entry *hist, int n, int enabled, orc_state actual,
                          double *cx, double *cy)
{
    *cx = 0.0; *cy = 0.0;
    if (!enabled) return;
    if (n < 10) return;                                          /* :18 */
    orc_vehicle expected = orc_vehicle_from_state(hist[0].s);    /* :19 (VehicleState built at insertion) */
    double time_difference = actual.time - expected.s.time;
    if (time_difference <= 0) return;                            /* :21 */
    /* :22 -- `m_History.begin()++` post-increments a temporary, so the loop starts AT the first entry */
    for (int i = 0; i < n; i++) {
        if (hist[i].s.time >= actual.time) {
            expected = orc_simulate(&expected, hist[i].rudder, hist[i].throttle,
                                    actual.time - expected.s.time, 0.0, 0.0);
            break;
        }
        expected = orc_simulate(&expected, hist[i].rudder, hist[i].throttle,
                                hist[i].s.time - expected.s.time, 0.0, 0.0);
    }
    if (expected.s.time < actual.time)                           /* :34-37 */
        expected = orc_simulate(&expected, hist[n - 1].rudder, hist[n - 1].throttle,
                                actual.time - expected.s.time, 0.0, 0.0);
    *cx = (actual.x - expected.s.x) / time_difference;           /* :39-40 */
    *cy = (actual.y - expected.s.y) / time_difference;
}
