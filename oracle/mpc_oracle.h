/*
 * mpc_oracle.h -- CPU ORACLE for the afb2001/mpc hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C, FP64, single-threaded restatement of the reference's algorithm for the
 * Controller::mpc path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker; nothing under mpc_amd/ or include/ links, imports or calls it.
 *
 * Reference lines restated (paths relative to /root/reference):
 *   src/VehicleState.h:20-22, 43-55, 86-101, 111-148   vehicle model
 *   src/controller.cpp:26-205                          limited-branching search + argmin
 *   src/controller.cpp:231-241                         tracking cost
 *   src/CurrentEstimator.cpp:5-41                      current estimate replay
 * Third-party arithmetic that is NOT in /root/reference (catkin workspace dependencies, no version
 * pinned: package.xml:31, CMakeLists.txt:16,19):
 *   path_planner_common  State::headingDifference, DubinsPlan::sample  -> defined here as SURVEY.md
 *                        section 8c documents them ("parity unpinned" for these two)
 *   dubins_curves        dubins_path_sample (A. Walker's Dubins-Curves, published algorithm)
 *
 * PINNING STATUS: PARITY UNPINNED (see DESIGN.md "Oracle").  The reference holds no golden numbers for
 * this path, and it cannot be compiled in this image without writing stand-ins for the absent
 * path_planner_common/dubins headers, so there is no oracle/_ref.  What anchors the restatement is weaker
 * than a pin and is checked all the same (tests/test_oracle_golden.py): (1) the outputs SURVEY.md section 4
 * and section 8c record from the survey session's own build of the reference sources (against a stub header,
 * without a committed recipe: tests/golden/survey_probe.json is typed in from there) and (2) the properties
 * the reference's own live tests assert (test/test_controller.cpp:396-433, :467, :384-394).
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* path_planner_common State is five doubles. heading: 0 = north (+y), clockwise. */
typedef struct { double x, y, heading, speed, time; } orc_state;

/* VehicleState.h:68-80 */
typedef struct { orc_state s; double rpm, cmg, sog; } orc_vehicle;

typedef struct {
    double rudder_granularity, throttle_granularity;     /* controller.h:156 */
    double w_distance, w_heading, w_speed;               /* controller.h:157 */
    double min_rudder, mid_rudder, max_rudder;            /* controller.h:146 */
    double min_throttle, max_throttle;                    /* controller.h:147 */
    double tolerance;                                     /* controller.h:231 */
    double last_rudder, last_throttle;                    /* controller.h:186 */
    double current_x, current_y;                          /* controller.h:177 */
} orc_config;

/* one MpcState of controller.h:28-34 */
typedef struct { orc_vehicle v; double last_rudder, last_throttle, score; } orc_mpc_state;

/* per deepening iteration (= per search depth) result of the tree search */
typedef struct {
    uint64_t n_leaves;       /* leaves visited at this depth */
    uint64_t best_leaf;      /* DFS ordinal (0-based) of the first leaf with the smallest last-step score */
    double   best_score;     /* that last-step score (controller.cpp:160 minScore) */
    double   best_cumulative;/* its cumulative Score (controller.cpp:156) */
} orc_level_result;

/* the boat's 11 parameters, VehicleState.h:86-96, in the reference's order */
typedef struct {
    double idle_rpm, max_rpm, max_rpm_change_rate, prop_ratio, prop_pitch, max_rudder_angle, rudder_coefficient,
           rudder_distance, mass, max_power, max_speed;
} orc_model;
void   orc_default_model(orc_model *m);
void   orc_set_model(const orc_model *m);            /* process-wide; NULL restores the reference's hull */

void   orc_default_config(orc_config *c);
orc_vehicle orc_vehicle_from_state(orc_state s);                                  /* VehicleState.h:20-22 */
int    orc_schedule(double t, double *out, int cap);                              /* VehicleState.h:49-53 */
void   orc_estimate(orc_vehicle *v, double rudder, double throttle, double dt,
                    double cx, double cy);                                        /* VehicleState.h:111-148 */
orc_vehicle orc_simulate(const orc_vehicle *v, double rudder, double throttle, double t,
                         double cx, double cy);                                   /* VehicleState.h:43-55 */
double orc_heading_difference(double heading, double other);
double orc_compare_states(const orc_state *ref, const orc_vehicle *sim,
                          double wd, double wh, double ws);                       /* controller.cpp:231-241 */

/* Dubins reference trajectory (path_planner_common DubinsWrapper / DubinsPlan, dubins_curves) */
enum { ORC_LSL = 0, ORC_LSR, ORC_RSL, ORC_RSR, ORC_RLR, ORC_LRL };
typedef struct {
    double qi[3], param[3], rho; int type;   /* dubins.h DubinsPath */
    double speed, start_time, end_time;      /* DubinsWrapper::fill(path, speed, startTime) */
} orc_dubins;
void   orc_dubins_fill(orc_dubins *w, const double qi[3], const double param[3], double rho, int type,
                       double speed, double start_time);
void   orc_dubins_sample(const orc_dubins *w, double arclen, double q[3]);
int    orc_plan_sample(const orc_dubins *plan, int n, orc_state *s);              /* DubinsPlan::sample */
double orc_plan_total_time(const orc_dubins *plan, int n);
int    orc_plan_contains_time(const orc_dubins *plan, int n, double t);

/* controller.cpp:61 */
int    orc_max_iterations(const orc_dubins *plan, int n, double start_time);

/*
 * Controller::mpc restated (controller.cpp:26-205).  checkpoints[i] is the reference trajectory sampled
 * at start.time + (i+1) (controller.cpp:96-98).  node_budget < 0 = frozen clock (never times out);
 * otherwise the "deadline" fires when that many simulate() calls have been made (controller.cpp:101).
 * trajectory must hold max_iterations+1 entries, levels max_iterations entries.
 * Returns the number of completed deepening iterations; trajectory[0..iters] is overallBestTrajectory.
 */
int    orc_mpc_tree(const orc_config *cfg, orc_state start, const orc_state *checkpoints,
                    int max_iterations, int64_t node_budget,
                    orc_mpc_state *trajectory, orc_level_result *levels, uint64_t *nodes_simulated);

/*
 * Explicit candidate sets (SURVEY.md section 8c "Explicit seeded candidate sets"): per candidate chain
 * estimate() over the substep schedule for each level and compareStates at the level's checkpoint,
 * exactly as controller.cpp:149-156 does; first strict minimum of the LAST-STEP score wins
 * (controller.cpp:158-166) or, with cumulative != 0, of the cumulative score.
 * rudders/throttles are [depth][n] (level-major); cost arrays may be NULL.
 */
typedef struct {
    uint64_t best_index; double best_last, best_cumulative;
} orc_best;
void   orc_eval_candidates(const orc_config *cfg, orc_vehicle start, const orc_state *checkpoints,
                           int depth, const double *substeps, int n_substeps,
                           const double *rudders, const double *throttles, uint64_t n,
                           int cumulative, double *cost_last, double *cost_cumulative, orc_best *best);
/* same, controls from include/mpc_candidates.h for ordinals first, first+stride, ... (count of them) */
void   orc_eval_seeded(const orc_config *cfg, orc_vehicle start, const orc_state *checkpoints,
                       int depth, const double *substeps, int n_substeps,
                       uint64_t seed, uint64_t first, uint64_t stride, uint64_t count,
                       int cumulative, double *cost_last, double *cost_cumulative, orc_best *best);
/* roll ONE control sequence and record the state after every level (trajectory[0] = start) */
void   orc_rollout(const orc_config *cfg, orc_vehicle start, const orc_state *checkpoints,
                   int depth, const double *substeps, int n_substeps,
                   const double *rudders, const double *throttles, orc_mpc_state *trajectory);

void   orc_seeded_control(uint64_t seed, uint64_t ordinal, uint32_t level, double last_rudder,
                          double last_throttle, double *rudder, double *throttle);
uint64_t orc_seeded_hash(uint64_t seed, uint64_t ordinal, uint32_t level);

/* CurrentEstimator.cpp restated.  history entries are (state, rudder, throttle). */
typedef struct { orc_state s; double rudder, throttle; } orc_history_entry;
void   orc_current_estimate(const orc_history_entry *hist, int n, int enabled, orc_state actual,
                            double *cx, double *cy);                              /* CurrentEstimator.cpp:15-41 */

#ifdef __cplusplus
}
#endif
#endif
