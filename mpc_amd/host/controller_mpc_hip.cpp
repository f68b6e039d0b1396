/*
 * controller_mpc_hip.cpp -- Controller::mpc on the MI355X: the drop-in replacement for the body of
 * src/controller.cpp:26-205.  It marshals one solve into the C-ABI of include/mpc_hip.h and rebuilds what the
 * reference's body produced: the MpcState one scoring step along the best trajectory, the displayTrajectory
 * call, and the same exceptions / empty returns.
 *
 *   reference step (controller.cpp)                      here
 *   :30-32  stale trajectory number -> return {}         same check, before touching the device
 *   :37     empty plan -> throw                          same text (mpc_hip_status_string(MPC_ERR_EMPTY_PLAN))
 *   :40-45  copy granularities under m_ConfigMutex       ONE snapshot of granularities AND weights AND current AND
 *                                                        last controls (the reference re-reads the weights per node)
 *   :59-61  maxIterations                                same expression
 *   :96-98  sample the plan per deepening iteration      all maxIterations checkpoints sampled up front
 *   :80-189 iterative-deepening DFS                      mpc_hip_solve_tree_deep (level-synchronous)
 *   :83     cancellation poll per iteration              polled in the time callback, between levels
 *   :101    deadline poll per node                       polled between levels through the same callback
 *   :196-204 cleanup, displayTrajectory, finite check    same
 *
 * There is no CPU fallback: a HIP failure throws std::runtime_error with the library's message.
 * Up to two mpc() calls overlap on one Controller (the service thread starts a solve while the old MPC thread is
 * still inside its last one, SURVEY.md section 7 hard part 7): handles come from a small pool.
 */
#include <cmath>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "controller.h"      /* the reference's own header once this file sits in its src/  */
#include "mpc_hip.h"         /* from the include path: -I<this repository>/include */
#include "controller_mpc_hip.h"

namespace {

/* a pool of handles on device 0 (MPC_HIP_DEVICE overrides); one checked out per running solve.  The pool also holds the
 * hull the solves run with (mpc_controller_set_hull; default: the reference's constants, VehicleState.h:86-96): a handle
 * whose hull is older than the pool's gets mpc_hip_set_model when it is next checked out. */
class HandlePool
{
public:
    struct Entry { mpc_hip_handle* h; unsigned long hull; };

    Entry acquire()
    {
        Entry e{nullptr, 0};
        mpc_model hull;
        unsigned long want;
        {
            std::lock_guard<std::mutex> lock(m_Mutex);
            if (!m_Free.empty()) { e = m_Free.back(); m_Free.pop_back(); }
            hull = m_Hull; want = m_HullGeneration;
        }
        if (!e.h) {
            int device = 0;
            if (const char* env = std::getenv("MPC_HIP_DEVICE")) device = std::atoi(env);
            const int s = mpc_hip_create(device, nullptr, &e.h);
            if (s != MPC_OK) throw std::runtime_error(std::string("mpc_hip_create: ") + mpc_hip_status_string(s) + ": " + mpc_hip_last_error());
            /* a controller wants latency, not kernel timings: events on the dispatches cost ~30 us of a 90-us solve */
            if (!std::getenv("MPC_HIP_TIMING")) mpc_hip_set_timing(e.h, 0);
        }
        if (e.hull != want) {
            const int s = mpc_hip_set_model(e.h, &hull, 1);
            if (s != MPC_OK) {
                const std::string why = std::string("mpc_hip_set_model: ") + mpc_hip_status_string(s) + ": " + mpc_hip_last_error();
                release(e);
                throw std::runtime_error(why);
            }
            e.hull = want;
        }
        return e;
    }
    void release(Entry e)
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Free.push_back(e);
    }
    void setHull(const mpc_model& m)
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Hull = m;
        m_HullGeneration++;
    }
    HandlePool() { mpc_hip_default_model(&m_Hull); }
    /* no destructor on purpose: see pool() */

private:
    std::mutex m_Mutex;
    std::vector<Entry> m_Free;
    mpc_model m_Hull;
    unsigned long m_HullGeneration = 0;      /* 0 = the library's default hull: a new handle already has it */
};

/* The pool is never destroyed: a static object's destructor would call mpc_hip_destroy during static teardown, whose
 * order against the HIP runtime's own teardown is unspecified (hipFree on a runtime that is already gone).  The handles'
 * device memory goes back to the driver with the process, as all device memory does. */
HandlePool& pool()
{
    static HandlePool* p = new HandlePool;
    return *p;
}

struct Lease {
    HandlePool::Entry e;
    mpc_hip_handle* h;
    Lease() : e(pool().acquire()), h(e.h) {}
    ~Lease() { pool().release(e); }
};

/* what the between-levels callback needs: the clock, the deadline, and the cancellation test */
struct PollContext {
    /* a constructor rather than brace initialisation: the reference builds with -std=c++11 (CMakeLists.txt:6), where
     * a struct with a default member initialiser is not an aggregate */
    PollContext(const ControlReceiver* r, double end, std::function<bool()> valid)
        : receiver(r), endTime(end), stillValid(std::move(valid)), cancelled(false) {}
    const ControlReceiver* receiver;
    double endTime;
    std::function<bool()> stillValid;
    bool cancelled;
};

double poll(void* p)
{
    auto* c = static_cast<PollContext*>(p);
    if (!c->stillValid()) { c->cancelled = true; return c->endTime; }     /* stop like a deadline; reported below */
    return c->receiver->getTime();
}

}  // namespace

/* Another boat (the reference's TODO at VehicleState.h:84): every later Controller::mpc solve of this process runs with
 * this hull.  Declared in controller_mpc_hip.h; the host-side VehicleState of this repository follows through
 * mpc_amd/host/hull.h (mpc_host_set_hull sets both), a maintainer's own VehicleState.h through its constants. */
extern "C" void mpc_controller_set_hull(const mpc_model* hull)
{
    mpc_model m;
    if (hull) m = *hull; else mpc_hip_default_model(&m);
    pool().setHull(m);
}

Controller::MpcState Controller::mpc(State startState, const DubinsPlan& referenceTrajectory, double endTime,
                                     long trajectoryNumber)
{
    if (!validTrajectoryNumber(trajectoryNumber)) return {};                                  /* :30-32 */
    if (referenceTrajectory.empty()) throw std::runtime_error(mpc_hip_status_string(MPC_ERR_EMPTY_PLAN));   /* :37 */

    mpc_problem problem;
    mpc_hip_default_problem(&problem);
    {
        std::unique_lock<std::mutex> lock(m_ConfigMutex);                                     /* :40-45 (+ weights) */
        problem.rudder_granularity = m_RudderGranularity;
        problem.throttle_granularity = m_ThrottleGranularity;
        problem.w_distance = m_DistanceWeight; problem.w_heading = m_HeadingWeight; problem.w_speed = m_SpeedWeight;
    }
    problem.min_rudder = m_MinRudder; problem.mid_rudder = m_MidRudder; problem.max_rudder = m_MaxRudder;
    problem.min_throttle = m_MinThrottle; problem.max_throttle = m_MaxThrottle;
    problem.tolerance = c_Tolerance;
    problem.start_x = startState.x(); problem.start_y = startState.y();
    problem.start_heading = startState.heading(); problem.start_speed = startState.speed();
    problem.last_rudder = m_LastRudder; problem.last_throttle = m_LastThrottle;               /* :71-72 */
    const std::pair<double, double> current = m_DisturbanceEstimate;                          /* :150, read once */
    problem.current_x = current.first; problem.current_y = current.second;

    const double trajectoryStartTime = referenceTrajectory.get().front().getStartTime();
    int maxIterations = (trajectoryStartTime - startState.time() + referenceTrajectory.totalTime()) / c_ScoringTimeStep - 1;   /* :59-61 */
    if (maxIterations > MPC_MAX_DEPTH) maxIterations = MPC_MAX_DEPTH;     /* 64 s of lookahead: far beyond what fits any deadline */
    if (maxIterations < 0) maxIterations = 0;
    problem.depth = maxIterations > 0 ? maxIterations : 1;
    for (int i = 0; i < maxIterations; i++) {                                                 /* :96-98 */
        State checkpoint;
        checkpoint.time() = startState.time() + ((i + 1) * c_ScoringTimeStep);
        referenceTrajectory.sample(checkpoint);
        problem.checkpoints[i] = mpc_checkpoint{checkpoint.x(), checkpoint.y(), checkpoint.heading(), checkpoint.speed()};
    }
    problem.n_substeps = mpc_hip_substep_schedule(c_ScoringTimeStep, problem.substeps, MPC_MAX_SUBSTEPS);   /* VehicleState.h:49-53 */

    PollContext ctx(m_ControlReceiver, endTime, [this, trajectoryNumber] { return validTrajectoryNumber(trajectoryNumber); });
    mpc_result result;
    int status;
    {
        Lease lease;
        /* Which model EXPANDS the tree; whichever it is, the strict model decides every level and the returned states
         * are the strict path's (same leaf, same bits: tests/test_gpu_tree.py).  The streamlined model is sooner from
         * depth 6 on (depth 7: 1.1 ms against 1.4 ms, depth 8: 7 ms against 11 ms), the strict one below (depth 4: 0.11 ms against 0.19 ms), so the
         * plan's length chooses; MPC_HIP_TREE_PRECISION=fast|strict (or f64) in the environment overrides. */
        static const int forced = [] {
            const char *e = getenv("MPC_HIP_TREE_PRECISION");
            if (e && std::string(e) == "fast") return (int)MPC_PREC_FAST;
            if (e && (std::string(e) == "strict" || std::string(e) == "f64")) return (int)MPC_PREC_F64;
            return -1;
        }();
        const int precision = forced >= 0 ? forced : (maxIterations > 5 ? (int)MPC_PREC_FAST : (int)MPC_PREC_F64);
        /* (the _deep entry point is mpc_hip_solve_tree for as long as one expansion holds a level -- depth 7 with the default
         * control sets, ~1.5 ms -- and then goes on a level at a time in pieces while the clock allows: depth 8 in 7-11 ms) */
        status = mpc_hip_solve_tree_deep(lease.h, &problem, maxIterations, precision, poll, &ctx, endTime, nullptr, &result);
    }
    if (ctx.cancelled) return {};                                                             /* :83-86 */
    if (status == MPC_ERR_NO_ITERATION || status == MPC_ERR_NON_FINITE) {
        if (status == MPC_ERR_NON_FINITE) {                                                   /* the reference displays first (:200-202) */
            std::vector<State> shown;
            double t = startState.time();
            for (int l = 0; l <= result.levels; l++) {
                if (l > 0) for (int k = 0; k < problem.n_substeps; k++) t += problem.substeps[k];
                shown.emplace_back(result.trajectory[l].x, result.trajectory[l].y, result.trajectory[l].heading,
                                   result.trajectory[l].speed, t);
            }
            m_ControlReceiver->displayTrajectory(shown, false, m_Achievable);
        }
        throw std::runtime_error(mpc_hip_status_string(status));                              /* :196, :202 */
    }
    if (status != MPC_OK)
        throw std::runtime_error(std::string("mpc_hip_solve_tree_deep: ") + mpc_hip_status_string(status) + ": " + mpc_hip_last_error());

    std::vector<State> forDisplay;                                                            /* :197-200 */
    forDisplay.reserve(result.levels + 1);
    double time = startState.time();
    for (int l = 0; l <= result.levels; l++) {
        const mpc_level_state& s = result.trajectory[l];
        /* a state's time is what estimate() accumulated: += d_time per substep (VehicleState.h:146), which for
         * the 11-substep level ends an ulp or two short of start + l */
        if (l > 0) for (int k = 0; k < problem.n_substeps; k++) time += problem.substeps[k];
        forDisplay.emplace_back(s.x, s.y, s.heading, s.speed, time);
    }
    m_ControlReceiver->displayTrajectory(forDisplay, false, m_Achievable);

    const mpc_level_state& first = result.trajectory[1];                                      /* :204 */
    MpcState out;
    out.state = VehicleState(forDisplay[1]);
    out.state.rpm = first.rpm;
    out.state.courseMadeGood = first.course_made_good;
    out.state.speedOverGround = first.speed_over_ground;
    out.LastRudder = first.rudder; out.LastThrottle = first.throttle;
    out.Score = first.score;
    return out;
}
