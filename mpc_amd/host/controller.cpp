/*
 * controller.cpp -- everything of Controller except the hot path: construction, configuration, the reference
 * trajectory hand-off with its achievability test, and the MPC thread loop (reference: src/controller.cpp:10-24,
 * 207-229, 243-400).  Controller::mpc lives in controller_mpc_hip.cpp.
 *
 * Behaviours kept because callers depend on them:
 *   - getStateAfterCurrentControl simulates the last control for the planning time and DISCARDS the result
 *     (controller.cpp:357: simulate() does not mutate), so MPC starts from the un-advanced position.
 *   - the in-class weights are zero until updateConfig is called (controller.h:157).
 *   - updateReferenceTrajectory waits at most one planning time for the previous MPC thread and throws
 *     "Old MPC thread refused to die" otherwise (controller.cpp:275-289).
 */
#include <chrono>
#include <ctime>
#include <stdexcept>

#include "controller.h"

Controller::Controller(ControlReceiver* controlReceiver)
    : m_ControlReceiver(controlReceiver), m_LastMpc(std::async(std::launch::async, [] {})), m_Output(&std::cerr)
{
}

Controller::~Controller() = default;

double Controller::getTime()
{
    timespec t{};
    clock_gettime(CLOCK_REALTIME, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

void Controller::terminate()
{
    std::lock_guard<std::mutex> lock(m_TrajectoryNumberMutex);
    m_TrajectoryNumber = -1;                       /* no thread's number matches: they all stop */
    m_DisturbanceEstimator.resetEstimate();
}

void Controller::updatePosition(State state)
{
    std::lock_guard<std::mutex> lock(m_CurrentLocationMutex);
    m_CurrentLocation = state;
}

void Controller::updateConfig(double rudderGranularity, double throttleGranularity, double distanceWeight,
                              double headingWeight, double speedWeight, double achievableThreshold,
                              bool currentEstimation, double timeout)
{
    std::lock_guard<std::mutex> lock(m_ConfigMutex);
    m_RudderGranularity = rudderGranularity; m_ThrottleGranularity = throttleGranularity;
    m_DistanceWeight = distanceWeight; m_HeadingWeight = headingWeight; m_SpeedWeight = speedWeight;
    m_AchievableScoreThreshold = achievableThreshold;
    if (currentEstimation) m_DisturbanceEstimator.enable(); else m_DisturbanceEstimator.disable();
    m_ReferenceTrajectoryExpirationTime = timeout;
}

void Controller::updateDisturbanceEstimate(double x, double y) { m_DisturbanceEstimate = std::make_pair(x, y); }

/* controller.cpp:231-241, host copy: used off the hot path by the achievability test below and by callers'
 * tests.  The statements are mpc::strict_score's (csrc/mpc_model.h), the same source the kernels compile. */
double Controller::compareStates(const State& s1, const VehicleState& s2) const
{
    std::lock_guard<std::mutex> lock(m_ConfigMutex);
    const double cp[4] = {s1.x(), s1.y(), s1.heading(), s1.speed()};
    const mpc::Veh v{s2.state.x(), s2.state.y(), s2.state.heading(), s2.state.speed(), s2.rpm, s2.courseMadeGood, s2.speedOverGround};
    return mpc::strict_score(cp, v, mpc::Weights{m_DistanceWeight, m_HeadingWeight, m_SpeedWeight});
}

bool Controller::validTrajectoryNumber(long trajectoryNumber)
{
    std::lock_guard<std::mutex> lock(m_TrajectoryNumberMutex);
    return trajectoryNumber == m_TrajectoryNumber;
}

void Controller::setTrajectoryNumber(long trajectoryNumber)
{
    std::lock_guard<std::mutex> lock(m_TrajectoryNumberMutex);
    m_TrajectoryNumber = trajectoryNumber;
}

VehicleState Controller::getStateAfterCurrentControl()
{
    State where;
    {
        std::lock_guard<std::mutex> lock(m_CurrentLocationMutex);
        where = m_CurrentLocation;
    }
    VehicleState start(where);
    (void)start.simulate(m_LastRudder, m_LastThrottle, m_PlanningTime, m_DisturbanceEstimate);   /* result unused, as in the reference */
    return start;
}

void Controller::sendControls(double r, double t)
{
    m_LastRudder = r;
    m_LastThrottle = t;
    m_ControlReceiver->receiveControl(r, t);
}

State Controller::updateReferenceTrajectory(const DubinsPlan& referenceTrajectory, long trajectoryNumber,
                                            bool getFutureStateEstimate)
{
    if (referenceTrajectory.empty()) {
        *m_Output << "Reference trajectory empty and cannot be used for MPC. Reference trajectory will not be updated." << std::endl;
        return {};
    }
    setTrajectoryNumber(trajectoryNumber);                    /* tells a running MPC thread to stop */
    const VehicleState start = getStateAfterCurrentControl();

    MpcState result;
    if (getFutureStateEstimate) {                             /* first solve on the caller's thread */
        result = mpc(start.state, referenceTrajectory, m_ControlReceiver->getTime() + m_PlanningTime, trajectoryNumber);
        sendControls(result.LastRudder, result.LastThrottle);
    }

    if (m_LastMpc.wait_for(std::chrono::milliseconds((int)(1000 * m_PlanningTime))) != std::future_status::ready)
        throw std::runtime_error("Old MPC thread refused to die");
    m_LastMpc.get();
    m_LastMpc = std::async(std::launch::async, [=] { runMpc(referenceTrajectory, trajectoryNumber); });

    if (!getFutureStateEstimate) return {};

    /* Achievability (controller.cpp:312-335): score the predicted state against the plan at the same time.  Close
     * enough -> hand the planner the state ON the plan; otherwise the predicted state with heading/speed over ground. */
    State onPlan;
    onPlan.time() = result.state.state.time();
    referenceTrajectory.sample(onPlan);
    const double score = compareStates(onPlan, result.state);
    std::lock_guard<std::mutex> lock(m_ConfigMutex);
    if (score <= m_AchievableScoreThreshold) {
        m_Achievable = true;
        return onPlan;
    }
    m_Achievable = m_AchievableScoreThreshold == 0;
    State predicted = result.state.state;
    predicted.heading() = result.state.courseMadeGood;
    predicted.speed() = result.state.speedOverGround;
    return predicted;
}

void Controller::runMpc(DubinsPlan trajectory, long trajectoryNumber)
{
    const double expires = m_ControlReceiver->getTime() + m_ReferenceTrajectoryExpirationTime;
    while (m_ControlReceiver->getTime() < expires) {
        if (!validTrajectoryNumber(trajectoryNumber)) break;
        const VehicleState from = getStateAfterCurrentControl();
        if (!trajectory.containsTime(from.state.time() + 3 * c_ScoringTimeStep)) break;     /* < 3 levels of plan left */
        const MpcState result = mpc(from.state, trajectory, m_ControlReceiver->getTime() + m_PlanningTime, trajectoryNumber);
        sendControls(result.LastRudder, result.LastThrottle);
    }
    if (m_ControlReceiver->getTime() >= expires) {
        *m_Output << "Controller's reference trajectory appears to have timed out. No more controls will be issued" << std::endl;
        m_ControlReceiver->timedOut();
    }
}
