/*
 * controller.h -- the Controller class with the public interface of the reference's (src/controller.h:41-132):
 * constructor, updateReferenceTrajectory, updatePosition, terminate, mpc, getTime, updateConfig,
 * updateDisturbanceEstimate, setTrajectoryNumber, compareStates, and the nested MpcState record.  mpc_node.cpp
 * and test/NodeStub compile against it unchanged.
 *
 * In the reference's own tree a maintainer keeps THEIR controller.h and only swaps the body of Controller::mpc
 * for controller_mpc_hip.cpp.  This header exists so that the host side builds and is tested
 * in this repository, where the reference's sources are not available to compile against.
 */
#ifndef MPC_AMD_HOST_CONTROLLER_H
#define MPC_AMD_HOST_CONTROLLER_H

#include <future>
#include <iostream>
#include <mutex>
#include <utility>

#include "control_receiver.h"
#include "path_planner_common/State.h"
#include "path_planner_common/DubinsPlan.h"
#include "VehicleState.h"
#include "CurrentEstimator.h"

class Controller
{
public:
    /* state + the control that led into it + cumulative score (controller.h:28-34) */
    struct MpcState {
        VehicleState state;
        double LastRudder{}, LastThrottle{};
        double Score{};
        MpcState() : state(State()) {}
    };

    explicit Controller(ControlReceiver* controlReceiver);
    ~Controller();

    State updateReferenceTrajectory(const DubinsPlan& referenceTrajectory, long trajectoryNumber,
                                    bool getFutureStateEstimate);
    void updatePosition(State state);
    void terminate();

    /* The hot path (controller.cpp:26-205), here on the GPU: controller_mpc_hip.cpp. */
    MpcState mpc(State startState, const DubinsPlan& referenceTrajectory, double endTime, long trajectoryNumber);

    static double getTime();

    void updateConfig(double rudderGranularity, double throttleGranularity, double distanceWeight,
                      double headingWeight, double speedWeight, double achievableThreshold,
                      bool currentEstimation, double timeout);
    void updateDisturbanceEstimate(double x, double y);
    void setTrajectoryNumber(long trajectoryNumber);
    double compareStates(const State& s1, const VehicleState& s2) const;

private:
    ControlReceiver* m_ControlReceiver;
    std::future<void> m_LastMpc;
    bool m_Achievable = true;

    double m_MinRudder = -1, m_MidRudder = 0, m_MaxRudder = 1;       /* controller.h:146 */
    double m_MinThrottle = 0, m_MaxThrottle = 1;                     /* controller.h:147 */
    double m_PlanningTime = 0.1;                                     /* controller.h:153 */

    double m_RudderGranularity = 0.0625, m_ThrottleGranularity = 0.125;
    double m_DistanceWeight{}, m_HeadingWeight{}, m_SpeedWeight{};   /* zero until updateConfig (controller.h:157) */
    double m_AchievableScoreThreshold = 2;
    double m_ReferenceTrajectoryExpirationTime = 5;
    mutable std::mutex m_ConfigMutex;

    std::mutex m_CurrentLocationMutex;
    State m_CurrentLocation;

    CurrentEstimator m_DisturbanceEstimator;
    std::pair<double, double> m_DisturbanceEstimate;

    long m_TrajectoryNumber = 0;
    std::mutex m_TrajectoryNumberMutex;

    double m_LastRudder = 0, m_LastThrottle = 0;
    std::ostream* m_Output;

    bool validTrajectoryNumber(long trajectoryNumber);
    VehicleState getStateAfterCurrentControl();
    void sendControls(double r, double t);
    void runMpc(DubinsPlan trajectory, long trajectoryNumber);

    static constexpr double c_ScoringTimeStep = 1;                   /* controller.h:227 */
    static constexpr double c_Tolerance = 1.0e-5;                    /* controller.h:231 */
};

#endif
