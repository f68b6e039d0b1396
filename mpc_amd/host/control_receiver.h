/*
 * control_receiver.h -- boundary A of SURVEY.md section 1: the callback sink Controller publishes through
 * (reference: src/control_receiver.h:9-34).  Four pure virtuals; MPCNode and the tests' NodeStub implement it.
 */
#ifndef MPC_AMD_HOST_CONTROL_RECEIVER_H
#define MPC_AMD_HOST_CONTROL_RECEIVER_H

#include <vector>

#include "path_planner_common/State.h"

class ControlReceiver
{
public:
    virtual ~ControlReceiver() = default;
    virtual void receiveControl(double rudder, double throttle) = 0;          /* rudder [-1, 1], throttle [0, 1] */
    virtual void displayTrajectory(const std::vector<State>& trajectory, bool plannerTrajectory, bool achievable) = 0;
    virtual void timedOut() = 0;
    virtual double getTime() const = 0;                                       /* seconds */
};

#endif
