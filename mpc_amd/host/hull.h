/*
 * hull.h -- one call that changes the boat on BOTH sides of the C-ABI, so that the device's solves (mpc_hip_set_model
 * on the drop-in's pooled handles) and the host's replays (VehicleState::simulate in CurrentEstimator::getCurrent and
 * Controller::getStateAfterCurrentControl) cannot disagree.  The reference has one hull, compiled in
 * (src/VehicleState.h:86-96, TODO at :84); the default here is that hull, bit for bit.
 */
#ifndef MPC_AMD_HOST_HULL_H
#define MPC_AMD_HOST_HULL_H

#include "VehicleState.h"
#include "controller_mpc_hip.h"

/* NULL restores the reference's hull.  Call before the controller runs (or between trajectories): solves in flight
 * keep the hull they started with, and VehicleState reads its hull without a lock. */
inline void mpc_host_set_hull(const mpc_model* hull)
{
    mpc_model m;
    if (hull) m = *hull; else mpc_hip_default_model(&m);
    VehicleState::setHull(mpc::make_model(m.idle_rpm, m.max_rpm, m.max_rpm_change_rate, m.prop_ratio, m.prop_pitch, m.max_rudder_angle,
                                          m.rudder_coefficient, m.rudder_distance, m.mass, m.max_power, m.max_speed));
    mpc_controller_set_hull(hull ? &m : nullptr);
}

#endif
