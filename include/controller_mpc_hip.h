/*
 * controller_mpc_hip.h -- what mpc_amd/host/controller_mpc_hip.cpp (the drop-in body of Controller::mpc,
 * reference src/controller.cpp:26-205) exports besides Controller::mpc itself.  Plain C linkage, no C++ types.
 */
#ifndef CONTROLLER_MPC_HIP_H
#define CONTROLLER_MPC_HIP_H

#include "mpc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The boat the drop-in's solves run with: replaces, for every later Controller::mpc call of the process, the
 * compile-time constants of VehicleState.h:86-96 (the reference's own TODO at :84, "should allow setting of boat
 * characteristics") through mpc_hip_set_model on the drop-in's pooled handles.  NULL restores the reference's hull.
 * An invalid hull makes the next Controller::mpc throw with mpc_hip_set_model's message.  The HOST-side model -- the
 * reference's VehicleState.h, used by CurrentEstimator and getStateAfterCurrentControl -- is the maintainer's to keep in
 * step (this repository's mirror: mpc_amd/host/hull.h, mpc_host_set_hull). */
void mpc_controller_set_hull(const mpc_model *hull);

#ifdef __cplusplus
}
#endif
#endif
