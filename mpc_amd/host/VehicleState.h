/*
 * VehicleState.h -- host-side vessel state with the public interface of the reference's VehicleState
 * (src/VehicleState.h:20-80: constructor from State, simulate(r, t, T[, current]), toString(), members rpm,
 * courseMadeGood, speedOverGround, state), which CurrentEstimator, Controller::getStateAfterCurrentControl and
 * the reference's tests call directly.
 *
 * The arithmetic is NOT written here: simulate() runs mpc::strict_level of csrc/mpc_model.h -- the same source the
 * strict FP64 HIP kernels compile -- so host and device share one statement of the model
 * (VehicleState.h:111-148), built here by g++ with -ffp-contract=off.
 */
#ifndef MPC_AMD_HOST_VEHICLESTATE_H
#define MPC_AMD_HOST_VEHICLESTATE_H

#include <string>
#include <utility>

#include "path_planner_common/State.h"
#include "../csrc/mpc_model.h"

class VehicleState
{
public:
    /* rpm follows from the speed, VehicleState.h:20-22; the reference leaves the two derived members unset */
    VehicleState(State s) : rpm(mpc::rpm_from_speed(s.speed(), hull())), courseMadeGood(0), speedOverGround(0), state(s) {}

    /* The boat.  The reference's 11 model parameters are compile-time constants (VehicleState.h:86-96) under the note
     * "should allow setting of boat characteristics" (:84); here they are one process-wide mpc::Model -- like constants,
     * the same for every VehicleState, so CurrentEstimator and getStateAfterCurrentControl follow without any change --
     * whose default is the reference's hull, bit for bit.  Set it before the controller runs (it is read without a
     * lock, as constants are); mpc_amd/host/hull.h sets it together with the device side's (mpc_hip_set_model). */
    static const mpc::Model& hull() { return hullStorage(); }
    static void setHull(const mpc::Model& m) { hullStorage() = m; }

    VehicleState simulate(double rudder, double throttle, double t) const
    {
        return simulate(rudder, throttle, t, std::make_pair(0.0, 0.0));
    }

    /* Non-mutating, like the reference's (VehicleState.h:48,54): returns the advanced copy. */
    VehicleState simulate(double rudder, double throttle, double t, std::pair<double, double> estimatedCurrent) const
    {
        /* substep lengths exactly as the loop of VehicleState.h:49-53 produces them */
        double rows[SUB_STRIDE * kMaxSubsteps];
        int n = 0;
        VehicleState out(*this);
        const mpc::Model hull = VehicleState::hull();          /* default: the reference's compile-time hull, VehicleState.h:86-101 */
        mpc::Veh v{state.x(), state.y(), state.heading(), state.speed(), rpm, courseMadeGood, speedOverGround};
        double elapsed = state.time();
        while (t > 0) {
            const double d = t < 0.1 ? t : 0.1;
            t -= 0.1;
            rows[n * SUB_STRIDE] = d;
            rows[n * SUB_STRIDE + 5] = 1.0 / d;          /* the schedule row's reciprocal slot (the device's div_known reads it) */
            elapsed += d;
            if (++n == kMaxSubsteps) { mpc::strict_level(v, rudder, throttle, rows, n, estimatedCurrent.first, estimatedCurrent.second, hull); n = 0; }
        }
        if (n) mpc::strict_level(v, rudder, throttle, rows, n, estimatedCurrent.first, estimatedCurrent.second, hull);
        out.state = State(v.x, v.y, v.h, v.v, elapsed);
        out.rpm = v.rpm; out.courseMadeGood = v.cmg; out.speedOverGround = v.sog;
        return out;
    }

    std::string toString() const { return state.toStringRad(); }

    double rpm;
    double courseMadeGood;
    double speedOverGround;
    State state;

private:
    static constexpr int kMaxSubsteps = 64;
    static mpc::Model& hullStorage() { static mpc::Model m = mpc::default_model(); return m; }
};

#endif
