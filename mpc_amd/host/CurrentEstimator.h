/*
 * CurrentEstimator.h -- estimate of the water current from the history of (state, rudder, throttle) samples
 * (reference: src/CurrentEstimator.h, src/CurrentEstimator.cpp).  Same public interface:
 * updateEstimate, getCurrent, resetEstimate, enable, disable.  The replay runs the shared strict model
 * (VehicleState::simulate -> csrc/mpc_model.h) with zero current, sequentially -- it is a scan, not a search.
 */
#ifndef MPC_AMD_HOST_CURRENTESTIMATOR_H
#define MPC_AMD_HOST_CURRENTESTIMATOR_H

#include <deque>
#include <mutex>
#include <utility>

#include "VehicleState.h"

class CurrentEstimator
{
public:
    CurrentEstimator() { resetEstimate(); }

    void updateEstimate(const State& state, double rudder, double throttle);
    std::pair<double, double> getCurrent(const State& actual) const;

    void resetEstimate()
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Samples.clear();
        m_SkipsLeft = kWarmupSkips;
    }
    void enable() { m_Enabled = true; }
    void disable() { m_Enabled = false; }

private:
    struct Sample { VehicleState state; double rudder, throttle; };
    static constexpr int kCapacity = 600;       /* CurrentEstimator.h:82: 60 s of history at 10 Hz */
    static constexpr int kWarmupSkips = 20;     /* CurrentEstimator.h:75 */
    static constexpr size_t kMinSamples = 10;   /* CurrentEstimator.cpp:18 */
    std::deque<Sample> m_Samples;
    mutable std::mutex m_Mutex;
    bool m_Enabled = true;
    int m_SkipsLeft = kWarmupSkips;
};

#endif
