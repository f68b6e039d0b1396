#include "CurrentEstimator.h"

/* Reference behaviour (src/CurrentEstimator.cpp:5-13): the first 20 updates after a reset are dropped and the
 * history holds at most 600 samples, oldest first. */
void CurrentEstimator::updateEstimate(const State& state, double rudder, double throttle)
{
    std::lock_guard<std::mutex> guard(m_Mutex);
    if (m_SkipsLeft > 0) {
        m_SkipsLeft -= 1;
        return;
    }
    if ((int)m_Samples.size() == kCapacity) m_Samples.pop_front();
    m_Samples.push_back(Sample{VehicleState(state), rudder, throttle});
}

/*
 * Reference behaviour (src/CurrentEstimator.cpp:15-41): dead-reckon from the oldest sample through the recorded
 * controls with NO current, up to the query time, and attribute the position error to a constant current acting
 * over the whole span.  Two quirks are kept on purpose:
 *   - the walk starts AT the oldest sample (the reference's `begin()++` advances a temporary, :22), so its first
 *     simulate call covers zero seconds;
 *   - every sample's VehicleState was built from a State, so the replay starts from rpm(speed), not the true rpm.
 */
std::pair<double, double> CurrentEstimator::getCurrent(const State& actual) const
{
    std::lock_guard<std::mutex> guard(m_Mutex);
    if (!m_Enabled) return {0.0, 0.0};
    if (m_Samples.size() < kMinSamples) return {0.0, 0.0};

    VehicleState reckoned = m_Samples.front().state;
    const double t_query = actual.time();
    const double span = t_query - reckoned.state.time();
    if (span <= 0) return {0.0, 0.0};

    for (size_t i = 0; i < m_Samples.size(); ++i) {
        const Sample& s = m_Samples[i];
        const double t_sample = s.state.state.time();
        if (t_sample >= t_query) {
            reckoned = reckoned.simulate(s.rudder, s.throttle, t_query - reckoned.state.time());
            break;
        }
        reckoned = reckoned.simulate(s.rudder, s.throttle, t_sample - reckoned.state.time());
    }
    if (reckoned.state.time() < t_query) {                      /* query is newer than all history: hold the last control */
        const Sample& newest = m_Samples.back();
        reckoned = reckoned.simulate(newest.rudder, newest.throttle, t_query - reckoned.state.time());
    }
    const double ex = (actual.x() - reckoned.state.x()) / span;
    const double ey = (actual.y() - reckoned.state.y()) / span;
    return {ex, ey};
}
