/*
 * compat/path_planner_common/State.h -- the API surface of path_planner_common's State that the mpc host code
 * uses (complete list: SURVEY.md section 8c).  path_planner_common is an un-vendored catkin dependency of the
 * reference (package.xml:31); this minimal implementation lets the host mirror and its tests build in this
 * repository.  In a catkin workspace the real package precedes this directory on the include path.
 *
 * Conventions (pinned by the reference's headingTowardsTest, test_controller.cpp:13-23): heading is a compass
 * heading in radians, 0 = north (+y), pi/2 = east (+x), clockwise; yaw is the mathematical angle.
 * headingDifference is defined as SURVEY.md section 8c records it (parity with the real package unpinned).
 */
#ifndef MPC_AMD_COMPAT_STATE_H
#define MPC_AMD_COMPAT_STATE_H

#include <cmath>
#include <iostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

class State
{
public:
    State() : m_V{0, 0, 0, 0, -1} {}
    State(double x, double y, double heading, double speed, double time) : m_V{x, y, heading, speed, time} {}

    double& x() { return m_V[0]; }
    double& y() { return m_V[1]; }
    double& heading() { return m_V[2]; }
    double& speed() { return m_V[3]; }
    double& time() { return m_V[4]; }
    const double& x() const { return m_V[0]; }
    const double& y() const { return m_V[1]; }
    const double& heading() const { return m_V[2]; }
    const double& speed() const { return m_V[3]; }
    const double& time() const { return m_V[4]; }

    double yaw() const
    {
        const double two_pi = 2 * M_PI;
        double a = std::fmod(M_PI_2 - heading(), two_pi);
        return a < 0 ? a + two_pi : a;
    }

    /* signed minimal angle from this heading to `other`, in [-pi, pi) */
    double headingDifference(double other) const
    {
        return std::fmod(std::fmod(other - heading(), 2 * M_PI) + 3 * M_PI, 2 * M_PI) - M_PI;
    }
    double headingDifference(const State& other) const { return headingDifference(other.heading()); }

    /* compass bearing from this state to `other` */
    double headingTo(const State& other) const
    {
        double h = std::atan2(other.x() - x(), other.y() - y());
        return h < 0 ? h + 2 * M_PI : h;
    }

    std::string toStringRad() const
    {
        std::ostringstream s;
        s.precision(17);
        s << x() << " " << y() << " " << heading() << " " << speed() << " " << time();
        return s.str();
    }
    std::string toString() const
    {
        std::ostringstream s;
        s.precision(17);
        s << x() << " " << y() << " " << heading() * 180 / M_PI << " " << speed() << " " << time();
        return s.str();
    }

private:
    double m_V[5];
};

#endif
