/*
 * compat/path_planner_common/DubinsPlan.h -- DubinsWrapper and DubinsPlan as the mpc host code uses them
 * (src/controller.cpp:37,59,61,98,250,314,383; src/mpc_node.cpp:245-248; test/test_controller.cpp:464-467).
 * Segment sampling follows A. Walker's Dubins-Curves closed forms; sample() returns the compass heading
 * wrap(pi/2 - yaw) and the wrapper's speed.  See State.h in this directory for why this
 * file exists.
 */
#ifndef MPC_AMD_COMPAT_DUBINSPLAN_H
#define MPC_AMD_COMPAT_DUBINSPLAN_H

#include <cmath>
#include <vector>

#include "../dubins.h"
#include "State.h"

class DubinsWrapper
{
public:
    void fill(const DubinsPath& path, double speed, double startTime)
    {
        m_Path = path; m_Speed = speed; m_StartTime = startTime;
        m_EndTime = startTime + (path.param[0] + path.param[1] + path.param[2]) * path.rho / speed;
    }
    double getStartTime() const { return m_StartTime; }
    double getEndTime() const { return m_EndTime; }
    void updateEndTime(double t) { m_EndTime = t; }
    double getSpeed() const { return m_Speed; }
    bool containsTime(double t) const { return m_StartTime <= t && t <= m_EndTime; }

    /* pose (x, y, yaw) at arclength s metres from the start */
    void poseAt(double s, double q[3]) const
    {
        static const int kinds[6][3] = {{0, 1, 0}, {0, 1, 2}, {2, 1, 0}, {2, 1, 2}, {2, 0, 2}, {0, 2, 0}};  /* L=0 S=1 R=2 */
        const int* k = kinds[m_Path.type];
        const double tp = s / m_Path.rho, p1 = m_Path.param[0], p2 = m_Path.param[1];
        const double qi[3] = {0.0, 0.0, m_Path.qi[2]};
        double q1[3], q2[3];
        segment(p1, qi, q1, k[0]);
        segment(p2, q1, q2, k[1]);
        if (tp < p1) segment(tp, qi, q, k[0]);
        else if (tp < p1 + p2) segment(tp - p1, q1, q, k[1]);
        else segment(tp - p1 - p2, q2, q, k[2]);
        q[0] = q[0] * m_Path.rho + m_Path.qi[0];
        q[1] = q[1] * m_Path.rho + m_Path.qi[1];
        q[2] = mod2pi(q[2]);
    }

    void sample(State& s, double t) const
    {
        double q[3];
        poseAt((t - m_StartTime) * m_Speed, q);
        s.x() = q[0]; s.y() = q[1];
        s.heading() = mod2pi(M_PI_2 - q[2]);
        s.speed() = m_Speed;
    }

    static double mod2pi(double a) { return a - 2 * M_PI * std::floor(a / (2 * M_PI)); }

private:
    static void segment(double t, const double qi[3], double qt[3], int kind)
    {
        const double st = std::sin(qi[2]), ct = std::cos(qi[2]);
        if (kind == 0) { qt[0] = std::sin(qi[2] + t) - st; qt[1] = -std::cos(qi[2] + t) + ct; qt[2] = t; }
        else if (kind == 2) { qt[0] = -std::sin(qi[2] - t) + st; qt[1] = std::cos(qi[2] - t) - ct; qt[2] = -t; }
        else { qt[0] = ct * t; qt[1] = st * t; qt[2] = 0.0; }
        qt[0] += qi[0]; qt[1] += qi[1]; qt[2] += qi[2];
    }
    DubinsPath m_Path{};
    double m_Speed = 0, m_StartTime = 0, m_EndTime = 0;
};

class DubinsPlan
{
public:
    bool empty() const { return m_Wrappers.empty(); }
    const std::vector<DubinsWrapper>& get() const { return m_Wrappers; }
    void append(const DubinsWrapper& w) { m_Wrappers.push_back(w); }
    double getStartTime() const { return m_Wrappers.front().getStartTime(); }
    double getEndTime() const { return m_Wrappers.back().getEndTime(); }
    double totalTime() const
    {
        double t = 0;
        for (const auto& w : m_Wrappers) t += w.getEndTime() - w.getStartTime();
        return t;
    }
    bool containsTime(double t) const
    {
        for (const auto& w : m_Wrappers) if (w.containsTime(t)) return true;
        return false;
    }
    /* fill x, y, heading, speed of `s` from the plan at s.time(); times outside the plan clamp to its ends */
    void sample(State& s) const
    {
        if (m_Wrappers.empty()) return;
        const double t = s.time();
        for (const auto& w : m_Wrappers) if (w.containsTime(t)) { w.sample(s, t); return; }
        if (t < m_Wrappers.front().getStartTime()) { m_Wrappers.front().sample(s, m_Wrappers.front().getStartTime()); return; }
        for (size_t i = 0; i + 1 < m_Wrappers.size(); i++)
            if (t > m_Wrappers[i].getEndTime() && t < m_Wrappers[i + 1].getStartTime()) { m_Wrappers[i].sample(s, m_Wrappers[i].getEndTime()); return; }
        m_Wrappers.back().sample(s, m_Wrappers.back().getEndTime());
    }

private:
    std::vector<DubinsWrapper> m_Wrappers;
};

#endif
