/* NodeStub -- the tests' ControlReceiver (reference: test/NodeStub.{h,cpp}), with the timedOut() override the
 * reference's stub lacks and an optional frozen clock so solves are deterministic. */
#ifndef MPC_AMD_HOST_TEST_NODESTUB_H
#define MPC_AMD_HOST_TEST_NODESTUB_H

#include <mutex>
#include <vector>

#include "controller.h"

class NodeStub : public ControlReceiver
{
public:
    explicit NodeStub(bool frozenClock = false, double frozenTime = 0.0, bool quiet = true)
        : m_Frozen(frozenClock), m_Time(frozenTime), m_Quiet(quiet) {}

    void receiveControl(double rudder, double throttle) override
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        controls.emplace_back(rudder, throttle);
        if (!m_Quiet) std::cerr << "NodeStub received control " << rudder << ", " << throttle << std::endl;
    }
    void displayTrajectory(const std::vector<State>& trajectory, bool plannerTrajectory, bool achievable) override
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        lastTrajectory = trajectory;
        if (displayed == 0) firstTrajectory = trajectory;
        displayed++;
        lastAchievable = achievable;
        (void)plannerTrajectory;
    }
    void timedOut() override { std::lock_guard<std::mutex> lock(m_Mutex); timedOutCalls++; }
    double getTime() const override { return m_Frozen ? m_Time : Controller::getTime(); }

    std::vector<std::pair<double, double>> controls;
    std::vector<State> lastTrajectory, firstTrajectory;
    int displayed = 0, timedOutCalls = 0;
    bool lastAchievable = true;            /* the m_Achievable the controller passed to its latest displayTrajectory */
    std::vector<std::pair<double, double>> controlsNow() { std::lock_guard<std::mutex> lock(m_Mutex); return controls; }
    int displayedNow() { std::lock_guard<std::mutex> lock(m_Mutex); return displayed; }
    bool achievableNow() { std::lock_guard<std::mutex> lock(m_Mutex); return lastAchievable; }

private:
    bool m_Frozen;
    double m_Time;
    bool m_Quiet;
    std::mutex m_Mutex;
};

#endif
