// Build-container-only probe (tests/test_reference_headers.py): pins the CONSTANTS and the tiny inline arithmetic of the hot path against the
// reference headers that need no Eigen — compiled as they lie under /root/reference/include (never copied):
//   clustering/general.hpp:7-357   Point2D / Point3D arithmetic, the colour enum the ground / debug labels are values of
//   clustering/point_types.hpp     RawPoint / RawPoints (the firing the drop-in class takes)
//   utils/thread_pool.hpp:29-83    ThreadPool: num_threads == 0 runs a job inline in enqueue() (the single-threaded mode the oracle restates)
// It does not lift "parity unpinned" (the algorithm, cc.cpp, needs Eigen3): it removes the constants and these helpers from the unpinned set.
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <list>
#include <memory>
#include <mutex>
#include <queue>
#include <random>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace ref
{
// (the headers' own #include lines are no-ops here: every standard header they name is included above)
#include <continuous_clustering/clustering/general.hpp>
#include <continuous_clustering/clustering/point_types.hpp>
#include <continuous_clustering/utils/thread_pool.hpp>
} // namespace ref

#include "../../include/cc_hip.h"
#include "../../continuous_clustering_amd/csrc/continuous_clustering.hpp" // the drop-in class's mirror of the reference's types
#include "../../oracle/cc_oracle.cpp"                                     // the checker's restatement (anonymous namespace: Oracle::len2, Oracle::close_enough)

namespace rc = ref::continuous_clustering;
namespace cc = continuous_clustering;

// ---- label values: include/cc_hip.h:45-62 against general.hpp:208-357 (cc.hpp:15-22 names the colours; checked as text by the test) ----
#define EQ(a, b) (int(a) == int(b))
static_assert(EQ(CC_GP_UNKNOWN, rc::WHITE) && EQ(CC_GP_GROUND, rc::GREEN) && EQ(CC_GP_OBSTACLE, rc::RED) && EQ(CC_GP_EGO_VEHICLE, rc::MAGENTA) && EQ(CC_GP_FOG, rc::LIGHTGRAY),
              "ground point labels");
static_assert(EQ(CC_DBG_GRAY, rc::GRAY) && EQ(CC_DBG_ORANGE, rc::ORANGE) && EQ(CC_DBG_GREEN, rc::GREEN) && EQ(CC_DBG_YELLOWGREEN, rc::YELLOWGREEN) && EQ(CC_DBG_YELLOW, rc::YELLOW) &&
                  EQ(CC_DBG_RED, rc::RED) && EQ(CC_DBG_DARKRED, rc::DARKRED) && EQ(CC_DBG_VIOLET, rc::VIOLET) && EQ(CC_DBG_LIGHTGRAY, rc::LIGHTGRAY) && EQ(CC_DBG_WHITE, rc::WHITE),
              "debug ground point labels");
static_assert(EQ(cc::GP_UNKNOWN, rc::WHITE) && EQ(cc::GP_GROUND, rc::GREEN) && EQ(cc::GP_OBSTACLE, rc::RED) && EQ(cc::GP_EGO_VEHICLE, rc::MAGENTA) && EQ(cc::GP_FOG, rc::LIGHTGRAY),
              "the drop-in class's enum");

// ---- RawPoint / RawPoints layout: csrc/continuous_clustering.hpp against point_types.hpp:10-28 ----
static_assert(sizeof(cc::RawPoint) == sizeof(rc::RawPoint) && alignof(cc::RawPoint) == alignof(rc::RawPoint), "RawPoint size");
static_assert(offsetof(cc::RawPoint, x) == offsetof(rc::RawPoint, x) && offsetof(cc::RawPoint, y) == offsetof(rc::RawPoint, y) && offsetof(cc::RawPoint, z) == offsetof(rc::RawPoint, z) &&
                  offsetof(cc::RawPoint, firing_index) == offsetof(rc::RawPoint, firing_index) && offsetof(cc::RawPoint, intensity) == offsetof(rc::RawPoint, intensity) &&
                  offsetof(cc::RawPoint, stamp) == offsetof(rc::RawPoint, stamp) &&
                  offsetof(cc::RawPoint, globally_unique_point_index) == offsetof(rc::RawPoint, globally_unique_point_index),
              "RawPoint field offsets");
static_assert(sizeof(cc::RawPoints) == sizeof(rc::RawPoints) && offsetof(cc::RawPoints, stamp) == offsetof(rc::RawPoints, stamp) && offsetof(cc::RawPoints, points) == offsetof(rc::RawPoints, points),
              "RawPoints layout");

static uint32_t bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

int main()
{
    int bad = 0;
    std::mt19937 rng(20260930);
    std::uniform_real_distribution<float> big(-150.f, 150.f), small(-2.f, 2.f);
    Oracle o; // default max_distance_squared = 0.7f * 0.7f (cc.hpp:74 max_distance 0.7)
    long n_close = 0, n_len = 0;
    for (int i = 0; i < 2000000; i++)
    {
        // association distance test, cc.cpp:638-641: (a - b).lengthSquared() < max_distance_squared  <->  Oracle::close_enough
        Cell a{}, b{};
        a.x = big(rng), a.y = big(rng), a.z = small(rng);
        const float s = (i & 1) ? 0.45f : 1.0f; // half of the pairs near the 0.7 m threshold
        b.x = a.x + small(rng) * s, b.y = a.y + small(rng) * s, b.z = a.z + small(rng) * s;
        const rc::Point3D pa(a.x, a.y, a.z), pb(b.x, b.y, b.z);
        const rc::Point3D d = pa - pb;
        const bool want = d.lengthSquared() < o.max_distance_squared;
        if (want != o.close_enough(a, b))
            bad++, std::printf("close_enough differs at pair %d\n", i);
        n_close += want;
        // ground segmentation, cc.cpp:396-400 etc.: Point2D(xy().length(), z) <-> Oracle::len2
        const float l_ref = pa.xy().length(), l_orc = Oracle::len2(a.x, a.y);
        if (bits(l_ref) != bits(l_orc))
            bad++, std::printf("len2 differs at %d: %a vs %a\n", i, l_ref, l_orc);
        if (bits(rc::Point2D(a.x, a.y).length()) != bits(l_orc) || bits(pa.lengthXY()) != bits(l_orc))
            bad++;
        n_len++;
    }
    // ThreadPool with num_threads == 0 (thread_pool.hpp:31-35,58-64): enqueue() runs the job on the caller's thread before it returns
    {
        rc::ThreadPool<int> pool("probe");
        std::vector<int> seen;
        const auto me = std::this_thread::get_id();
        bool same_thread = true;
        pool.init([&](int j) { seen.push_back(j); same_thread = same_thread && std::this_thread::get_id() == me; }, 0);
        for (int j = 0; j < 5; j++)
        {
            pool.enqueue(int(j));
            if ((int) seen.size() != j + 1 || seen.back() != j)
                bad++, std::printf("ThreadPool(0): job %d not run inline\n", j);
        }
        if (!same_thread || pool.getNumberOfUnprocessedJobs() != 0)
            bad++, std::printf("ThreadPool(0): not the caller's thread / jobs left\n");
    }
    std::printf("pairs %d close %ld lengths %ld sizeof(RawPoint) %zu bad %d\n", 2000000, n_close, n_len, sizeof(rc::RawPoint), bad);
    return bad ? 1 : 0;
}
