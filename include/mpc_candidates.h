/*
 * mpc_candidates.h -- definition of the SEEDED SYNTHETIC candidate sets used by the
 * benchmark configurations of BASELINE.json (configs 2-5).
 *
 * The reference (afb2001/mpc) has no API that accepts an explicit candidate list; its only
 * candidate source is the limited-branching tree of controller.cpp:121-146.  The seeded sets
 * are a workload definition of this repository.  They are pure 64-bit
 * integer arithmetic, so host C, the CPU oracle and the HIP kernels produce identical bits.
 *
 * control(seed, ordinal, level):
 *   ordinal 0          -> (0, 1)                     "straight ahead, full throttle"
 *   ordinal 1          -> (last_rudder, last_throttle) "hold the last control"
 *   otherwise          -> rudder   = -1 + 0.0625 * (u % 33)      u = bits 0..31 of the hash
 *                         throttle =      0.125  * (w % 9)       w = bits 32..63 of the hash
 * i.e. rudder ~ U[-1,1] quantised to the reference's default rudder granularity (cfg/mpc.cfg:8)
 * and throttle ~ U[0,1] quantised to its throttle granularity (cfg/mpc.cfg:9).  All values are
 * exactly representable in binary32, so the FP32 and FP64 paths see the same controls.
 */
#ifndef MPC_CANDIDATES_H
#define MPC_CANDIDATES_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MPC_HD __host__ __device__ static inline
#else
#define MPC_HD static inline
#endif

/* SplitMix64 finaliser (Steele, Lea, Flood 2014) used as a counter-based generator. */
MPC_HD uint64_t mpc_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

MPC_HD uint64_t mpc_candidate_hash(uint64_t seed, uint64_t ordinal, uint32_t level)
{
    return mpc_mix64(mpc_mix64(seed ^ (ordinal * 0xD1342543DE82EF95ull)) + (uint64_t)level);
}

/* index form: rudder index 0..32, throttle index 0..8 */
MPC_HD void mpc_candidate_index(uint64_t seed, uint64_t ordinal, uint32_t level,
                                uint32_t *ri, uint32_t *ti)
{
    uint64_t h = mpc_candidate_hash(seed, ordinal, level);
    *ri = (uint32_t)(h & 0xFFFFFFFFull) % 33u;
    *ti = (uint32_t)(h >> 32) % 9u;
}

MPC_HD void mpc_candidate_control(uint64_t seed, uint64_t ordinal, uint32_t level,
                                  double last_rudder, double last_throttle,
                                  double *rudder, double *throttle)
{
    if (ordinal == 0) { *rudder = 0.0; *throttle = 1.0; return; }
    if (ordinal == 1) { *rudder = last_rudder; *throttle = last_throttle; return; }
    uint32_t ri, ti;
    mpc_candidate_index(seed, ordinal, level, &ri, &ti);
    *rudder = -1.0 + 0.0625 * (double)ri;
    *throttle = 0.125 * (double)ti;
}

#endif /* MPC_CANDIDATES_H */
