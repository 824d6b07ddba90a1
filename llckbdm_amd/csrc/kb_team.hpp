// Cooperative panels: T workgroups ("a team") share the panel factorisation of ONE ensemble member.
//
// The panel kernels of the two blocked reductions (bidiagonalisation, reference kbdm.py:166; Hessenberg reduction,
// kbdm.py:192) are chains of matrix-vector products with the untouched trailing matrix.  One workgroup keeps about 64 KB of
// loads in flight, so a member's chain runs at the latency-bound rate of ONE compute unit.  A team splits exactly those
// products - the column dots A0^H v by blocks of 32 columns, the row products A0 u / A0 v by chunks of 64 rows - and
// replicates everything else (the small sweeps over the panel factors, the Householder generators): every workgroup of a
// team executes the same instructions on the same numbers there and writes the same values to the panel arrays, so a
// workgroup only ever reads what it wrote itself, what the previous kernel wrote, or what came through the exchange
// buffer.  The split products are defined by a FIXED decomposition (one wavefront per column dot; KB_TEAM_NCG column groups
// per chunk of rows, summed in a fixed order): the team size only decides which workgroup executes which slot.  A member
// therefore gets the same bits from a team of any size, a team of one included.
//
// Synchronisation (MI355X_MICROARCH.md, inter-workgroup visibility): the exchanged vectors are stored write-through (sc1),
// every storing wavefront drains its stores, ONE lane of the workgroup adds to the team's counter, ONE lane polls it with
// relaxed agent-scope loads, the workgroup barrier follows, and every load of the exchanged bytes is an sc1 load.  No
// acquire / release fence, no dependence on the placement of the workgroups (same XCD is faster, never required).  All
// waits are bounded: a team whose wait times out sets its abort word, every workgroup of it returns and the member's status
// word reports it.  The host launches a team kernel only with all its workgroups resident (see launch_svd / launch_eig).
#pragma once
#include "kb_ctx.hpp"

namespace kb {

struct alignas(128) PanelTeamCtl {
    unsigned arrive;     // arrivals over all synchronisations of a run (monotone; the host zeroes it before the run)
    unsigned abort_;     // a wait timed out: every workgroup of the team leaves
    unsigned pad[30];
};

constexpr int KB_TEAM_NCG = 8;           // column groups of a row product (fixed: part of the arithmetic)
constexpr int KB_TEAM_CB = 32;           // columns per ownership block of the column dots
constexpr unsigned long long KB_TEAM_TIMEOUT_TICKS = 1000000000ull;     // 10 s of the 100 MHz wall clock

template <class C>
struct PanelTeam;

#if defined(__HIPCC__)
template <>
struct PanelTeam<DevCtx> {
    int T, role;
    unsigned epoch;          // synchronisations behind this workgroup (all threads keep the same count)
    PanelTeamCtl* ctl;
    cd* xb;                  // exchange buffer of the team (global memory)
    int failed;

    __device__ __forceinline__ void put(int i, cd v) const {
        if (T == 1) { xb[i] = v; return; }
        double* p = reinterpret_cast<double*>(xb + i);
        __hip_atomic_store(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ cd get(int i) const {
        if (T == 1) return xb[i];
        const double* p = reinterpret_cast<const double*>(xb + i);
        cd v;
        v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    }
    // Every workgroup of the team calls it the same number of times.  Stores made with put() before it are visible to
    // get() after it.  false: the team has given up (timeout or abort), the caller returns at once.
    __device__ __forceinline__ bool sync(const DevCtx& ctx) {
        ++epoch;
        if (T == 1) {
            __syncthreads();
            return true;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wavefront drains its sc1 stores
        __syncthreads();
        int* flag = reinterpret_cast<int*>(ctx.smem);           // (the reduction slots: free between block reductions)
        if (threadIdx.x == 0) {
            const unsigned target = epoch * (unsigned)T;
            __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int ok = 1;
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(&ctl->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (__hip_atomic_load(&ctl->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
                if (wall_clock64() - t0 > KB_TEAM_TIMEOUT_TICKS) {
                    __hip_atomic_store(&ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            *flag = ok;
        }
        __syncthreads();
        const int ok = *flag;
        __syncthreads();
        if (!ok) failed = 1;
        return ok != 0;
    }
};
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
// Host simulation: one thread per "workgroup" (tests/hostsim runs the T roles as std::threads that meet in `barrier`).
template <>
struct PanelTeam<HostCtx> {
    int T, role;
    unsigned epoch;
    void* ctl;
    cd* xb;
    int failed;
    void (*barrier)(void*);
    void* barg;
    void put(int i, cd v) const { xb[i] = v; }
    cd get(int i) const { return xb[i]; }
    bool sync(const HostCtx&) {
        ++epoch;
        if (T > 1) barrier(barg);
        return true;
    }
};
#endif

// slots of a team member: blocks / chunks b = role, role + T, ... below nb
KB_HD int team_own_count(int nb, int role, int T) { return role < nb ? (nb - role + T - 1) / T : 0; }

}  // namespace kb
