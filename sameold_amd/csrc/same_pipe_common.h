// same_pipe_common.h -- what the wavefront pipelines (same_kernels_pipe.hip, same_kernels_sym.hip) share: LDS mailbox
// words, the LDS-only step barrier, the burst-pool copy and the event wavefront's log context.
#pragma once

#include <hip/hip_runtime.h>

#include "same_dev_common.h"
#include "same_device.h"

namespace same {

// Mailbox words in LDS.  The address space is part of the type: a plain `volatile uint32_t *`
// derived from the LDS base degrades to a generic pointer, and every access becomes a
// system-scope FLAT instruction followed by s_waitcnt vmcnt(0).
typedef volatile __attribute__((address_space(3))) uint32_t lds_u32;

// Workgroup barrier for stages that talk through LDS only.  __syncthreads() also waits for the
// wavefront's outstanding GLOBAL memory operations (vmcnt(0)): stage 1's input prefetch for the
// next block and stage 3's framer-byte and event stores would be waited for at every step,
// although no other wavefront ever reads them.  LDS traffic is ordered by lgkmcnt alone.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Stage 3 keeps the framer rows; a finished burst is copied into the pool here, its slot travels on.
__device__ __forceinline__ uint32_t burst_to_pool(const State &S, const Output &O, uint32_t c)
{
    const uint32_t b = atomicAdd(O.n_events + 1, 1u);
    if (b >= O.burst_cap) { atomicOr(O.n_events + 2, 2u); return 0xffffffffu; }
    copy_burst_row(O.bursts + (size_t)b * kBurstCap, S.fr_msg + (size_t)c * kBurstCap);
    return b;
}

// Stage 4's context: the event log.  emit_event takes a slot with a returning atomic per event --
// an L2 round trip on the critical path of whichever stage emits.  Here the wavefront reserves
// runs of 64 slots (one atomic per run) and hands them out with a ballot; slots of a run that stay
// unused are marked kDevEventNone for the host to skip.  The deadline ring stays in HBM.
constexpr uint32_t kEvChunk = 64;
struct IoCtx : TickRingGlobal {
    lds_u32 *chunk;            // LDS: [0] first slot of the current run, [1] slots of it already handed out
    uint32_t pending_slot;     // burst-pool slot of the Burst event about to be emitted
    __device__ __forceinline__ void mark(int) const {}
    __device__ __forceinline__ void emit(const Params &P, const State &S, const Output &O, uint32_t c, uint32_t kind,
                                         uint64_t sample_counter, uint64_t symbols, uint32_t burst_len)
    {
        const uint64_t act = __builtin_amdgcn_ballot_w64(true);          // the lanes emitting right now
        const uint32_t n = (uint32_t)__popcll(act);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
        uint32_t base = chunk[0], used = chunk[1];
        if (used + n > kEvChunk) {                                       // wave-uniform
            for (uint32_t i = used + rank; i < kEvChunk; i += n)
                if (base + i < O.event_cap) { O.events[base + i].channel = 0; O.events[base + i].kind = kDevEventNone; }
            uint32_t nb = 0;
            if (rank == 0u) nb = atomicAdd(O.n_events, kEvChunk);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);    // first active lane = rank 0
            used = 0;
        }
        const uint32_t e = base + used + rank;
        chunk[0] = base; chunk[1] = used + n;
        if (e < O.event_cap) {
            DevEvent ev;
            ev.channel = c; ev.kind = kind; ev.sample_counter = sample_counter;
            ev.symbol_count = symbols; ev.burst_len = burst_len;
            ev.burst_slot = kind == 3u ? pending_slot : 0xffffffffu;
            O.events[e] = ev;
        } else {
            atomicOr(O.n_events + 2, 1u);
        }
    }
    // end of the launch, all lanes: mark what is left of the current run as empty
    __device__ __forceinline__ void retire(const Output &O, uint32_t lane, uint32_t lanes)
    {
        const uint32_t base = chunk[0], used = chunk[1];
        for (uint32_t i = used + lane; i < kEvChunk; i += lanes)
            if (base + i < O.event_cap) { O.events[base + i].channel = 0; O.events[base + i].kind = kDevEventNone; }
    }
};

// The event wavefront's context with the deadline ring (kTickRing instants and their
// count per channel) in LDS for the launch.  In the HBM state arrays every burst and every expired deadline cost that wavefront a chain
// of dependent global round trips -- count, oldest deadline, the ring moved down -- and it is the wavefront that also issues
// the input prefetch (same_kernels_sym.hip) or runs the matched filters stage 2 waits for (same_kernels_pipe.hip).  One in
// eight steps of a 64-channel workgroup has such a lane: 8 % of a 22.05 kHz launch with the transport layer on, 12 % at 48 kHz.
constexpr uint32_t kIoRingWords = (2u * (uint32_t)kTickRing + 1u) * kWave;
struct IoCtxLds : IoCtx {
    lds_u32 *tk;               // this lane's column: deadline i in words [2 i][lane], [2 i + 1][lane]; the count behind them
    __device__ __forceinline__ uint32_t tk_count(const State &, uint32_t) const { return tk[2 * kTickRing * kWave]; }
    __device__ __forceinline__ void tk_set_count(const State &, uint32_t, uint32_t n) const { tk[2 * kTickRing * kWave] = n; }
    __device__ __forceinline__ uint64_t tk_at(const State &, uint32_t, uint32_t, uint32_t i) const
    {
        return (uint64_t)tk[2u * i * kWave] | ((uint64_t)tk[(2u * i + 1u) * kWave] << 32);
    }
    __device__ __forceinline__ void tk_set(const State &, uint32_t, uint32_t, uint32_t i, uint64_t v) const
    {
        tk[2u * i * kWave] = (uint32_t)v; tk[(2u * i + 1u) * kWave] = (uint32_t)(v >> 32);
    }
    __device__ __forceinline__ void ring_load(const Params &P, const State &S, uint32_t c)
    {
        if (!P.ticks) return;
        const TickRingGlobal G;
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)kTickRing; ++i) tk_set(S, P.n_channels, c, i, G.tk_at(S, P.n_channels, c, i));
        tk_set_count(S, c, G.tk_count(S, c));
    }
    __device__ __forceinline__ void ring_store(const Params &P, const State &S, uint32_t c) const
    {
        if (!P.ticks) return;
        const TickRingGlobal G;
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)kTickRing; ++i) G.tk_set(S, P.n_channels, c, i, tk_at(S, P.n_channels, c, i));
        G.tk_set_count(S, c, tk_count(S, c));
    }
};

}  // namespace same
