// same_profile.h -- cycle-attribution instrumentation of the standard-rate kernels (same_kernels_pipe.hip,
// same_kernels_fast.hip).
//
// A SAME_PROFILE=1 build (python -m sameold_amd.build with SAME_PROFILE set; never the shipped one) stamps
// the shader clock at section boundaries of workgroup 0's wavefronts and tools/run_once.py prints the
// totals (profiles/r0x_cycle_attribution.txt).  Everything is behind the macros below, which expand to
// nothing in a normal build, so the kernel source carries no conditional compilation of its own.
//   SAME_P3_MARKS=1  additionally splits the symbol path (stage 3) into its sections,
//   SAME_P1_SPLIT=1  splits stage 1 at "inputs have arrived".
#pragma once

#include <hip/hip_runtime.h>

#ifdef SAME_PROFILE
namespace same {
// Section marks of the symbol path / the one-wavefront kernel's block loop.  The accumulators live in one LDS
// location shared by the whole wavefront: a mark inside a divergent region is executed by the active lanes
// only, and every one of them reads and writes the same values, so the totals are per wavefront whichever
// lanes were active.  pl[0] = time of the previous mark, pl[1 + i] = cycles attributed to section i;
// section 8 is the cost of a mark itself (two marks back to back).
struct ProfMarks {
    unsigned long long *pl;
    bool pon = true;
    __device__ __forceinline__ void mark(int i)
    {
        if (!pon) return;
        const unsigned long long t = clock64();
        volatile unsigned long long *p = pl;
        const unsigned long long prev = p[0];
        p[0] = t;
        p[1 + i] = p[1 + i] + (t - prev);
    }
};
// one-wavefront kernel: shader-clock time per section of the block loop, summed over the blocks of wavefront 0
static __device__ unsigned long long g_same_prof[9];
// per stage of workgroup 0: cycles working, waiting at the step barrier, handling feedback
static __device__ unsigned long long g_same_prof_pipe[15];    // roles 0..4 (4 = DC wave)
// [role] = HW_ID of workgroup 0's wavefront in that role (SIMD = bits 5:4); [5] cycles stage 2 polled
// the helper for the filter magnitudes, [6] cycles the helper spent filtering, [7] second instants of a block
static __device__ unsigned long long g_same_prof_hw[8];
// stage 2 of workgroup 0, cycles per section of a block: [0] mark filter + hypot, [1] polling the helper,
// [2] combine + timing loop + next instant, [3] posting (mailboxes), [4] checkpoint + loop + barrier entry
static __device__ unsigned long long g_same_prof_s2[8];
}  // namespace same
// relaxed kernel (same_kernels_relaxed.hip), wavefront 0: shader-clock time per section of a sub-block
// [0] DC blocker (with the wait for its inputs) [1] AGC + window push [2] matched filters [3] timing loop + symbol path
// [4] AGC replay, loop ends [5] hand-over check; [6] sub-blocks [7] TED passes
namespace same { static __device__ unsigned long long g_same_prof_relaxed[8]; }
#define RX_T0() unsigned long long rx_acc[6] = {0, 0, 0, 0, 0, 0}, rx_n[2] = {0, 0}, rx_t = clock64()
#define RX_LAP(i) do { const unsigned long long t_ = clock64(); rx_acc[i] += t_ - rx_t; rx_t = t_; } while (0)
#define RX_COUNT(i) do { rx_n[i] += 1; } while (0)
#define RX_REPORT() do { if (blockIdx.x == 0 && lane == 0) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&same::g_same_prof_relaxed[i_], rx_acc[i_]); \
        atomicAdd(&same::g_same_prof_relaxed[6], rx_n[0]); atomicAdd(&same::g_same_prof_relaxed[7], rx_n[1]); } } while (0)
#define RELAXED_PROFILE_EXPORTS()                                                                               \
    extern "C" int same_debug_profile_relaxed(unsigned long long *out8, int reset)                              \
    {                                                                                                           \
        unsigned long long z[8] = {0};                                                                          \
        if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(same::g_same_prof_relaxed), sizeof(z)) != hipSuccess) return -1;       \
        if (reset && hipMemcpyToSymbol(HIP_SYMBOL(same::g_same_prof_relaxed), z, sizeof(z)) != hipSuccess) return -1;   \
        return 0;                                                                                               \
    }
#define PIPE_PROF_TAP_PAD 20
// symbol-paced pipeline (same_kernels_sym.hip), workgroup 0: [3 r .. 3 r + 2] cycles role r (S, T, A, E, Y1, Y2) worked / waited at
// the step barrier / spent on feedback; [18] launches, [19] steps, [20] passes of E with a symbol, [22] cycles of E's filters,
// [23] of its timing updates and posting, [24] waiting for A's soft sample.  The group of 64 columns that reports: SAME_PIPE_PRIO >> 16
namespace same { static __device__ unsigned long long g_same_prof_sym[32]; }
#define SYM_REPORT(role_) do { if (vwg == ((uint32_t)P.knob_prio >> 16) && lane == 0) { same::g_same_prof_sym[3 * (role_)] += p3_work; \
        same::g_same_prof_sym[3 * (role_) + 1] += p3_wait; same::g_same_prof_sym[3 * (role_) + 2] += p3_fb; } } while (0)
#define SYM_COUNT(i_, n_) do { if (vwg == ((uint32_t)P.knob_prio >> 16) && lane == 0) same::g_same_prof_sym[i_] += (n_); } while (0)
#define SYM_TCOUNT(i_, n_) do { symt_n[(i_) - 20] += (n_); } while (0)
#define SYM_T_DECL() unsigned long long symt_acc[3] = {0, 0, 0}, symt_n[2] = {0, 0}, symt_t = 0
#define SYM_T_BEGIN() do { symt_t = clock64(); } while (0)
#define SYM_T_LAP(i_) do { const unsigned long long t_ = clock64(); symt_acc[(i_) - 22] += t_ - symt_t; symt_t = t_; } while (0)
#define SYM_T_REPORT() do { if (vwg == ((uint32_t)P.knob_prio >> 16) && lane == 0) { for (int i_ = 0; i_ < 3; ++i_) same::g_same_prof_sym[22 + i_] += symt_acc[i_]; \
        same::g_same_prof_sym[20] += symt_n[0]; same::g_same_prof_sym[21] += symt_n[1]; } } while (0)
// timeline of the reporting group: shader clock at mark k (0 step begins, 1 first wait over, 2 work done / published) of role r in steps
// SYM_TRACE_S0 .. + SYM_TRACE_N - 1 of a launch (tools/sym_probe.py timeline)
#define SYM_TL_WORDS 0u      /* (the light timeline of release builds keeps its marks in LDS: SAME_SYM_TL) */
#define SYM_TRACE_S0 600u
#define SYM_TRACE_N 12u
namespace same { static __device__ unsigned long long g_same_prof_trace[6 * 12 * 4]; }
#define SYM_TRACE(role_, s_, k_) do { if (vwg == ((uint32_t)P.knob_prio >> 16) && lane == 0 && (s_) >= SYM_TRACE_S0 && (s_) < SYM_TRACE_S0 + SYM_TRACE_N) \
        same::g_same_prof_trace[((role_) * SYM_TRACE_N + ((s_) - SYM_TRACE_S0)) * 4u + (k_)] = clock64(); } while (0)
#define SYM_PROFILE_EXPORTS()                                                                                   \
    extern "C" int same_debug_profile_sym_trace(unsigned long long *out288)                                     \
    { return hipMemcpyFromSymbol(out288, HIP_SYMBOL(same::g_same_prof_trace), 288 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1; } \
    extern "C" int same_debug_profile_sym(unsigned long long *out32, int reset)                                 \
    {                                                                                                           \
        unsigned long long z[32] = {0};                                                                         \
        if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(same::g_same_prof_sym), sizeof(z)) != hipSuccess) return -1;  \
        if (reset && hipMemcpyToSymbol(HIP_SYMBOL(same::g_same_prof_sym), z, sizeof(z)) != hipSuccess) return -1; \
        return 0;                                                                                               \
    }                                                                                                           \
    extern "C" int same_debug_profile_sym_marks(unsigned long long *out15, int reset)   /* SAME_P3_MARKS build: Y's sections */ \
    {                                                                                                           \
        unsigned long long z[15] = {0};                                                                         \
        if (hipMemcpyFromSymbol(out15, HIP_SYMBOL(same::g_same_prof_pipe), sizeof(z)) != hipSuccess) return -1; \
        if (reset && hipMemcpyToSymbol(HIP_SYMBOL(same::g_same_prof_pipe), z, sizeof(z)) != hipSuccess) return -1; \
        return 0;                                                                                               \
    }
#define FAST_MARKS_BEGIN(X_, lds_, nt_) do { (X_).pl = reinterpret_cast<unsigned long long *>((lds_) + (nt_) * 4); \
        for (int i_ = 0; i_ < 10; ++i_) (X_).pl[i_] = 0; } while (0)
#define FAST_MARKS_START(X_) do { (X_).pl[0] = clock64(); } while (0)
#define FAST_MARKS_REPORT(X_) do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 9; ++i_) atomicAdd(&g_same_prof[i_], (X_).pl[1 + i_]); } while (0)
#define FAST_PROFILE_EXPORTS()                                                                                  \
    extern "C" int same_debug_profile(unsigned long long *out9, int reset)                                      \
    {                                                                                                           \
        unsigned long long z[9] = {0};                                                                          \
        if (hipMemcpyFromSymbol(out9, HIP_SYMBOL(same::g_same_prof), sizeof(z)) != hipSuccess) return -1;       \
        if (reset && hipMemcpyToSymbol(HIP_SYMBOL(same::g_same_prof), z, sizeof(z)) != hipSuccess) return -1;   \
        return 0;                                                                                               \
    }
#define S2_BEGIN() unsigned long long s2_acc[5] = {0, 0, 0, 0, 0}, s2_t = clock64()
#define S2_LAP(i) do { const unsigned long long t_ = clock64(); s2_acc[i] += t_ - s2_t; s2_t = t_; } while (0)
#define S2_REPORT() do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 5; ++i_) g_same_prof_s2[i_] += s2_acc[i_]; } while (0)
#define P3_HWID(role_) do { if (blockIdx.x == 0 && lane == 0) g_same_prof_hw[role_] = __builtin_amdgcn_s_getreg((31 << 11) | 4); } while (0)
#define P3_T0() unsigned long long p3_work = 0, p3_wait = 0, p3_fb = 0, p3_t = clock64()
#define P3_LAP(acc) do { const unsigned long long t_ = clock64(); acc += t_ - p3_t; p3_t = t_; } while (0)
#ifdef SAME_P3_MARKS
#define P3_REPORT(role_) do {} while (0)
#define P3_MARKS_BEGIN(X_, lds_, nt_) do { (X_).pl = reinterpret_cast<unsigned long long *>((lds_) + (nt_) * 4); (X_).pon = true; \
        for (int i_ = 0; i_ < 10; ++i_) (X_).pl[i_] = 0; (X_).pl[0] = clock64(); } while (0)
#define P3_MARKS_REPORT(X_) do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_same_prof_pipe[i_], (X_).pl[1 + 2 + i_]); } while (0)
#else
#define P3_REPORT(role_) do { if (blockIdx.x == 0 && lane == 0) { g_same_prof_pipe[3 * (role_)] += p3_work; \
        g_same_prof_pipe[3 * (role_) + 1] += p3_wait; g_same_prof_pipe[3 * (role_) + 2] += p3_fb; } } while (0)
#define P3_MARKS_BEGIN(X_, lds_, nt_) do { (X_).pl = reinterpret_cast<unsigned long long *>((lds_) + (nt_) * 4); (X_).pon = false; } while (0)
#define P3_MARKS_REPORT(X_) do {} while (0)
#endif
#ifdef SAME_P1_SPLIT
#define P1_INPUTS_ARRIVED(n_) do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n_) : "memory"); P3_LAP(p3_fb); } while (0)   /* (reported in the "feedback" column) */
#else
#define P1_INPUTS_ARRIVED(n_) do {} while (0)
#endif
#define SPIN_BEGIN() const unsigned long long spin_t0 = clock64()
#define SPIN_END() do { if (blockIdx.x == 0 && lane == (uint32_t)__builtin_amdgcn_readfirstlane((int)lane)) g_same_prof_hw[5] += clock64() - spin_t0; } while (0)
#define HELP_BEGIN() const unsigned long long help_t0 = clock64()
#define HELP_END(on_) do { if ((on_) && blockIdx.x == 0 && lane == 0) g_same_prof_hw[6] += clock64() - help_t0; } while (0)
#define COUNT_SECOND_INSTANT() atomicAdd(&g_same_prof_hw[7], 1ull)          /* lanes that took this path (any workgroup) */
/* knock-out experiments (results are garbage, the timing says what the step is waiting for): SAME_PIPE_PRIO bits
   8 helper's event half, 16 stage 3's symbol path, 32 stage 2's block, 64 stage 1's AGC block, 128 DC blocker, 256 helper's filters */
#define PROF_SKIP(P_, bit_) (((P_).knob_prio & (bit_)) != 0)
#define PIPE_PROFILE_EXPORTS()                                                                                  \
    static int prof_fetch_(const void *sym, unsigned long long *out, size_t n, int reset)                       \
    {                                                                                                           \
        unsigned long long z[15] = {0};                                                                         \
        if (hipMemcpyFromSymbol(out, sym, n * sizeof(unsigned long long)) != hipSuccess) return -1;             \
        if (reset && hipMemcpyToSymbol(sym, z, n * sizeof(unsigned long long)) != hipSuccess) return -1;        \
        return 0;                                                                                               \
    }                                                                                                           \
    extern "C" int same_debug_profile_s2(unsigned long long *out8, int reset) { return prof_fetch_(HIP_SYMBOL(same::g_same_prof_s2), out8, 8, reset); }   \
    extern "C" int same_debug_profile_hw(unsigned long long *out8, int reset) { return prof_fetch_(HIP_SYMBOL(same::g_same_prof_hw), out8, 8, reset); }   \
    extern "C" int same_debug_profile_pipe(unsigned long long *out15, int reset) { return prof_fetch_(HIP_SYMBOL(same::g_same_prof_pipe), out15, 15, reset); }
#else
namespace same { struct ProfMarks { __device__ __forceinline__ void mark(int) {} }; }
#define RX_T0() do {} while (0)
#define RX_LAP(i) do {} while (0)
#define RX_COUNT(i) do {} while (0)
#define RX_REPORT() do {} while (0)
#define RELAXED_PROFILE_EXPORTS()
#define PIPE_PROF_TAP_PAD 0
#ifndef SAME_SYM_TL
#define SYM_REPORT(role_) do {} while (0)
#endif
#define SYM_COUNT(i_, n_) do {} while (0)
#define SYM_TCOUNT(i_, n_) do {} while (0)
#define SYM_T_DECL() do {} while (0)
#define SYM_T_BEGIN() do {} while (0)
#define SYM_T_LAP(i_) do {} while (0)
#define SYM_T_REPORT() do {} while (0)
#ifdef SAME_SYM_TL
// A timeline of the symbol-paced pipeline that barely disturbs it (release code otherwise; python -m sameold_amd.build with
// SAME_SYM_TL set): lane 0 of every role of ONE group of 64 columns notes the shader clock's low word in LDS at mark k (0 step begins,
// 1 first wait over, 2 about to publish, 3 role-specific) of steps SYM_TRACE_S0 .. + SYM_TRACE_N - 1; the rows go to a global array
// when the role ends.  tools/sym_probe.py timeline.
#ifndef SYM_TL_GROUP
#define SYM_TL_GROUP 0u
#endif
#define SYM_TRACE_S0 600u
#define SYM_TRACE_N 12u
#define SYM_TL_WORDS (6u * SYM_TRACE_N * 4u)
namespace same { static __device__ unsigned long long g_same_prof_trace[6 * 12 * 4]; }
#define SYM_TRACE(role_, s_, k_) do { if (vwg == SYM_TL_GROUP && lane == 0 && (s_) >= SYM_TRACE_S0 && (s_) < SYM_TRACE_S0 + SYM_TRACE_N) \
        tlbox[((role_) * SYM_TRACE_N + ((s_) - SYM_TRACE_S0)) * 4u + (k_)] = (uint32_t)clock64(); } while (0)
#define SYM_REPORT(role_) do { if (vwg == SYM_TL_GROUP && lane < SYM_TRACE_N * 4u) \
        same::g_same_prof_trace[(role_) * SYM_TRACE_N * 4u + lane] = tlbox[(role_) * SYM_TRACE_N * 4u + lane]; } while (0)
#define SYM_PROFILE_EXPORTS()                                                                                   \
    extern "C" int same_debug_profile_sym_trace(unsigned long long *out288)                                     \
    { return hipMemcpyFromSymbol(out288, HIP_SYMBOL(same::g_same_prof_trace), 288 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1; }
#else
#define SYM_TL_WORDS 0u
#define SYM_PROFILE_EXPORTS()
#ifdef SYM_ASM_MARKS   /* analysis listings only (tools/sym_role_mix.py): comments in the assembly where a role's step begins and ends */
#define SYM_TRACE(role_, s_, k_) asm volatile("; SYMMARK " #role_ " " #k_)
#else
#define SYM_TRACE(role_, s_, k_) do {} while (0)
#endif
#endif
#define FAST_MARKS_BEGIN(X_, lds_, nt_) do {} while (0)
#define FAST_MARKS_START(X_) do {} while (0)
#define FAST_MARKS_REPORT(X_) do {} while (0)
#define FAST_PROFILE_EXPORTS()
#define S2_BEGIN() do {} while (0)
#define S2_LAP(i) do {} while (0)
#define S2_REPORT() do {} while (0)
#define P3_HWID(role_) do {} while (0)
#define P3_T0() do {} while (0)
#define P3_LAP(acc) do {} while (0)
#define P3_REPORT(role_) do {} while (0)
#define P3_MARKS_BEGIN(X_, lds_, nt_) do {} while (0)
#define P3_MARKS_REPORT(X_) do {} while (0)
#define P1_INPUTS_ARRIVED(n_) do {} while (0)
#define SPIN_BEGIN() do {} while (0)
#define SPIN_END() do {} while (0)
#define HELP_BEGIN() do {} while (0)
#define HELP_END(on_) do {} while (0)
#define COUNT_SECOND_INSTANT() do {} while (0)
#define PROF_SKIP(P_, bit_) false
#define PIPE_PROFILE_EXPORTS()
#endif
