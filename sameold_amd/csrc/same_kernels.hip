// same_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched SAME demodulator.
//
// One wavefront lane owns one channel (one `SameReceiver`); a workgroup is one 64-lane
// wavefront.  All channels advance in lockstep over a time-major input x[t][channel],
// so each global load of a wavefront is 256 contiguous bytes.
//
// Arithmetic contract: every f32 operation is evaluated in the order, and with the
// single IEEE rounding, of the reference's scalar Rust code (no FMA contraction: this
// file is built with -ffp-contract=off; IEEE division; hypot as
// (float)sqrt((double)re*re + (double)im*im) == glibc hypotf).  The kernel is therefore
// bit-identical to the scalar CPU restatement used as the test oracle: identical TED
// instants, soft symbols, bytes and
// event sample counters.  Only the *schedule* differs (see "deferred TED" below).
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "same_dev_common.h"
#include "same_device.h"
#include "same_launch.h"

namespace same {

// ---------------------------------------------------------------------------------
// The demodulation kernel.
//
// Deferred TED: time advances in blocks of B samples for all lanes together.  B is
// chosen on the host so that a lane's TED can fire at most once per block (the timing
// loop cannot command a period shorter than B + 0.5 samples).  Within a block every
// lane runs the uniform per-sample work (DC blocker, AGC, window push, sample clock)
// and notes the offset at which its TED fired; after the block, all lanes that fired
// evaluate their matched filters TOGETHER over the window as it stood at their own fire
// offset.  Without this, each of the ~21 different phases in a wavefront would pay for
// its own divergent 42-tap filter pass.
//
// Nothing after a TED instant can influence the per-sample stages of the same block
// except the AGC lock (agc.lock(true) on sync, lock(false) in end()); when the symbol
// processing of a lane changes its lock, the AGC outputs after the fire offset are
// recomputed from the saved gain ("replay"), which reproduces the sequential result
// exactly.
// ---------------------------------------------------------------------------------
template <int B, typename SampleT>
__global__ __launch_bounds__(kWave) void demod_kernel(Params P, State S, Output O,
                                                      const float4 *__restrict__ taps,
                                                      const SampleT *__restrict__ x,
                                                      uint32_t n_samples, uint64_t counter0)
{
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t C = P.n_channels;
    const uint32_t c_raw = blockIdx.x * kWave + lane;
    if (c_raw >= C) return;                                 // no barriers below: tail lanes just leave
    const uint32_t c = c_raw;
    const uint32_t DL = P.dc_len;
    float *ff = lds;                                        // [DL][64]
    float *fb = lds + DL * kWave;                           // [DL][64]
    float *win = lds + 2 * DL * kWave;                      // [win_ring][64]
    const uint32_t wmask = P.win_ring - 1u;

    // ring positions are common to all channels: every channel pushes once per sample
    uint32_t dpos = (uint32_t)(counter0 % DL);
    uint32_t wpos = (uint32_t)(counter0 & wmask);           // slot the next sample is written to

    for (uint32_t i = 0; i < DL; ++i) {
        ff[i * kWave + lane] = S.dc_ff_ring[i * C + c];
        fb[i * kWave + lane] = S.dc_fb_ring[i * C + c];
    }
    for (uint32_t i = 0; i < P.win_ring; ++i) win[i * kWave + lane] = S.win_ring[i * C + c];

    Lane L;
    lane_load(L, S, c);
    GlobalCtx X{S, c, C};

    for (uint32_t t0 = 0; t0 < n_samples; t0 += B) {
        const uint32_t nb = min((uint32_t)B, n_samples - t0);
        float xs[B];
#pragma unroll
        for (int k = 0; k < B; ++k)
            xs[k] = (k < (int)nb) ? (float)x[(size_t)(t0 + k) * C + c] : 0.0f;

        float ys[B];
        int fire_k = -1;
        float fire_rem = 0.0f, fire_gain = 0.0f;
        const float unlocked0 = (L.flags & F_AGC_LOCKED) ? 0.0f : 1.0f;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (k < (int)nb) {
                // DCBlocker::filter rx/dcblock.rs:45-49, MovingAverage::filter :104-108
                const uint32_t dnext = (dpos + 1u == DL) ? 0u : dpos + 1u;
                float aged0 = ff[dpos * kWave + lane];
                ff[dpos * kWave + lane] = xs[k];
                float d0 = xs[k] - aged0;
                L.sum0 += d0;
                float ma0 = L.sum0 * P.dc_inv_len;
                float sig = ff[dnext * kWave + lane];       // window.front() after the push
                float aged1 = fb[dpos * kWave + lane];
                fb[dpos * kWave + lane] = ma0;
                float d1 = ma0 - aged1;
                L.sum1 += d1;
                float ma1 = L.sum1 * P.dc_inv_len;
                float km = P.dc_k * ma1;
                float y = sig - km;
                ys[k] = y;
                dpos = dnext;
                // Agc::input rx/agc.rs:72-77
                float out = y * L.gain;
                float e = 1.0f - fabsf(out);
                float ke = unlocked0 * e;
                float upd = ke * P.agc_bw;
                L.gain += upd;
                L.gain = rs_clamp(L.gain, P.agc_min, P.agc_max);
                // demod.push_scalar receiver.rs:346
                win[((wpos + k) & wmask) * kWave + lane] = out;
                // sample clock receiver.rs:347-355
                L.ted_clock += 1;
                float rem = L.until_next_ted - (float)L.ted_clock;
                bool fire = (fire_k < 0) && (rem <= 0.0f || fabsf(rem) < 0.5f);
                if (fire) { fire_k = k; fire_rem = rem; fire_gain = L.gain; L.ted_clock = 0; }
            }
        }

        if (fire_k >= 0) {
            const uint32_t newest = (wpos + (uint32_t)fire_k) & wmask;
            float sa_low = demod_now(P, taps, win, newest, lane);
            const uint32_t locked_before = L.flags & F_AGC_LOCKED;
            ted_instant(P, L, S, O, X, c, sa_low, fire_rem, counter0 + t0 + (uint32_t)fire_k + 1u);
            if ((L.flags & F_AGC_LOCKED) != locked_before) {
                // replay the AGC over the samples after the TED instant with the new lock
                const float unl = (L.flags & F_AGC_LOCKED) ? 0.0f : 1.0f;
                float g = fire_gain;
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    if (k > fire_k && k < (int)nb) {
                        float out = ys[k] * g;
                        float e = 1.0f - fabsf(out);
                        float ke = unl * e;
                        float upd = ke * P.agc_bw;
                        g += upd;
                        g = rs_clamp(g, P.agc_min, P.agc_max);
                        win[((wpos + k) & wmask) * kWave + lane] = out;
                    }
                }
                L.gain = g;
            }
        }
        wpos = (wpos + nb) & wmask;
    }

    lane_store(L, S, c);
    for (uint32_t i = 0; i < DL; ++i) {
        S.dc_ff_ring[i * C + c] = ff[i * kWave + lane];
        S.dc_fb_ring[i * C + c] = fb[i * kWave + lane];
    }
    for (uint32_t i = 0; i < P.win_ring; ++i) S.win_ring[i * C + c] = win[i * kWave + lane];
}

// ---------------------------------------------------------------------------------
// layout adaptor: channel-major x[c][t] -> time-major y[t][c], 64x64 tiles through LDS
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_to_time_major(const T *__restrict__ in,
                                                               T *__restrict__ out,
                                                               uint32_t n_channels, uint32_t n_samples)
{
    __shared__ T tile[64][65];
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;   // 64 x 4
    const uint32_t t0 = blockIdx.x * 64u, c0 = blockIdx.y * 64u;
    for (uint32_t r = ty; r < 64u; r += 4u) {
        uint32_t cc = c0 + r, tt = t0 + tx;
        if (cc < n_channels && tt < n_samples) tile[r][tx] = in[(size_t)cc * n_samples + tt];
    }
    __syncthreads();
    for (uint32_t r = ty; r < 64u; r += 4u) {
        uint32_t tt = t0 + r, cc = c0 + tx;
        if (cc < n_channels && tt < n_samples) out[(size_t)tt * n_channels + cc] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------------
// state initialisation: SameReceiver::from(&builder) receiver.rs:539-558 and reset() :182-198
// ---------------------------------------------------------------------------------
__global__ void init_state_kernel(Params P, State S, int is_reset, uint32_t first_col)
{
    const uint32_t c = first_col + blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t C = P.n_channels;
    if (c >= C) return;
    for (uint32_t i = 0; i < P.dc_len; ++i) { S.dc_ff_ring[i * C + c] = 0.0f; S.dc_fb_ring[i * C + c] = 0.0f; }
    S.dc_sum0[c] = 0.0f; S.dc_sum1[c] = 0.0f;
    // Agc::new starts at min(1, min_gain) (rx/agc.rs:55) but reset() sets 1.0 (:61)
    S.agc_gain[c] = is_reset ? 1.0f : P.agc_gain0;
    for (uint32_t i = 0; i < P.win_ring; ++i) S.win_ring[i * C + c] = 0.0f;
    S.ted_clock[c] = 0; S.until_next_ted[c] = P.samples_per_ted;
    S.ted_h0[c] = 0.0f; S.ted_h1[c] = 0.0f; S.ted_h2[c] = 0.0f;
    S.period_avg[c] = P.samples_per_ted; S.period_inst[c] = P.samples_per_ted;
    S.sq_data[c] = 0; S.sq_power[c] = 0.0f; S.sq_phist[c] = 0; S.sq_fill[c] = 0;
    S.sq_clock[c] = -1; S.sq_symbols[c] = 0;
    for (uint32_t i = 0; i < (uint32_t)kSquelchHist; ++i) S.sq_hist[i * C + c] = 0.0f;
    for (uint32_t i = 0; i < P.eq_nff; ++i) { S.eq_ffc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_ffw[i * C + c] = 0.0f; S.eq_snap_ffc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_snap_ffw[i * C + c] = 0.0f; }
    for (uint32_t i = 0; i < P.eq_nfb; ++i) { S.eq_fbc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_fbw[i * C + c] = 0.0f; S.eq_snap_fbc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_snap_fbw[i * C + c] = 0.0f; }
    // Equalizer::reset() preserves its mode, training word and count (rx/equalize.rs:191-196)
    if (!is_reset) { S.eq_word[c] = 0; S.eq_count[c] = 0; }
    S.fr_word[c] = 0; S.fr_count[c] = 0; S.fr_invalid[c] = 0; S.fr_len[c] = 0;
    // the timing loop keeps the bandwidth it had (receiver.rs:186 only calls symsync.reset());
    // the framer goes idle and the reported link state returns to NoCarrier.
    uint32_t fl = is_reset ? (S.flags[c] & (F_EQ_MODE_MASK | F_BW_LOCKED)) : (1u << F_EQ_MODE_SHIFT);
    S.flags[c] = fl;
    S.tk_next[c] = kNoDeadline; S.tk_last[c] = kNoDeadline; S.tk_n[c] = 0; S.wake_sample[c] = 0; S.wake_fired[c] = 0;
    for (uint32_t i = 0; i < (uint32_t)kTickRing; ++i) S.tk_ring[i * C + c] = kNoDeadline;
    if (P.trace_cap) S.trace_n[c] = 0;
}

// ---------------------------------------------------------------------------------
// cursor housekeeping as kernels, so a launch is compute-queue work only (a memset or a
// device-to-host copy in the stream drags the SDMA engine and host-resolved dependencies
// between consecutive launches into the critical path)
// ---------------------------------------------------------------------------------
__global__ void counters_kernel(uint32_t *dev, volatile uint32_t *host, int publish)
{
    if (threadIdx.x < 3) {
        if (publish) host[threadIdx.x] = dev[threadIdx.x];
        else dev[threadIdx.x] = 0;
    }
}
hipError_t launch_counters(uint32_t *dev, uint32_t *host_mapped, int publish, hipStream_t stream)
{
    hipLaunchKernelGGL(counters_kernel, dim3(1), dim3(64), 0, stream, dev, host_mapped, publish);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// The event log ordered by state column on the device (the host did this with a serial counting sort: 0.35 ms of a
// 2.2 ms harvest at configs[1]): events per column, an exclusive scan, and the log indices scattered into their
// column's range.  The scatter's order inside a range is that of its atomics; the host sorts each range (a handful of
// records) back into log order, which is the column's time order.
// ---------------------------------------------------------------------------------
__global__ void ev_hist_kernel(const DevEvent *__restrict__ ev, const uint32_t *__restrict__ counters, uint32_t cap, uint32_t n_bins,
                               uint32_t *__restrict__ cnt)
{
    const uint32_t n = min(counters[0], cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (ev[i].kind != kDevEventNone) atomicAdd(&cnt[min(ev[i].channel, n_bins - 1u)], 1u);
}
// The exclusive scan, two launches of one workgroup per 1 024 columns: the scan inside each workgroup (coalesced: a thread per
// column) with the workgroups' totals set aside, then every workgroup adds the totals before it.  (One workgroup walking
// contiguous stretches per thread -- 48 dependent, uncoalesced loads and as many stores each at 49 152 columns -- took
// 111 us between two demodulation launches.)
constexpr uint32_t kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void ev_scan_local_kernel(uint32_t n_bins, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ first,
                                                                      uint32_t *__restrict__ tot)
{
    __shared__ uint32_t wave_sum[kScanThreads / kWave];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1u), wave = tid / kWave, c = blockIdx.x * kScanThreads + tid;
    const uint32_t mine = c < n_bins ? cnt[c] : 0u;
    uint32_t incl = mine;
    for (int off = 1; off < (int)kWave; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += t; }
    if (lane == kWave - 1u) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t w = 0; w < wave; ++w) run += wave_sum[w];
    if (c < n_bins) first[c] = run;
    if (tid == kScanThreads - 1u) tot[blockIdx.x] = run + mine;
}
__global__ __launch_bounds__(kScanThreads) void ev_scan_base_kernel(uint32_t n_bins, uint32_t *__restrict__ cnt, uint32_t *__restrict__ first,
                                                                     const uint32_t *__restrict__ tot)
{
    // cnt becomes the scatter's running offsets
    uint32_t base = 0;
    for (uint32_t j = 0; j < blockIdx.x; ++j) base += tot[j];          // (wave-uniform: scalar loads)
    const uint32_t c = blockIdx.x * kScanThreads + threadIdx.x;
    if (c < n_bins) { const uint32_t v = first[c] + base; first[c] = v; cnt[c] = v; }
    if (blockIdx.x + 1u == gridDim.x && threadIdx.x == 0u) first[n_bins] = base + tot[blockIdx.x];
}
__global__ void ev_scatter_kernel(const DevEvent *__restrict__ ev, const uint32_t *__restrict__ counters, uint32_t cap, uint32_t n_bins,
                                  uint32_t *__restrict__ fill, DevEvent *__restrict__ sorted)
{
    // the records themselves move (the host then walks a column's events in one contiguous stretch instead of chasing
    // indices through the log); `channel`, which the range implies, carries the record's log index for the host's
    // fix-up of the order inside a range
    const uint32_t n = min(counters[0], cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        DevEvent e = ev[i];
        if (e.kind == kDevEventNone) continue;
        const uint32_t pos = atomicAdd(&fill[min(e.channel, n_bins - 1u)], 1u);
        e.channel = i;
        sorted[pos] = e;
    }
}
uint32_t event_sort_extra_words(uint32_t n_bins) { return (n_bins + kScanThreads - 1u) / kScanThreads; }
hipError_t launch_event_sort(const DevEvent *ev, const uint32_t *counters, uint32_t cap, uint32_t n_bins, uint32_t *cnt, uint32_t *first,
                             DevEvent *sorted, hipStream_t stream, bool cnt_is_zero)
{
    // (cnt_is_zero: the launch's prologue kernel emptied the bins -- a memset on the stream costs a launch and a gap of its own)
    if (!cnt_is_zero) {
        hipError_t e = hipMemsetAsync(cnt, 0, (size_t)n_bins * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(ev_hist_kernel, dim3(256), dim3(256), 0, stream, ev, counters, cap, n_bins, cnt);
    const uint32_t scan_grid = (n_bins + kScanThreads - 1u) / kScanThreads;
    uint32_t *tot = first + n_bins + 1u;                   // (first: n_bins + 1 + event_sort_extra_words(n_bins) words)
    hipLaunchKernelGGL(ev_scan_local_kernel, dim3(scan_grid), dim3(kScanThreads), 0, stream, n_bins, cnt, first, tot);
    hipLaunchKernelGGL(ev_scan_base_kernel, dim3(scan_grid), dim3(kScanThreads), 0, stream, n_bins, cnt, first, tot);
    hipLaunchKernelGGL(ev_scatter_kernel, dim3(256), dim3(256), 0, stream, ev, counters, cap, n_bins, cnt, sorted);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// launchers (called from same_batch.cpp)
// ---------------------------------------------------------------------------------
template <typename SampleT>
static hipError_t launch_demod_t(const Params &P, const State &S, const Output &O, const float4 *taps,
                                 const SampleT *x, uint32_t n_samples, uint64_t counter0,
                                 hipStream_t stream)
{
    const uint32_t grid = (P.n_channels + kWave - 1) / kWave;
    const size_t lds = (size_t)(2 * P.dc_len + P.win_ring) * kWave * sizeof(float);
#define SAME_LAUNCH(BV)                                                                          \
    hipLaunchKernelGGL((demod_kernel<BV, SampleT>), dim3(grid), dim3(kWave), lds, stream, P, S, \
                       O, taps, x, n_samples, counter0)
    switch (P.block_len) {
    case 16: SAME_LAUNCH(16); break;
    case 8: SAME_LAUNCH(8); break;
    case 4: SAME_LAUNCH(4); break;
    case 2: SAME_LAUNCH(2); break;
    default: SAME_LAUNCH(1); break;
    }
#undef SAME_LAUNCH
    return hipGetLastError();
}

hipError_t launch_demod(const Params &P, const State &S, const Output &O, const float4 *taps,
                        const float *x, uint32_t n_samples, uint64_t counter0, hipStream_t stream)
{ return launch_demod_t<float>(P, S, O, taps, x, n_samples, counter0, stream); }

hipError_t launch_demod_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                            const int16_t *x, uint32_t n_samples, uint64_t counter0, hipStream_t stream)
{ return launch_demod_t<int16_t>(P, S, O, taps, x, n_samples, counter0, stream); }

size_t demod_lds_bytes(const Params &P)
{ return (size_t)(2 * P.dc_len + P.win_ring) * kWave * sizeof(float); }

hipError_t launch_init_state(const Params &P, const State &S, int is_reset, hipStream_t stream, uint32_t first_col)
{
    if (first_col >= P.n_channels) return hipSuccess;
    const uint32_t grid = (P.n_channels - first_col + 255) / 256;
    hipLaunchKernelGGL(init_state_kernel, dim3(grid), dim3(256), 0, stream, P, S, is_reset, first_col);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// time-parallel chunks: state columns in and out of the wide state blob
// ---------------------------------------------------------------------------------
__global__ void copy_state_columns_kernel(const StateArrayDesc *desc, uint32_t Csrc, uint32_t Cdst,
                                          const uint32_t *src_col, uint32_t n_cols, uint32_t src_base)
{
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    const StateArrayDesc d = desc[blockIdx.y];
    const uint32_t sc = src_col ? src_col[col] : col + src_base;
    const uint32_t w = d.elem_words;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(d.src);
    uint32_t *dst = reinterpret_cast<uint32_t *>(d.dst);
    // (the words of a column spread over blockIdx.z: one thread walking the 64 rows of a window ring, or the 72 words of a
    // framer row, is that many dependent round trips)
    const uint32_t n = d.rows * w;
    for (uint32_t k = blockIdx.z; k < n; k += gridDim.z) {
        const uint32_t r = k / w, i = k - r * w;
        dst[((size_t)r * Cdst + col) * w + i] = src[((size_t)r * Csrc + sc) * w + i];
    }
}
hipError_t launch_copy_state_columns(const StateArrayDesc *desc, uint32_t n_desc, uint32_t src_channels,
                                     uint32_t dst_channels, const uint32_t *src_col, uint32_t n_cols, hipStream_t stream,
                                     uint32_t src_base)
{
    hipLaunchKernelGGL(copy_state_columns_kernel, dim3((n_cols + 255) / 256, n_desc, 16), dim3(256), 0, stream, desc,
                       src_channels, dst_channels, src_col, n_cols, src_base);
    return hipGetLastError();
}
__global__ void chunk_final_column_kernel(const uint64_t *handover, uint32_t C, ChunkGeom g, uint32_t *final_col)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    uint32_t cur = 0;
    for (;;) {
        const uint64_t h = handover[(size_t)cur * C + c];
        if (h == kNoHandover) break;
        const uint32_t nxt = g.owner_of(h);
        if (nxt <= cur) break;                   // (cannot happen: a hand-over lies in a later chunk's range)
        cur = nxt;
    }
    final_col[c] = cur * C + c;
}
hipError_t launch_chunk_final_column(const uint64_t *handover, uint32_t in_channels, ChunkGeom g, uint32_t *final_col,
                                     hipStream_t stream)
{
    hipLaunchKernelGGL(chunk_final_column_kernel, dim3((in_channels + 255) / 256), dim3(256), 0, stream, handover, in_channels, g, final_col);
    return hipGetLastError();
}
// the same for per-channel boundaries: the chunk that owns a hand-over instant is the last one whose own range
// (own_start[k][c], relative to the call's first sample) begins at or before it
__global__ void chunk_final_column_pc_kernel(const uint64_t *handover, const uint32_t *own_start, uint32_t C, uint32_t K,
                                             uint64_t counter0, uint32_t *final_col)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    uint32_t cur = 0;
    for (;;) {
        const uint64_t h = handover[(size_t)cur * C + c];
        if (h == kNoHandover) break;
        uint32_t nxt = cur;
        for (uint32_t k = cur + 1u; k < K; ++k)
            if (counter0 + own_start[(size_t)k * C + c] <= h) nxt = k;
        if (nxt <= cur) break;
        cur = nxt;
    }
    final_col[c] = cur * C + c;
}
hipError_t launch_chunk_final_column_pc(const uint64_t *handover, const uint32_t *own_start, uint32_t in_channels, uint32_t n_chunks,
                                        uint64_t counter0, uint32_t *final_col, hipStream_t stream)
{
    hipLaunchKernelGGL(chunk_final_column_pc_kernel, dim3((in_channels + 255) / 256), dim3(256), 0, stream, handover, own_start,
                       in_channels, n_chunks, counter0, final_col);
    return hipGetLastError();
}
// ---- per-channel chunk boundaries (channel-major input) ----------------------------------------------------------
#ifndef SAME_SCOUT_BLOCK
#define SAME_SCOUT_BLOCK 256
#endif
constexpr uint32_t kScoutBlock = SAME_SCOUT_BLOCK;        // samples per energy reading (one 64-byte sector of them is read)
uint32_t tp_scout_block() { return kScoutBlock; }
__global__ void tp_scout_kernel(const float *__restrict__ x, TpPlan g, float *__restrict__ energy)
{
    // four lanes per reading, 16 bytes each: a wavefront's load covers 16 sectors of 64 bytes (one lane per reading with
    // four loads touched 64 sectors per instruction: 115 us for 4 096 channels x 10 s against 105 -- one sector per KB of
    // input is what it costs, ~2 TB/s of 64-byte reads)
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t >> 2;
    const uint32_t q = (uint32_t)t & 3u;
    const bool live = i < (size_t)g.channels * g.scout_blocks;
    float e = 0.0f;
    if (live) {
        const uint32_t c = (uint32_t)(i / g.scout_blocks), j = (uint32_t)(i % g.scout_blocks);
        const float4 v = reinterpret_cast<const float4 *>(x + (size_t)c * g.in_samples + (size_t)j * kScoutBlock)[q];
        e = (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
    }
    e += __shfl_xor(e, 1);
    e += __shfl_xor(e, 2);
    if (live && q == 0u) energy[i] = e;
}
// A boundary may sit at a scout block j when the channel has been quiet from kQuietBefore
// blocks before it (the carrier stopped >= 2 048 samples ago: the link layer is back to NoCarrier -- 31 symbols of
// power history, 1 312 samples, plus the framer's end -- or the chunk simply runs on until it is) to one block after it.
// "Quiet": below 8 % of the channel's loudest reading.  Among all ways to cut the call at such instants into at most
// n_chunks pieces the planner takes one that minimises the LONGEST piece (a workgroup runs as long as its longest
// lane): bisection on that length, each trial a greedy scan that always cuts at the latest allowed instant.  Chunks
// that are left over own nothing (they warm up and hand over at once).  Where a stretch has no quiet instant at all
// the cut falls where the length limit puts it and the chunk before it runs on until idle, as with uniform boundaries.
__global__ __launch_bounds__(kWave) void tp_boundaries_kernel(const float *__restrict__ energy, TpPlan g, uint32_t *__restrict__ own_start,
                                                               uint32_t *__restrict__ row0, uint32_t *__restrict__ nominal)
{
    // one wavefront per channel: the readings, the "latest quiet instant" table and the cuts live in LDS; the search for
    // the shortest feasible limit runs 64 candidates at a time, one per lane
    extern __shared__ float tp_lds[];
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    const int NB = (int)g.scout_blocks, K = (int)g.n_chunks;
    float *e = tp_lds;                                             // [NB]
    int *lastq = reinterpret_cast<int *>(tp_lds + NB);             // [NB] latest allowed instant at or before block j (-1: none)
    int *cut = lastq + NB;                                         // [64]
    constexpr int kQuietBefore = 2048 / (int)kScoutBlock, kQuietAfter = 1;
    float m = 0.0f;
    for (int j = (int)lane; j < NB; j += (int)kWave) { const float v = energy[(size_t)c * NB + j]; e[j] = v; m = fmaxf(m, v); }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const float thr = 0.08f * m;
    __syncthreads();
    // allowed instants, and the running "latest allowed": every lane a contiguous stretch, the stretches' last values
    // scanned across the wavefront
    const int S = (NB + (int)kWave - 1) / (int)kWave, j0 = (int)lane * S, j1 = min(j0 + S, NB);
    int run = -1;
    for (int j = j0; j < j1; ++j) {
        bool ok = j - kQuietBefore >= 0 && j + kQuietAfter < NB;
        if (ok) for (int t = j - kQuietBefore; t <= j + kQuietAfter; ++t) if (e[t] > thr) { ok = false; break; }
        if (ok) run = j;
        lastq[j] = run;
    }
    int before = run;                                              // inclusive scan (max) of the stretches' last values ...
    for (int off = 1; off < (int)kWave; off <<= 1) { const int o = __shfl_up(before, off); if ((int)lane >= off) before = max(before, o); }
    before = __shfl_up(before, 1);                                 // ... made exclusive
    if (lane == 0) before = -1;
    for (int j = j0; j < j1; ++j) if (lastq[j] < 0) lastq[j] = before;
    __syncthreads();
    const uint32_t kB = g.block_len;
    // shortest own range, in scout blocks.  Small on purpose: with a large minimum the greedy cut below is no longer
    // optimal (a cut just before a burst may be unreachable from the latest allowed cut before it, while the burst's
    // end is out of reach from there), and the search over it no longer monotone
    const int min_blocks = 2;
    // greedy cut for a given limit L (scout blocks per chunk): always at the latest allowed instant in reach; returns
    // the number of pieces, cuts in cut[1..] when `store`.  forced: where no allowed instant is in reach, cut at the
    // limit -- the chunk before it then runs on until idle; only used when no plan without such cuts exists
    auto plan = [&](int L, bool store, bool forced) __attribute__((always_inline)) {
        int pos = 0, n = 1;
        while (NB - pos > L) {
            int q = lastq[min(pos + L, NB - 1)];
            if (q < pos + min_blocks) { if (!forced) return 1 << 20; q = pos + L; }
            if (q > NB - min_blocks) q = NB - min_blocks;
            if (q <= pos) return 1 << 20;
            if (n < 64 && store) cut[n] = q;
            pos = q; ++n;
            if (n > K) return n;
        }
        return n;
    };
    const bool forced = plan(NB - 1, false, false) > K;            // not even the loosest limit works with clean cuts only
    // the smallest limit in [lo, NB] whose plan has at most K pieces (feasibility is monotone in the limit): 64 evenly
    // spaced candidates, then the stretch below the first feasible one, until the stretch is a single value
    int lo = (NB + K - 1) / K, hi = NB;
    while (lo < hi) {
        const int step = (hi - lo + (int)kWave - 1) / (int)kWave;            // candidates lo, lo + step, ... (at or beyond hi: feasible)
        const int cand = lo + step * (int)lane;
        const bool ok = cand >= hi || plan(cand, false, forced) <= K;
        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
        if (okm == 0ull) { lo += step * ((int)kWave - 1) + 1; continue; }    // every candidate below hi fails
        const int first = __builtin_ctzll(okm);
        hi = min(lo + step * first, hi);
        if (first > 0) lo += step * (first - 1) + 1;
    }
    int used = 0;
    if (lane == 0) used = plan(lo, true, forced);
    used = __shfl(used, 0);
    __syncthreads();
    // `used` pieces: cuts 1 .. used-1.  Chunks that are left over sit, empty, at the last cut: the piece after it is
    // the last chunk (K-1), which ends with the input and leaves the channel's state.
    auto own = [&](int k) __attribute__((always_inline)) -> uint32_t {
        const int i = k < used ? k : used - 1;
        uint32_t p = (k >= 1 && i >= 1) ? (uint32_t)cut[i] * kScoutBlock : 0u;
        return p - p % kB;
    };
    for (int k = (int)lane; k < K; k += (int)kWave) {
        const uint32_t p = own(k);
        const uint32_t r = (k == 0 || p < g.warmup_samples) ? 0u : p - g.warmup_samples;     // multiples of the block length
        const size_t v = (size_t)k * g.channels + c;
        own_start[v] = p;
        row0[v] = r;
        nominal[v] = k + 1 < K ? (own(k + 1) - r + kB - 1u) / kB : 0xffffffffu;
    }
}
// Pieces of similar length share a workgroup (a workgroup runs as long as its longest lane, and with more workgroups
// than the machine holds at once the long ones should start first): a bucket sort of the state columns by length,
// descending, in three groups that keep workgroups of their own -- last chunk (grid positions 0 .. C-1: they end with
// the input and store the channels' state), chunk 0 (they load it), everything in between.
constexpr uint32_t kSortBuckets = 4096, kSortBucketBlocks = 4;
__device__ __forceinline__ void tp_sort_key(const TpPlan &g, const uint32_t *row0, const uint32_t *nominal, uint32_t v,
                                            uint32_t *group, uint32_t *bucket)
{
    const uint32_t chunk = v / g.channels;
    const uint32_t avail = (g.whole_samples - row0[v]) / g.block_len;
    *group = chunk + 1u == g.n_chunks ? 0u : (chunk == 0u ? 1u : 2u);
    const uint32_t len = *group == 0u ? avail : min(nominal[v], avail);
    *bucket = kSortBuckets - 1u - min(len / kSortBucketBlocks, kSortBuckets - 1u);         // longest first
}
// One workgroup sorts them all: histogram, scan and scatter in LDS (three launches with global atomics -- every empty
// piece on the same counter -- took 160 us for 40 960 columns; this takes 48).
constexpr uint32_t kSortThreads = 1024;
__global__ __launch_bounds__(kSortThreads) void tp_sort_kernel(TpPlan g, const uint32_t *__restrict__ row0, const uint32_t *__restrict__ nominal,
                                                                uint32_t *__restrict__ perm)
{
    __shared__ uint32_t hist[3u * kSortBuckets];
    __shared__ uint32_t wave_sum[kSortThreads / kWave];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1u), wave = tid / kWave, n = g.n_chunks * g.channels;
    for (uint32_t i = tid; i < 3u * kSortBuckets; i += kSortThreads) hist[i] = 0u;
    __syncthreads();
    // (the last chunks, group 0, keep their channel order: tp_align gives the lanes of such a workgroup one common first
    // row, so what a last-chunk column computes depends on who its neighbours are -- they must not depend on the order
    // in which atomics land)
    for (uint32_t v = tid; v < n; v += kSortThreads) {
        uint32_t grp, b;
        tp_sort_key(g, row0, nominal, v, &grp, &b);
        if (grp != 0u) atomicAdd(&hist[grp * kSortBuckets + b], 1u);
    }
    __syncthreads();
    // exclusive prefix sums per group, each offset by where its group starts in the grid: a thread owns four buckets
    constexpr uint32_t PER = kSortBuckets / kSortThreads;
    static_assert(PER * kSortThreads == kSortBuckets, "buckets per thread");
    const uint32_t base[3] = {0u, g.channels, 2u * g.channels};
    for (uint32_t grp = 0; grp < 3u; ++grp) {
        uint32_t *h = hist + grp * kSortBuckets + tid * PER;
        uint32_t mine = 0;
        for (uint32_t i = 0; i < PER; ++i) mine += h[i];
        uint32_t incl = mine;
        for (int off = 1; off < (int)kWave; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += t; }
        if (lane == kWave - 1u) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = base[grp] + incl - mine;
        for (uint32_t w = 0; w < wave; ++w) before += wave_sum[w];
        for (uint32_t i = 0; i < PER; ++i) { const uint32_t c = h[i]; h[i] = before; before += c; }
        __syncthreads();
    }
    for (uint32_t v = tid; v < n; v += kSortThreads) {
        uint32_t grp, b;
        tp_sort_key(g, row0, nominal, v, &grp, &b);
        if (grp == 0u) perm[v - (g.n_chunks - 1u) * g.channels] = v;
        else perm[atomicAdd(&hist[grp * kSortBuckets + b], 1u)] = v;
    }
}
__global__ void tp_iota_kernel(uint32_t *perm, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = i;
}
// One wavefront per workgroup of the demodulation launch (64 grid positions): how many blocks it runs at most, and --
// last chunk -- one common first row for its lanes, so that they all end with the input.
__global__ void tp_align_kernel(TpPlan g, const uint32_t *__restrict__ perm, uint32_t *__restrict__ row0, const uint32_t *__restrict__ nominal,
                                uint32_t *__restrict__ wg_blocks, uint32_t *__restrict__ wg_len)
{
    const uint32_t v = perm[blockIdx.x * kWave + threadIdx.x];
    const uint32_t chunk = v / g.channels;
    uint32_t avail = (g.whole_samples - row0[v]) / g.block_len;
    uint32_t m = avail;
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if (chunk + 1u == g.n_chunks) row0[v] = g.whole_samples - m * g.block_len;
    if (threadIdx.x == 0) wg_blocks[blockIdx.x] = m;
    if (wg_len) {
        // how long the workgroup will run if every lane hands over on time: its longest lane's warm-up + own range
        uint32_t len = chunk + 1u == g.n_chunks ? m : min(nominal[v], avail);
        for (int off = 32; off > 0; off >>= 1) len = max(len, (uint32_t)__shfl_xor((int)len, off));
        if (threadIdx.x == 0) wg_len[blockIdx.x] = len;
    }
}
// Longest workgroups first, whatever group they are of: with more workgroups than the machine holds at once the ones that
// wait must be the short ones, and the long ones -- which set the launch's length -- should start at time zero.  One
// wavefront per workgroup of the demodulation launch: its rank among all of them by length (ties by position), then
// its 64 grid positions and its block count move to that rank.
// `pairs` (the symbol-paced pipeline, whose workgroup is TWO groups of 64 columns with a CU to themselves): the group of rank r
// and the group of rank n - 1 - r share a workgroup -- the longest with the shortest -- so that every long group runs the second
// part of its launch alone on its CU, where a step is ~12 % shorter.
__global__ void tp_wg_order_kernel(uint32_t n_wg, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ wg_blocks,
                                   const uint32_t *__restrict__ wg_len, uint32_t *__restrict__ perm_out, uint32_t *__restrict__ wg_blocks_out,
                                   uint32_t pairs)
{
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    const uint32_t mine = wg_len[w];
    uint32_t rank = 0;
    for (uint32_t j = lane; j < n_wg; j += kWave) { const uint32_t b = wg_len[j]; rank += (b > mine || (b == mine && j < w)) ? 1u : 0u; }
    for (int off = 32; off > 0; off >>= 1) rank += (uint32_t)__shfl_xor((int)rank, off);
    const uint32_t pos = (pairs && (n_wg & 1u) == 0u) ? (rank < n_wg / 2u ? 2u * rank : 2u * (n_wg - 1u - rank) + 1u) : rank;
    perm_out[pos * kWave + lane] = perm[w * kWave + lane];
    if (lane == 0) wg_blocks_out[pos] = wg_blocks[w];
}
hipError_t launch_tp_plan(const float *x, const TpPlan &g, float *energy, uint32_t *own_start, uint32_t *row0,
                          uint32_t *nominal, uint32_t *perm, uint32_t *wg_blocks, bool sorted, hipStream_t stream,
                          uint32_t *perm_out, uint32_t *wg_blocks_out, bool pairs)
{
    const size_t n = (size_t)g.channels * g.scout_blocks;
    const uint32_t columns = g.n_chunks * g.channels;
    hipLaunchKernelGGL(tp_scout_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, stream, x, g, energy);
    hipLaunchKernelGGL(tp_boundaries_kernel, dim3(g.channels), dim3(kWave), (size_t)g.scout_blocks * 8 + 64 * sizeof(int), stream, energy, g, own_start, row0, nominal);
    if (sorted) {
        hipLaunchKernelGGL(tp_sort_kernel, dim3(1), dim3(kSortThreads), 0, stream, g, row0, nominal, perm);
    } else {
        hipLaunchKernelGGL(tp_iota_kernel, dim3((columns + 255) / 256), dim3(256), 0, stream, perm, columns);
    }
    // (the workgroups' lengths sit behind the reordered block counts: wg_blocks_out[n_wg ..])
    uint32_t *wg_len = perm_out ? wg_blocks_out + columns / kWave : nullptr;
    hipLaunchKernelGGL(tp_align_kernel, dim3(columns / kWave), dim3(kWave), 0, stream, g, perm, row0, nominal, wg_blocks, wg_len);
    if (perm_out)
        hipLaunchKernelGGL(tp_wg_order_kernel, dim3(columns / kWave), dim3(kWave), 0, stream, columns / kWave, perm, wg_blocks, wg_len,
                           perm_out, wg_blocks_out, pairs ? 1u : 0u);
    return hipGetLastError();
}

// The start of a time-parallel launch in ONE kernel (round 6; it was init_state_kernel over the fresh columns -- a thread per
// column walking ~230 strided words, 70 us for 32 768 columns -- then fill_u64, a memset and the cursor reset: ~90 us with their
// gaps on the stream of a 1.75 ms demodulation launch).  Fresh receivers in every column: the state arrays are copied, 16 bytes a
// thread, from a template blob that init_state_kernel filled once when the wide state was laid out (the channels' own state goes
// over chunk 0's columns next: launch_copy_state_columns); no column has handed over; the event sort's bins are empty; the
// launch cursors are zero.
__global__ __launch_bounds__(256) void tp_prologue_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16,
                                                          uint64_t *__restrict__ handover, uint32_t *__restrict__ sort_cnt, uint32_t n_cols,
                                                          uint32_t *__restrict__ counters)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < n16; i += stride) dst[i] = src[i];
    for (size_t i = t; i < n_cols; i += stride) { handover[i] = kNoHandover; if (sort_cnt) sort_cnt[i] = 0u; }
    if (t < 3 && counters) counters[t] = 0u;
}
hipError_t launch_tp_prologue(void *blob, const void *fresh, size_t bytes, uint64_t *handover, uint32_t *sort_cnt, uint32_t n_cols,
                              uint32_t *counters, hipStream_t stream)
{
    const size_t n16 = bytes / 16;
    const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(tp_prologue_kernel, dim3(grid ? grid : 1u), dim3(256), 0, stream, reinterpret_cast<uint4 *>(blob),
                       reinterpret_cast<const uint4 *>(fresh), n16, handover, sort_cnt, n_cols, counters);
    return hipGetLastError();
}
// int16 samples as the f32 the kernels make of them (unscaled: crates/samedec/src/app.rs:112) -- SAME_BATCH_CALL_INVARIANT's waiting buffer
__global__ void cast_i16_f32_kernel(const int16_t *__restrict__ in, float *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
hipError_t launch_cast_i16_f32(const int16_t *in, float *out, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(cast_i16_f32_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}
__global__ void fill_u64_kernel(uint64_t *p, size_t n, uint64_t v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t launch_fill_u64(uint64_t *p, size_t n, uint64_t v, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n, v);
    return hipGetLastError();
}

hipError_t launch_transpose_f32(const float *in, float *out, uint32_t n_channels, uint32_t n_samples,
                                hipStream_t stream)
{
    dim3 grid((n_samples + 63) / 64, (n_channels + 63) / 64);
    hipLaunchKernelGGL(transpose_to_time_major<float>, grid, dim3(256), 0, stream, in, out, n_channels, n_samples);
    return hipGetLastError();
}
hipError_t launch_transpose_i16(const int16_t *in, int16_t *out, uint32_t n_channels, uint32_t n_samples,
                                hipStream_t stream)
{
    dim3 grid((n_samples + 63) / 64, (n_channels + 63) / 64);
    hipLaunchKernelGGL(transpose_to_time_major<int16_t>, grid, dim3(256), 0, stream, in, out, n_channels, n_samples);
    return hipGetLastError();
}

}  // namespace same
