"""Python mirror of sameold's `SameReceiverBuilder` / `SameReceiver` over the MI355X C ABI.

The names and argument meaning follow the reference (crates/sameold/src/receiver/
builder.rs:22-356, receiver.rs:92-224) so tests read like the reference's own:

    rx = SameReceiverBuilder(22050).with_timing_max_deviation(0.01).build()
    for evt in rx.iter_events(samples): ...

`build_batch(n)` returns the batched receiver (n independent SameReceivers in one GPU).
Everything here goes through include/same_rx.h; there is no CPU fallback -- if the HIP
library is missing or no gfx950 device is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Iterator, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SAME_LIB_VARIANT: a measurement build made with SAME_BUILD_VARIANT, tools/ only -- see sameold_amd/build.py)
LIB_PATH = os.path.join(_HERE, f"libsame_rx.{os.environ['SAME_LIB_VARIANT']}.so" if os.environ.get("SAME_LIB_VARIANT") else "libsame_rx.so")

LINK_NO_CARRIER, LINK_SEARCHING, LINK_READING, LINK_BURST = 0, 1, 2, 3
TRANSPORT_IDLE, TRANSPORT_ASSEMBLING, TRANSPORT_MSG_START, TRANSPORT_MSG_END, TRANSPORT_MSG_ERR = 16, 17, 18, 19, 20
LAYOUT_TIME_MAJOR, LAYOUT_CHANNEL_MAJOR = 0, 1
BATCH_LINK_ONLY, BATCH_TRACE_SYMBOLS, BATCH_GENERIC_KERNEL, BATCH_TIME_PARALLEL, BATCH_RELAXED, BATCH_CALL_INVARIANT = 1, 2, 4, 8, 16, 32
TP_EVENT_TOLERANCE_SYMBOLS = 2      # SAME_TP_EVENT_TOLERANCE_SYMBOLS
STREAM_OWN = (1 << 64) - 1          # SAME_STREAM_OWN: (void *)-1, the library's own stream
EVENT_MAX_BYTES = 288

ERRORS = {-1: "EINVAL", -2: "EDCLEN", -3: "EAGCLIMITS", -4: "EEQORDER", -5: "ENODEVICE",
          -6: "EHIP", -7: "EOVERFLOW", -8: "ENOMEM", -9: "ERATE", -10: "EKERNEL"}

KIND_NAMES = {0: "no_carrier", 1: "searching", 2: "reading", 3: "burst", 16: "idle",
              17: "assembling", 18: "message_start", 19: "message_end", 20: "message_err"}


class SameError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"{ERRORS.get(code, code)}: {text}")
        self.code = code


class Event(C.Structure):
    """`SameReceiverEvent` (receiver/output.rs:24-27)."""
    _fields_ = [
        ("kind", C.c_uint32), ("channel", C.c_uint32),
        ("sample_counter", C.c_uint64), ("symbol_count", C.c_uint64),
        ("len", C.c_uint32), ("aux", C.c_uint32), ("aux2", C.c_uint32), ("reserved", C.c_uint32),
        ("bytes", C.c_uint8 * EVENT_MAX_BYTES),
    ]

    def data(self) -> bytes:
        return bytes(self.bytes[: min(self.len, EVENT_MAX_BYTES)])

    def input_sample_counter(self) -> int:
        return int(self.sample_counter)

    def burst(self) -> Optional[bytes]:
        return self.data() if self.kind == LINK_BURST else None

    def message(self) -> Optional[str]:
        """`into_message_ok()`: header text, or "NNNN" for EndOfMessage."""
        if self.kind == TRANSPORT_MSG_START:
            return self.data().decode("ascii")
        if self.kind == TRANSPORT_MSG_END:
            return "NNNN"
        return None

    def as_tuple(self):
        return (int(self.kind), int(self.sample_counter), self.data())

    def __repr__(self):
        return f"Event(ch{self.channel} {KIND_NAMES.get(self.kind, self.kind)}@{self.sample_counter}, {self.data()!r})"


EVENT_DTYPE = np.dtype([("kind", "<u4"), ("channel", "<u4"), ("sample_counter", "<u8"), ("symbol_count", "<u8"),
                        ("len", "<u4"), ("aux", "<u4"), ("aux2", "<u4"), ("reserved", "<u4"),
                        ("bytes", "u1", (EVENT_MAX_BYTES,))])
assert EVENT_DTYPE.itemsize == C.sizeof(Event)


class SymbolTrace(C.Structure):
    _fields_ = [("sample_counter", C.c_uint64), ("zero", C.c_float), ("sym", C.c_float),
                ("err", C.c_float), ("samples_until_next_ted", C.c_float)]


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels carry their own libamdhip64 (same SONAME as the system one, different
    file).  If this library is loaded before torch, the process ends up with two HIP runtimes
    and the one initialised second sees no device.  Loading torch's copy first -- without
    importing torch -- makes both bind to the same runtime whichever order they arrive in."""
    import importlib.util
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library() -> C.CDLL:
    """Load libsame_rx.so and declare every prototype of include/same_rx.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SameError(-5, f"{LIB_PATH} is missing: run `python -m sameold_amd.build` "
                            "(hipcc, gfx950); there is no CPU implementation to fall back to")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    P, vp, u32, u64, f32 = C.POINTER, C.c_void_p, C.c_uint32, C.c_uint64, C.c_float

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("same_last_error", C.c_char_p)
    sig("same_rx_abi_version", u32)
    sig("same_rx_builder_new", vp, u32)
    sig("same_rx_builder_default", vp)
    sig("same_rx_builder_clone", vp, vp)
    sig("same_rx_builder_free", None, vp)
    sig("same_rx_builder_with_dc_blocker_length", None, vp, f32)
    sig("same_rx_builder_with_agc_bandwidth", None, vp, f32)
    sig("same_rx_builder_with_agc_gain_limits", None, vp, f32, f32)
    sig("same_rx_builder_with_timing_bandwidth", None, vp, f32, f32)
    sig("same_rx_builder_with_timing_max_deviation", None, vp, f32)
    sig("same_rx_builder_with_squelch_power", None, vp, f32, f32)
    sig("same_rx_builder_with_squelch_bandwidth", None, vp, f32)
    sig("same_rx_builder_with_preamble_max_errors", None, vp, u32)
    sig("same_rx_builder_with_adaptive_equalizer", None, vp, u32, u32, f32, f32)
    sig("same_rx_builder_without_adaptive_equalizer", None, vp)
    sig("same_rx_builder_with_frame_prefix_max_errors", None, vp, u32)
    sig("same_rx_builder_with_frame_max_invalid", None, vp, u32)
    sig("same_rx_builder_input_rate", u32, vp)
    sig("same_rx_builder_dc_blocker_length", f32, vp)
    sig("same_rx_builder_agc_bandwidth", f32, vp)
    sig("same_rx_builder_agc_gain_limits", None, vp, P(f32))
    sig("same_rx_builder_timing_bandwidth", None, vp, P(f32), P(f32))
    sig("same_rx_builder_timing_max_deviation", f32, vp)
    sig("same_rx_builder_squelch_power", None, vp, P(f32), P(f32))
    sig("same_rx_builder_squelch_bandwidth", f32, vp)
    sig("same_rx_builder_preamble_max_errors", u32, vp)
    sig("same_rx_builder_adaptive_equalizer", C.c_int, vp, P(u32), P(u32), P(f32), P(f32))
    sig("same_rx_builder_frame_prefix_max_errors", u32, vp)
    sig("same_rx_builder_frame_max_invalid", u32, vp)

    sig("same_batch_new", C.c_int, vp, u32, C.c_int, u32, P(vp))
    sig("same_batch_free", None, vp)
    sig("same_batch_reset", C.c_int, vp)
    sig("same_batch_input_rate", u32, vp)
    sig("same_batch_n_channels", u32, vp)
    sig("same_batch_input_sample_counter", u64, vp)
    sig("same_batch_device", C.c_int, vp)
    sig("same_batch_process_device", C.c_int, vp, vp, C.c_size_t, u32, vp)
    sig("same_batch_order_after", C.c_int, vp, vp)
    sig("same_batch_set_call_window", C.c_int, vp, C.c_uint32)
    sig("same_batch_time_parallel_config", C.c_int, vp, u32, u32, u32)
    sig("same_batch_time_parallel_chunks", u32, vp)
    sig("same_batch_time_parallel_per_channel", C.c_int, vp)
    sig("same_rx_source_hash", C.c_char_p)
    sig("same_batch_process_device_i16", C.c_int, vp, vp, C.c_size_t, u32, vp)
    sig("same_batch_process_host", C.c_int, vp, vp, C.c_size_t, u32)
    sig("same_batch_process_host_i16", C.c_int, vp, vp, C.c_size_t, u32)
    sig("same_batch_flush", C.c_int, vp)
    sig("same_batch_sync", C.c_int, vp)
    sig("same_batch_poll_events", C.c_int, vp, P(Event), C.c_size_t, P(C.c_size_t), P(C.c_size_t))
    sig("same_batch_pending_events", C.c_size_t, vp)
    sig("same_batch_peek_events", C.c_int, vp, P(P(Event)), P(C.c_size_t))
    sig("same_batch_drop_events", C.c_int, vp, C.c_size_t)
    sig("same_batch_pack_bursts", C.c_int, vp, u32, vp, C.c_size_t, P(C.c_size_t))
    sig("same_batch_read_trace", C.c_int, vp, u32, P(SymbolTrace), C.c_size_t, P(C.c_size_t))
    sig("same_batch_last_kernel_ms", C.c_int, vp, P(f32))
    sig("same_batch_last_demod_kernel_ms", C.c_int, vp, P(f32))
    sig("same_batch_set_kernel_timing", None, vp, C.c_int)
    sig("same_batch_kernel_name", C.c_char_p, vp)

    sig("same_rx_build", C.c_int, vp, C.c_int, P(vp))
    sig("same_rx_free", None, vp)
    sig("same_rx_process", C.c_int, vp, vp, C.c_size_t, P(C.c_size_t), P(Event))
    sig("same_rx_flush", C.c_int, vp, P(Event))
    sig("same_rx_reset", C.c_int, vp)
    sig("same_rx_input_rate", u32, vp)
    sig("same_rx_input_sample_counter", u64, vp)

    sig("same_synth_afsk_device", C.c_int, vp, u32, C.c_size_t, u32, u64, f32, u32, C.c_int, vp)
    sig("same_synth_payload", u32, u64, u32, C.c_char_p, u32)
    _lib = L
    return L


def _check(rc: int):
    if rc < 0:
        raise SameError(rc, load_library().same_last_error().decode(errors="replace"))
    return rc


class SameReceiverBuilder:
    """`sameold::SameReceiverBuilder` (receiver/builder.rs)."""

    def __init__(self, input_rate: int = 22050):
        self._L = load_library()
        self._h = C.c_void_p(self._L.same_rx_builder_new(input_rate))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.same_rx_builder_free(self._h)
            self._h = None

    # setters return self, as the reference's `&mut Self` chaining does
    def with_dc_blocker_length(self, length: float):
        self._L.same_rx_builder_with_dc_blocker_length(self._h, length); return self

    def with_agc_bandwidth(self, bw: float):
        self._L.same_rx_builder_with_agc_bandwidth(self._h, bw); return self

    def with_agc_gain_limits(self, mn: float, mx: float):
        self._L.same_rx_builder_with_agc_gain_limits(self._h, mn, mx); return self

    def with_timing_bandwidth(self, unlocked: float, locked: float):
        self._L.same_rx_builder_with_timing_bandwidth(self._h, unlocked, locked); return self

    def with_timing_max_deviation(self, max_dev: float):
        self._L.same_rx_builder_with_timing_max_deviation(self._h, max_dev); return self

    def with_squelch_power(self, open_: float, close: float):
        self._L.same_rx_builder_with_squelch_power(self._h, open_, close); return self

    def with_squelch_bandwidth(self, bw: float):
        self._L.same_rx_builder_with_squelch_bandwidth(self._h, bw); return self

    def with_preamble_max_errors(self, n: int):
        self._L.same_rx_builder_with_preamble_max_errors(self._h, n); return self

    def with_adaptive_equalizer(self, nfeedforward=6, nfeedback=4, relaxation=0.05, regularization=1.0e-6):
        self._L.same_rx_builder_with_adaptive_equalizer(self._h, nfeedforward, nfeedback, relaxation, regularization)
        return self

    def without_adaptive_equalizer(self):
        self._L.same_rx_builder_without_adaptive_equalizer(self._h); return self

    def with_frame_prefix_max_errors(self, n: int):
        self._L.same_rx_builder_with_frame_prefix_max_errors(self._h, n); return self

    def with_frame_max_invalid(self, n: int):
        self._L.same_rx_builder_with_frame_max_invalid(self._h, n); return self

    def samedec(self):
        """samedec's settings (crates/samedec/src/main.rs:29-37): AGC limits 1/32767 .. 1/200."""
        return self.with_agc_gain_limits(np.float32(1.0) / np.float32(32767.0), np.float32(1.0) / np.float32(200.0))

    # getters
    def input_rate(self) -> int:
        return self._L.same_rx_builder_input_rate(self._h)

    def dc_blocker_length(self) -> float:
        return self._L.same_rx_builder_dc_blocker_length(self._h)

    def agc_bandwidth(self) -> float:
        return self._L.same_rx_builder_agc_bandwidth(self._h)

    def agc_gain_limits(self):
        a = (C.c_float * 2)()
        self._L.same_rx_builder_agc_gain_limits(self._h, a)
        return (a[0], a[1])

    def timing_bandwidth(self):
        a, b = C.c_float(), C.c_float()
        self._L.same_rx_builder_timing_bandwidth(self._h, C.byref(a), C.byref(b))
        return (a.value, b.value)

    def timing_max_deviation(self) -> float:
        return self._L.same_rx_builder_timing_max_deviation(self._h)

    def squelch_power(self):
        a, b = C.c_float(), C.c_float()
        self._L.same_rx_builder_squelch_power(self._h, C.byref(a), C.byref(b))
        return (a.value, b.value)

    def squelch_bandwidth(self) -> float:
        return self._L.same_rx_builder_squelch_bandwidth(self._h)

    def preamble_max_errors(self) -> int:
        return self._L.same_rx_builder_preamble_max_errors(self._h)

    def adaptive_equalizer(self):
        nff, nfb, r, g = C.c_uint32(), C.c_uint32(), C.c_float(), C.c_float()
        if not self._L.same_rx_builder_adaptive_equalizer(self._h, C.byref(nff), C.byref(nfb), C.byref(r), C.byref(g)):
            return None
        return (nff.value, nfb.value, r.value, g.value)

    def frame_prefix_max_errors(self) -> int:
        return self._L.same_rx_builder_frame_prefix_max_errors(self._h)

    def frame_max_invalid(self) -> int:
        return self._L.same_rx_builder_frame_max_invalid(self._h)

    def build(self, device: int = 0) -> "SameReceiver":
        return SameReceiver(self, device)

    def build_batch(self, n_channels: int, device: int = 0, link_only: bool = False,
                    trace_symbols: bool = False, generic_kernel: bool = False,
                    time_parallel: bool = False, relaxed: bool = False, call_invariant: bool = False) -> "SameBatchReceiver":
        """call_invariant (SAME_BATCH_CALL_INVARIANT): the stream is demodulated in windows that begin at fixed stream positions,
        so the events do not depend on how it is cut into calls (they arrive when a window's last sample has; flush() brings in
        what is waiting)."""
        return SameBatchReceiver(self, n_channels, device, link_only, trace_symbols, generic_kernel, time_parallel, relaxed, call_invariant)


class SameBatchReceiver:
    """n independent `SameReceiver`s advancing in lockstep on one MI355X."""

    def __init__(self, builder: SameReceiverBuilder, n_channels: int, device: int = 0,
                 link_only: bool = False, trace_symbols: bool = False, generic_kernel: bool = False,
                 time_parallel: bool = False, relaxed: bool = False, call_invariant: bool = False):
        self._L = load_library()
        h = C.c_void_p()
        flags = ((BATCH_LINK_ONLY if link_only else 0) | (BATCH_TRACE_SYMBOLS if trace_symbols else 0)
                 | (BATCH_GENERIC_KERNEL if generic_kernel else 0) | (BATCH_TIME_PARALLEL if time_parallel else 0)
                 | (BATCH_RELAXED if relaxed else 0) | (BATCH_CALL_INVARIANT if call_invariant else 0))
        _check(self._L.same_batch_new(builder._h, n_channels, device, flags, C.byref(h)))
        self._h = h
        self._inflight = []          # input tensors of launches that may still be running (process_tensor)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.same_batch_free(self._h)
            self._h = None

    @property
    def n_channels(self) -> int:
        return self._L.same_batch_n_channels(self._h)

    def input_rate(self) -> int:
        return self._L.same_batch_input_rate(self._h)

    def input_sample_counter(self) -> int:
        return self._L.same_batch_input_sample_counter(self._h)

    def kernel_name(self) -> str:
        return self._L.same_batch_kernel_name(self._h).decode()

    def set_call_window(self, samples: int) -> None:
        """call_invariant batches: the window length in samples, before the first sample (same_batch_set_call_window)."""
        _check(self._L.same_batch_set_call_window(self._h, int(samples)))

    def reset(self):
        _check(self._L.same_batch_reset(self._h))

    def time_parallel_config(self, max_chunks: int = 0, min_own_samples: int = 0, warmup_samples: int = 0):
        """SAME_BATCH_TIME_PARALLEL tuning (0 = default), see include/same_rx.h."""
        _check(self._L.same_batch_time_parallel_config(self._h, max_chunks, min_own_samples, warmup_samples))

    def time_parallel_chunks(self) -> int:
        """Chunks per channel of the most recent process call (1 = one strict launch)."""
        return self._L.same_batch_time_parallel_chunks(self._h)

    def time_parallel_per_channel(self) -> bool:
        """Whether that call's chunk boundaries were chosen per channel (channel-major input, see same_rx.h)."""
        return bool(self._L.same_batch_time_parallel_per_channel(self._h))

    def process_device_ptr(self, ptr: int, n_samples: int, layout: int = LAYOUT_TIME_MAJOR,
                           stream: Optional[int] = None, i16: bool = False):
        """Hot path: `ptr` is a device pointer (e.g. torch_tensor.data_ptr()).  `stream`: None = the
        library's own non-blocking stream (SAME_STREAM_OWN), otherwise a hipStream_t handle -- 0 is the
        legacy default stream, like any HIP API.  The caller keeps the buffer alive and unmodified until
        `sync()` or until the second-next process call returns, and orders the library's own stream after
        whatever produced the buffer (`order_after`); `process_tensor` does both."""
        fn = self._L.same_batch_process_device_i16 if i16 else self._L.same_batch_process_device
        _check(fn(self._h, C.c_void_p(ptr), n_samples, layout, C.c_void_p(STREAM_OWN if stream is None else stream)))

    def order_after(self, producer_stream: int):
        """Make the library's own stream wait for the work queued so far on `producer_stream`
        (a hipStream_t handle, 0 = the legacy default stream)."""
        _check(self._L.same_batch_order_after(self._h, C.c_void_p(producer_stream)))

    def process_tensor(self, x, layout: int = LAYOUT_TIME_MAJOR, stream: Optional[int] = None):
        """x: torch CUDA tensor, float32 or int16, [T, C] (time-major) or [C, T].

        stream None: runs on the library's own stream, ordered after the work already queued on
        torch's current stream of x's device (whatever produced x: a generator kernel, a cast, a
        `.contiguous()` copy).  A reference to x is held until its launch can no longer be running
        (two process calls later, or `sync()`), so torch's caching allocator cannot hand the memory
        to someone else while the kernel still reads it."""
        import torch
        assert x.is_cuda and x.is_contiguous()
        if layout == LAYOUT_TIME_MAJOR:
            n, ch = x.shape
        else:
            ch, n = x.shape
        assert ch == self.n_channels
        if x.dtype not in (torch.float32, torch.int16):
            raise TypeError("float32 or int16 input")
        if stream is None:
            self.order_after(torch.cuda.current_stream(x.device).cuda_stream)
        self._inflight.append(x)
        self.process_device_ptr(x.data_ptr(), n, layout, stream, x.dtype == torch.int16)
        # (trimmed AFTER the call: it is the call that collects launch k-2, whose input may be dropped only then)
        if len(self._inflight) > 2:
            del self._inflight[0]

    def process_host(self, x: np.ndarray, layout: int = LAYOUT_TIME_MAJOR):
        """x: numpy float32/int16 array, [T, C] (time-major) or [C, T]."""
        x = np.ascontiguousarray(x)
        if x.ndim == 1:
            x = x.reshape(-1, 1) if layout == LAYOUT_TIME_MAJOR else x.reshape(1, -1)
        n, ch = x.shape if layout == LAYOUT_TIME_MAJOR else x.shape[::-1]
        assert ch == self.n_channels, (ch, self.n_channels)
        if x.dtype == np.int16:
            _check(self._L.same_batch_process_host_i16(self._h, C.c_void_p(x.ctypes.data), n, layout))
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            _check(self._L.same_batch_process_host(self._h, C.c_void_p(x.ctypes.data), n, layout))

    def flush(self):
        _check(self._L.same_batch_flush(self._h))

    def sync(self):
        _check(self._L.same_batch_sync(self._h))
        self._inflight.clear()

    def set_kernel_timing(self, enable: bool):
        self._L.same_batch_set_kernel_timing(self._h, int(enable))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(self._L.same_batch_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_demod_kernel_ms(self) -> float:
        """The demodulation kernel alone (a channel-major time-parallel launch's last_kernel_ms covers its planning kernels too)."""
        ms = C.c_float()
        _check(self._L.same_batch_last_demod_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def poll_events(self, max_events: int = 1 << 20) -> List[Event]:
        """Drain the events already on the host, ordered by (channel, sample_counter) within
        each call's batch.  Non-blocking: call sync() first to include a launch in flight."""
        out: List[Event] = []
        buf = (Event * 1024)()
        while len(out) < max_events:
            n, left = C.c_size_t(), C.c_size_t()
            _check(self._L.same_batch_poll_events(self._h, buf, min(1024, max_events - len(out)),
                                                  C.byref(n), C.byref(left)))
            for i in range(n.value):
                e = Event()
                C.memmove(C.byref(e), C.byref(buf[i]), C.sizeof(Event))
                out.append(e)
            if left.value == 0 or n.value == 0:
                break
        return out

    def poll_events_np(self, max_events: int = 1 << 22) -> np.ndarray:
        """Drain events into one numpy structured array (no per-event Python objects)."""
        n_pending = self._L.same_batch_pending_events(self._h)
        n_take = min(n_pending, max_events)
        out = np.zeros(n_take, dtype=EVENT_DTYPE)
        if n_take:
            n, left = C.c_size_t(), C.c_size_t()
            _check(self._L.same_batch_poll_events(self._h, C.cast(out.ctypes.data, C.POINTER(Event)), n_take,
                                                  C.byref(n), C.byref(left)))
            out = out[: n.value]
        return out

    def pending_events(self) -> int:
        """Events the host already has (same_batch_pending_events): does not wait for launches still in flight, makes no
        copies."""
        return int(self._L.same_batch_pending_events(self._h))

    def peek_events_np(self) -> np.ndarray:
        """The queued events as a read-only numpy view of the array the handle builds for them
        (same_batch_peek_events: built once per batch of new events, re-used while events are only
        dropped; no copy on the Python side).  Valid until the next call on this receiver other than
        `drop_events`; copy what you keep, then `drop_events(len(view))`."""
        ptr, n = C.POINTER(Event)(), C.c_size_t()
        _check(self._L.same_batch_peek_events(self._h, C.byref(ptr), C.byref(n)))
        if not n.value:
            return np.zeros(0, dtype=EVENT_DTYPE)
        buf = (C.c_char * (n.value * C.sizeof(Event))).from_address(C.addressof(ptr.contents))
        out = np.frombuffer(buf, dtype=EVENT_DTYPE, count=n.value)
        out.flags.writeable = False
        return out

    def pack_bursts_np(self, first_channel: int = 0, out: np.ndarray | None = None) -> np.ndarray:
        """The queued bursts as uint8 [n, 304] records with global channel numbers (same_batch_pack_bursts;
        the layout of sameold_amd.distributed).  The queue is left as it is.  `out`: a C-contiguous uint8 [cap, 304] array
        to fill instead of a fresh one (a consumer that packs every step keeps its pages: at 131 072 channels the fresh
        40 MB array costs more in page faults than the copy, tools/big_host_split.py); the result is then the view
        out[:n], and a ValueError if the bursts do not fit."""
        n = C.c_size_t()
        _check(self._L.same_batch_pack_bursts(self._h, first_channel, None, 0, C.byref(n)))
        if out is None:
            out = np.empty((n.value, 304), dtype=np.uint8)
        else:
            if out.dtype != np.uint8 or out.ndim != 2 or out.shape[1] != 304 or not out.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous uint8 [cap, 304] array")
            if out.shape[0] < n.value:
                raise ValueError(f"out holds {out.shape[0]} records, {n.value} bursts are queued")
            out = out[:n.value]
        if n.value:
            _check(self._L.same_batch_pack_bursts(self._h, first_channel, C.c_void_p(out.ctypes.data), n.value, C.byref(n)))
        return out

    def drop_events(self, n: int) -> None:
        _check(self._L.same_batch_drop_events(self._h, n))

    def read_trace(self, channel: int, cap: int = 4096) -> np.ndarray:
        buf = (SymbolTrace * cap)()
        n = C.c_size_t()
        _check(self._L.same_batch_read_trace(self._h, channel, buf, cap, C.byref(n)))
        dt = np.dtype([("sample_counter", "<u8"), ("zero", "<f4"), ("sym", "<f4"), ("err", "<f4"), ("next", "<f4")])
        return np.frombuffer(buf, dtype=dt, count=n.value).copy()


class SameReceiver:
    """`sameold::SameReceiver` (receiver.rs:71-224) backed by a one-channel GPU batch."""

    def __init__(self, builder: SameReceiverBuilder, device: int = 0):
        self._L = load_library()
        h = C.c_void_p()
        _check(self._L.same_rx_build(builder._h, device, C.byref(h)))
        self._h = h
        self.consumed = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.same_rx_free(self._h)
            self._h = None

    def input_rate(self) -> int:
        return self._L.same_rx_input_rate(self._h)

    def input_sample_counter(self) -> int:
        return self._L.same_rx_input_sample_counter(self._h)

    def reset(self):
        _check(self._L.same_rx_reset(self._h))

    def iter_events(self, samples: Sequence[float]) -> Iterator[Event]:
        """`iter_events()` (receiver.rs:119-130): lazy; `self.consumed` tracks how many
        input samples have been taken, as the reference's iterator adaptor would."""
        x = np.ascontiguousarray(np.asarray(samples), dtype=np.float32)
        off = 0
        self.consumed = 0
        while True:
            ev = Event()
            used = C.c_size_t()
            got = _check(self._L.same_rx_process(self._h, C.c_void_p(x.ctypes.data + 4 * off),
                                                 len(x) - off, C.byref(used), C.byref(ev)))
            off += used.value
            self.consumed = off
            if not got:
                return
            yield ev

    def iter_messages(self, samples: Sequence[float]) -> Iterator[str]:
        """`iter_messages()` (receiver.rs:155-161)."""
        for ev in self.iter_events(samples):
            m = ev.message()
            if m is not None:
                yield m

    def flush(self) -> Optional[str]:
        """`flush()` (receiver.rs:216-224)."""
        ev = Event()
        got = _check(self._L.same_rx_flush(self._h, C.byref(ev)))
        return ev.message() if got else None


def synth_afsk(n_channels: int, n_samples: int, input_rate: int = 22050, seed: int = 1,
               noise_sigma: float = 0.0, integer_symbols: bool = False, device: int = 0):
    """Seeded synthetic multi-channel AFSK workload as a torch CUDA tensor [T, C] float32."""
    import torch
    x = torch.empty((n_samples, n_channels), dtype=torch.float32, device=f"cuda:{device}")
    stream = torch.cuda.current_stream(device).cuda_stream
    _check(load_library().same_synth_afsk_device(C.c_void_p(x.data_ptr()), n_channels, n_samples, input_rate,
                                                 seed, noise_sigma, 1 if integer_symbols else 0, device,
                                                 C.c_void_p(stream)))
    return x


def synth_payload(seed: int, channel: int) -> bytes:
    buf = C.create_string_buffer(256)
    n = load_library().same_synth_payload(seed, channel, buf, 256)
    return buf.raw[:n]
