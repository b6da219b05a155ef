"""ctypes binding for the parity oracle (oracle/libsame_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from sameold_amd/ (the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Iterator, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsame_oracle.so")

LINK_NO_CARRIER, LINK_SEARCHING, LINK_READING, LINK_BURST = 0, 1, 2, 3
TRANSPORT_IDLE, TRANSPORT_ASSEMBLING, TRANSPORT_MSG_START, TRANSPORT_MSG_END, TRANSPORT_MSG_ERR = 16, 17, 18, 19, 20
EVENT_MAX_BYTES = 288
SQ_NO_CARRIER, SQ_DROPPED, SQ_READING, SQ_READY = 0, 1, 2, 3

KIND_NAMES = {
    0: "no_carrier", 1: "searching", 2: "reading", 3: "burst",
    16: "idle", 17: "assembling", 18: "message_start", 19: "message_end", 20: "message_err",
}


class Config(C.Structure):
    _fields_ = [
        ("input_rate", C.c_uint32),
        ("dc_blocker_len", C.c_float),
        ("agc_bandwidth", C.c_float),
        ("agc_gain_min", C.c_float),
        ("agc_gain_max", C.c_float),
        ("timing_bw_unlocked", C.c_float),
        ("timing_bw_locked", C.c_float),
        ("timing_max_deviation", C.c_float),
        ("squelch_power_open", C.c_float),
        ("squelch_power_close", C.c_float),
        ("squelch_bandwidth", C.c_float),
        ("preamble_max_errors", C.c_uint32),
        ("eq_enabled", C.c_uint32),
        ("eq_nff", C.c_uint32),
        ("eq_nfb", C.c_uint32),
        ("eq_relaxation", C.c_float),
        ("eq_regularization", C.c_float),
        ("frame_prefix_max_errors", C.c_uint32),
        ("frame_max_invalid", C.c_uint32),
    ]


class Event(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("len", C.c_uint32),
        ("sample_counter", C.c_uint64),
        ("symbol_count", C.c_uint64),
        ("aux", C.c_uint32),
        ("aux2", C.c_uint32),
        ("bytes", C.c_uint8 * EVENT_MAX_BYTES),
    ]

    def data(self) -> bytes:
        return bytes(self.bytes[: min(self.len, EVENT_MAX_BYTES)])

    def as_tuple(self):
        return (int(self.kind), int(self.sample_counter), self.data())

    def __repr__(self):
        return f"Event({KIND_NAMES.get(self.kind, self.kind)}@{self.sample_counter}, {self.data()!r})"


class SymbolTrace(C.Structure):
    _fields_ = [
        ("sample_counter", C.c_uint64),
        ("zero", C.c_float),
        ("sym", C.c_float),
        ("err", C.c_float),
        ("samples_until_next_ted", C.c_float),
    ]


class Derived(C.Structure):
    _fields_ = [
        ("sps", C.c_float), ("ntaps", C.c_uint32), ("dc_len", C.c_uint32),
        ("agc_bw", C.c_float), ("agc_gain0", C.c_float), ("samples_per_ted", C.c_float),
        ("period_min", C.c_float), ("period_max", C.c_float),
        ("alpha_unlocked", C.c_float), ("beta_unlocked", C.c_float),
        ("alpha_locked", C.c_float), ("beta_locked", C.c_float),
        ("power_open", C.c_float), ("power_close", C.c_float), ("power_bw", C.c_float),
    ]


class Agc(C.Structure):
    _fields_ = [("bandwidth", C.c_float), ("min_gain", C.c_float), ("max_gain", C.c_float),
                ("gain", C.c_float), ("locked", C.c_int)]


class Ted(C.Structure):
    _fields_ = [("h", C.c_float * 3), ("counter", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("samples_per_ted", C.c_float), ("period_min", C.c_float), ("period_max", C.c_float),
                ("alpha", C.c_float), ("beta", C.c_float), ("period_avg", C.c_float),
                ("period_inst", C.c_float), ("ted", Ted)]


class Squelch(C.Structure):
    _fields_ = [("max_errors", C.c_uint32), ("power_open", C.c_float), ("power_close", C.c_float),
                ("sync_to", C.c_uint32), ("data", C.c_uint32), ("pt_bw", C.c_float),
                ("pt_power", C.c_float), ("hist", C.c_float * 64), ("hist_len", C.c_uint32),
                ("hist_head", C.c_uint32), ("phist", C.c_uint8 * 32), ("phist_len", C.c_uint32),
                ("phist_head", C.c_uint32), ("symbol_counter", C.c_uint64),
                ("sample_clock", C.c_int), ("sync_lock", C.c_int)]


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "same_oracle.c")
    hdr = os.path.join(_HERE, "same_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "libsame_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p
    f32p = P(C.c_float)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("so_config_default", None, P(Config), C.c_uint32)
    sig("so_config_samedec", None, P(Config), C.c_uint32)
    sig("so_config_with_dc_blocker_length", None, P(Config), C.c_float)
    sig("so_config_with_agc_bandwidth", None, P(Config), C.c_float)
    sig("so_config_with_agc_gain_limits", None, P(Config), C.c_float, C.c_float)
    sig("so_config_with_timing_bandwidth", None, P(Config), C.c_float, C.c_float)
    sig("so_config_with_timing_max_deviation", None, P(Config), C.c_float)
    sig("so_config_with_squelch_power", None, P(Config), C.c_float, C.c_float)
    sig("so_config_with_squelch_bandwidth", None, P(Config), C.c_float)
    sig("so_config_with_preamble_max_errors", None, P(Config), C.c_uint32)
    sig("so_config_with_adaptive_equalizer", None, P(Config), C.c_uint32, C.c_uint32, C.c_float, C.c_float)
    sig("so_config_without_adaptive_equalizer", None, P(Config))
    sig("so_config_with_frame_prefix_max_errors", None, P(Config), C.c_uint32)
    sig("so_config_with_frame_max_invalid", None, P(Config), C.c_uint32)

    sig("so_rx_new", C.c_int, P(Config), P(vp))
    sig("so_rx_free", None, vp)
    sig("so_rx_clone", vp, vp)
    sig("so_rx_reset", None, vp)
    sig("so_rx_input_rate", C.c_uint32, vp)
    sig("so_rx_input_sample_counter", C.c_uint64, vp)
    sig("so_rx_set_input_sample_counter", None, vp, C.c_uint64)
    sig("so_rx_force_eom_pending", C.c_int, vp, P(C.c_uint64))
    sig("so_rx_process", C.c_int, vp, vp, C.c_size_t, P(C.c_size_t), P(Event))
    sig("so_rx_process_i16", C.c_int, vp, vp, C.c_size_t, P(C.c_size_t), P(Event))
    sig("so_rx_run", C.c_size_t, vp, vp, C.c_size_t, P(Event), C.c_size_t)
    sig("so_rx_run_i16", C.c_size_t, vp, vp, C.c_size_t, P(Event), C.c_size_t)
    sig("so_rx_flush", C.c_size_t, vp, P(Event), C.c_size_t)
    sig("so_rx_set_trace", None, vp, P(SymbolTrace), C.c_size_t)
    sig("so_rx_trace_count", C.c_size_t, vp)
    sig("so_rx_set_link_only", None, vp, C.c_int)
    sig("so_batch_run_time_major", C.c_size_t, P(Config), vp, C.c_size_t, C.c_size_t, C.c_int, P(Event), C.c_size_t)
    sig("so_batch_run_channel_major", C.c_size_t, P(Config), vp, C.c_size_t, C.c_size_t, C.c_int, P(C.c_int), C.c_int,
        P(Event), C.c_size_t)

    sig("so_derive", None, P(Config), P(Derived))
    sig("so_matched_filter_taps", None, C.c_uint32, f32p, P(C.c_uint32))
    sig("so_cisoid_matched_filter", None, C.c_uint32, C.c_float, f32p, f32p)
    sig("so_compute_loop_alphabeta", None, C.c_float, f32p, f32p)

    sig("so_dcblock_new", vp, C.c_uint32)
    sig("so_dcblock_free", None, vp)
    sig("so_dcblock_filter", C.c_float, vp, C.c_float)
    sig("so_movavg_new", vp, C.c_uint32)
    sig("so_movavg_free", None, vp)
    sig("so_movavg_filter", C.c_float, vp, C.c_float, f32p)
    sig("so_agc_init", None, P(Agc), C.c_float, C.c_float, C.c_float)
    sig("so_agc_reset", None, P(Agc))
    sig("so_agc_input", C.c_float, P(Agc), C.c_float)
    sig("so_mac_ff", C.c_float, f32p, C.c_size_t, f32p, C.c_size_t)
    sig("so_demod_new", vp, C.c_uint32)
    sig("so_demod_free", None, vp)
    sig("so_demod_push", None, vp, C.c_float)
    sig("so_demod_demod", C.c_float, vp)
    sig("so_demod_ntaps", C.c_uint32, vp)
    sig("so_ted_reset", None, P(Ted))
    sig("so_ted_input", C.c_int, P(Ted), C.c_float, f32p, f32p, f32p)
    sig("so_zero_crossing_metric", C.c_float, f32p)
    sig("so_timing_init", None, P(Timing), C.c_float, C.c_float, C.c_float)
    sig("so_timing_reset", None, P(Timing))
    sig("so_timing_set_bw", None, P(Timing), C.c_float)
    sig("so_timing_advance", C.c_float, P(Timing), C.c_float, C.c_int, C.c_float)
    sig("so_timing_input", C.c_float, P(Timing), C.c_float, C.c_float, P(C.c_int), f32p, f32p, f32p)
    sig("so_squelch_init", None, P(Squelch), C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float)
    sig("so_squelch_reset", None, P(Squelch))
    sig("so_squelch_end", None, P(Squelch))
    sig("so_squelch_lock", None, P(Squelch), C.c_int)
    sig("so_squelch_input", C.c_int, P(Squelch), f32p, P(C.c_int), f32p, P(C.c_uint64), f32p)
    sig("so_squelch_is_sync", C.c_int, P(Squelch))
    sig("so_code_search", C.c_uint32, P(C.c_uint32), C.c_uint32, C.c_float)
    sig("so_power_track", C.c_float, f32p, C.c_float, C.c_float)
    sig("so_equalizer_new", vp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_uint32)
    sig("so_equalizer_free", None, vp)
    sig("so_equalizer_reset", None, vp)
    sig("so_equalizer_enable", None, vp, C.c_int)
    sig("so_equalizer_train", C.c_int, vp)
    sig("so_equalizer_mode", C.c_int, vp, P(C.c_uint32), P(C.c_uint32))
    sig("so_equalizer_estimate_symbol", C.c_int, vp, f32p, f32p)
    sig("so_equalizer_input", C.c_uint8, vp, f32p, f32p)
    sig("so_nlms_update", None, C.c_float, C.c_float, C.c_float, f32p, C.c_size_t, f32p)
    sig("so_framer_new", vp, C.c_uint32, C.c_uint32)
    sig("so_framer_free", None, vp)
    sig("so_framer_reset", None, vp)
    sig("so_framer_input", C.c_int, vp, C.c_uint8, C.c_uint64, C.c_int, P(P(C.c_uint8)), P(C.c_size_t))
    sig("so_framer_end", C.c_int, vp, P(P(C.c_uint8)), P(C.c_size_t))
    sig("so_framer_state", C.c_int, vp)
    sig("so_message_prefix_errors", C.c_uint32, C.c_uint32)
    sig("so_is_allowed_byte", C.c_int, C.c_uint8)
    sig("so_bit_vote_detect", None, C.c_uint8, C.c_uint8, P(C.c_uint8), P(C.c_uint32))
    sig("so_bit_vote_correct", None, C.c_uint8, C.c_uint8, C.c_uint8, P(C.c_uint8), P(C.c_uint32))
    sig("so_check_header", C.c_int, C.c_char_p, C.c_size_t, P(C.c_size_t), P(C.c_size_t))
    sig("so_estimate_message", C.c_uint32, P(C.c_char_p), P(C.c_size_t), C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p)
    sig("so_combine", C.c_int, P(C.c_char_p), P(C.c_size_t), C.c_uint32, P(Event))
    sig("so_assembler_new", vp)
    sig("so_assembler_free", None, vp)
    sig("so_assembler_reset", None, vp)
    sig("so_assembler_assemble", None, vp, C.c_char_p, C.c_size_t, C.c_uint64, P(Event))
    sig("so_assembler_idle", None, vp, C.c_uint64, P(Event))
    sig("so_max_interburst_symbols", C.c_uint64)
    sig("so_max_history_duration", C.c_uint64)
    sig("so_modulate_len", C.c_size_t, C.c_size_t, C.c_uint32, P(C.c_uint32))
    sig("so_modulate_afsk_bytes", None, C.c_char_p, C.c_size_t, C.c_uint32, f32p)
    _lib = L
    return L


def default_config(rate: int = 22050) -> Config:
    c = Config()
    lib().so_config_default(C.byref(c), rate)
    return c


def samedec_config(rate: int = 22050) -> Config:
    c = Config()
    lib().so_config_samedec(C.byref(c), rate)
    return c


def derive(cfg: Config) -> Derived:
    d = Derived()
    lib().so_derive(C.byref(cfg), C.byref(d))
    return d


def matched_filter_taps(fs: int) -> np.ndarray:
    n = C.c_uint32()
    lib().so_matched_filter_taps(fs, None, C.byref(n))
    out = np.zeros((n.value, 4), dtype=np.float32)
    lib().so_matched_filter_taps(fs, out.ctypes.data_as(C.POINTER(C.c_float)), None)
    return out


def modulate_afsk(data: bytes, fs: int = 22050) -> np.ndarray:
    """The reference's test modulator (rx/waveform.rs:73-104) over LSb-first bits."""
    n = lib().so_modulate_len(len(data), fs, None)
    out = np.zeros(n, dtype=np.float32)
    lib().so_modulate_afsk_bytes(data, len(data), fs, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


class Receiver:
    """Mirror of sameold::SameReceiver over the oracle."""

    def __init__(self, cfg: Optional[Config] = None, link_only: bool = False):
        self.cfg = cfg if cfg is not None else default_config()
        h = C.c_void_p()
        rc = lib().so_rx_new(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise ValueError(f"so_rx_new failed: {rc}")
        self._h = h
        self._trace = None
        if link_only:
            lib().so_rx_set_link_only(self._h, 1)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().so_rx_free(self._h)
            self._h = None

    def reset(self):
        lib().so_rx_reset(self._h)

    @property
    def input_rate(self) -> int:
        return lib().so_rx_input_rate(self._h)

    @property
    def input_sample_counter(self) -> int:
        return lib().so_rx_input_sample_counter(self._h)

    @input_sample_counter.setter
    def input_sample_counter(self, v: int):
        lib().so_rx_set_input_sample_counter(self._h, v)

    def force_eom_at_sample(self) -> Optional[int]:
        at = C.c_uint64()
        return int(at.value) if lib().so_rx_force_eom_pending(self._h, C.byref(at)) else None

    def enable_trace(self, cap: int):
        self._trace = (SymbolTrace * cap)()
        lib().so_rx_set_trace(self._h, self._trace, cap)

    def trace(self) -> np.ndarray:
        n = min(lib().so_rx_trace_count(self._h), len(self._trace))
        dt = np.dtype([("sample_counter", "<u8"), ("zero", "<f4"), ("sym", "<f4"),
                       ("err", "<f4"), ("next", "<f4")])
        return np.frombuffer(self._trace, dtype=dt, count=n).copy()

    def iter_events(self, x: np.ndarray) -> Iterator[Event]:
        """SameReceiver::iter_events (receiver.rs:119-130): lazy, consumes only as many
        samples as needed for the next event."""
        if x.dtype == np.int16:
            fn = lib().so_rx_process_i16
            itemsize = 2
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            fn = lib().so_rx_process
            itemsize = 4
        x = np.ascontiguousarray(x)
        off = 0
        n = x.shape[0]
        base = x.ctypes.data
        while True:
            ev = Event()
            used = C.c_size_t()
            got = fn(self._h, C.c_void_p(base + off * itemsize), n - off, C.byref(used), C.byref(ev))
            off += used.value
            self.consumed = off
            if not got:
                return
            yield ev

    def run(self, x: np.ndarray, cap: int = 4096) -> List[Event]:
        evs = (Event * cap)()
        x = np.ascontiguousarray(x)
        if x.dtype == np.int16:
            n = lib().so_rx_run_i16(self._h, C.c_void_p(x.ctypes.data), x.shape[0], evs, cap)
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            n = lib().so_rx_run(self._h, C.c_void_p(x.ctypes.data), x.shape[0], evs, cap)
        assert n <= cap, "event capacity exceeded"
        out = []
        for i in range(n):
            e = Event()
            C.memmove(C.byref(e), C.byref(evs[i]), C.sizeof(Event))
            out.append(e)
        return out

    def flush_first_message(self) -> Optional[Event]:
        """SameReceiver::flush (receiver.rs:216-224): feed 4 s of zeros, stop at the
        first Message(Ok(..))."""
        z = np.zeros(self.input_rate * 4, dtype=np.float32)
        for ev in self.iter_events(z):
            if ev.kind in (TRANSPORT_MSG_START, TRANSPORT_MSG_END):
                return ev
        return None


def samedec_lines(pcm: np.ndarray, rate: int = 22050) -> List[str]:
    """What `samedec --rate R --file X -- child` prints for decoder output
    (crates/samedec/src/app.rs:103-193): every Message in order, then EOF flushes
    until a flush yields nothing."""
    rx = Receiver(samedec_config(rate))
    lines = []
    for ev in rx.iter_events(pcm):
        if ev.kind == TRANSPORT_MSG_START:
            lines.append(ev.data().decode("ascii"))
        elif ev.kind == TRANSPORT_MSG_END:
            lines.append("NNNN")
    while True:
        ev = rx.flush_first_message()
        if ev is None:
            break
        lines.append(ev.data().decode("ascii") if ev.kind == TRANSPORT_MSG_START else "NNNN")
    return lines


def batch_run_time_major(cfg: Config, x: np.ndarray, nthreads: int, cap: int = 1 << 16):
    """CPU baseline: x is [T, C] float32 time-major; returns (n_events, events)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    T, Cn = x.shape
    evs = (Event * cap)()
    n = lib().so_batch_run_time_major(C.byref(cfg), C.c_void_p(x.ctypes.data), Cn, T, nthreads, evs, cap)
    return n, evs


_EVENT_NP = np.dtype([("kind", "<u4"), ("len", "<u4"), ("sample_counter", "<u8"), ("symbol_count", "<u8"),
                      ("aux", "<u4"), ("aux2", "<u4"), ("bytes", "u1", (EVENT_MAX_BYTES,))])
assert _EVENT_NP.itemsize == C.sizeof(Event)


def events_by_channel(n: int, evs, n_channels: int):
    """Link events of a batch run (aux = channel) -> per channel list of (kind, sample_counter, bytes),
    in time order.  Workers append under a mutex, so events of one channel are already in order."""
    a = np.frombuffer(evs, dtype=_EVENT_NP, count=min(n, len(evs)))
    out = [[] for _ in range(n_channels)]
    order = np.argsort(a["aux"], kind="stable")
    for i in order:
        r = a[i]
        out[int(r["aux"])].append((int(r["kind"]), int(r["sample_counter"]),
                                   r["bytes"][: min(int(r["len"]), EVENT_MAX_BYTES)].tobytes()))
    return out


def physical_cores() -> List[int]:
    """One allowed logical CPU per physical core (the first sibling of each core this process may run on)."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for cpu in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list") as f:
                key = f.read().strip()
        except OSError:
            key = str(cpu)
        if key not in seen:
            seen.add(key)
            out.append(cpu)
    return out


def batch_run_channel_major(cfg: Config, x: np.ndarray, cpus: Optional[List[int]] = None, nthreads: Optional[int] = None,
                            reps: int = 1, cap: int = 1 << 16):
    """CPU baseline proper: x is [C, T] float32, every channel contiguous; one worker per entry of
    `cpus` (pinned) or `nthreads` unpinned workers.  Returns (n_events of the first repetition, events)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    Cn, T = x.shape
    evs = (Event * cap)()
    if cpus:
        arr = (C.c_int * len(cpus))(*cpus)
        n = lib().so_batch_run_channel_major(C.byref(cfg), C.c_void_p(x.ctypes.data), Cn, T, len(cpus), arr, reps, evs, cap)
    else:
        n = lib().so_batch_run_channel_major(C.byref(cfg), C.c_void_p(x.ctypes.data), Cn, T, nthreads or 1, None, reps, evs, cap)
    return n, evs
