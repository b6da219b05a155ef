"""CPU-side checks of the C ABI: the library loads, exports every symbol include/same_rx.h
declares, the builder mirrors SameReceiverBuilder's defaults and clamping, and compute
entry points fail loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
import sameold_amd as sa
from sameold_amd import build as sbuild


@pytest.fixture(scope="module")
def lib():
    sbuild.build()
    return sa.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "same_rx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(same_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) > 50
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.same_rx_abi_version() == 2


def test_builder_defaults_and_clamping(lib):
    """rx/builder.rs:50-67 defaults, :95-279 setters, :369-425 equalizer."""
    b = sa.SameReceiverBuilder(22050)
    assert b.input_rate() == 22050
    assert b.dc_blocker_length() == np.float32(0.38)
    assert b.agc_bandwidth() == np.float32(0.01)
    assert b.agc_gain_limits() == (0.0, 1.0e6)
    assert b.timing_bandwidth() == (0.125, np.float32(0.05))
    assert b.timing_max_deviation() == np.float32(0.01)
    assert b.squelch_power() == (np.float32(0.10), np.float32(0.05))
    assert b.squelch_bandwidth() == 0.125
    assert b.preamble_max_errors() == 2
    assert b.adaptive_equalizer() == (6, 4, np.float32(0.05), np.float32(1.0e-6))
    assert b.frame_prefix_max_errors() == 2 and b.frame_max_invalid() == 5
    b.with_timing_bandwidth(2.0, 3.0)
    assert b.timing_bandwidth() == (1.0, 1.0)
    b.with_timing_bandwidth(0.1, 0.5)
    assert b.timing_bandwidth() == (np.float32(0.1), np.float32(0.1))
    b.with_squelch_power(2.0, 1.5)
    assert b.squelch_power() == (1.0, 1.5)
    b.with_frame_prefix_max_errors(12)
    assert b.frame_prefix_max_errors() == 7
    b.with_timing_max_deviation(0.9)
    assert b.timing_max_deviation() == 0.5
    b.with_dc_blocker_length(-1.0)
    assert b.dc_blocker_length() == 0.0
    b.with_adaptive_equalizer(0, 9, 2.0, -1.0)
    assert b.adaptive_equalizer() == (1, 1, 1.0, 0.0)
    b.without_adaptive_equalizer()
    assert b.adaptive_equalizer() is None
    assert sa.SameReceiverBuilder().input_rate() == 22050


def test_no_cpu_fallback(lib):
    """Without a GPU the compute entry points must fail, not silently run elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sa.SameError) as e:
        sa.SameReceiverBuilder(22050).build_batch(4)
    assert e.value.code in (-5, -6)
    with pytest.raises(sa.SameError):
        sa.SameReceiverBuilder(22050).build()


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under sameold_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sameold_amd")):
        if "build" in dirpath.split(os.sep)[-1:]:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "same_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_synth_payload_is_valid_header(lib):
    from oracle import binding as ob
    for ch in range(64):
        p = sa.synth_payload(1234, ch)
        a, b = C.c_size_t(), C.c_size_t()
        assert ob.lib().so_check_header(p, len(p), C.byref(a), C.byref(b)) == 0
        assert b.value == len(p)
        assert all(ob.lib().so_is_allowed_byte(c) for c in p)
