#!/usr/bin/env python3
"""Regenerates tests/golden/link_events.json and soft_symbols_npt.npz from the oracle.

Run only after tests/test_oracle_golden.py::test_sample_text passes (the oracle is
then pinned against the reference's own sample/*.txt).  The reference itself is Rust
and cannot be run in this image, so these fixtures are regression vectors for the
oracle and the exact-parity targets for the HIP path, not outputs of the reference.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import binding as ob  # noqa: E402


def main():
    out = {}
    for name in ["npt", "two_and_two", "long_message"]:
        pcm = np.fromfile(os.path.join(HERE, f"{name}.22050.s16le.bin"), dtype="<i2")
        rx = ob.Receiver(ob.samedec_config())
        evs = rx.run(pcm)
        out[name] = {
            "config": "samedec (agc limits 1/32767..1/200, library defaults otherwise)",
            "n_samples": int(len(pcm)),
            "events": [[int(e.kind), int(e.sample_counter), int(e.symbol_count), e.data().hex()] for e in evs],
            "lines": ob.samedec_lines(pcm),
        }
    with open(os.path.join(HERE, "link_events.json"), "w") as f:
        json.dump(out, f, indent=1)
    rx = ob.Receiver(ob.samedec_config(), link_only=True)
    rx.enable_trace(4096)
    pcm = np.fromfile(os.path.join(HERE, "npt.22050.s16le.bin"), dtype="<i2")
    rx.run(pcm)
    tr = rx.trace()
    np.savez_compressed(os.path.join(HERE, "soft_symbols_npt.npz"), trace=tr[:1024])


if __name__ == "__main__":
    main()
