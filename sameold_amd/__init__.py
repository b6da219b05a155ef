"""sameold_amd -- MI355X-native batched SAME/EAS AFSK demodulator (hot path of sameold).

The package holds only what the path needs: csrc/ (gfx950 HIP kernels + the C ABI of
include/same_rx.h) and receiver.py (a Python mirror of SameReceiverBuilder/SameReceiver).
"""
from .receiver import (  # noqa: F401
    Event, SameBatchReceiver, SameError, SameReceiver, SameReceiverBuilder,
    LAYOUT_CHANNEL_MAJOR, LAYOUT_TIME_MAJOR,
    LINK_BURST, LINK_NO_CARRIER, LINK_READING, LINK_SEARCHING,
    TRANSPORT_ASSEMBLING, TRANSPORT_IDLE, TRANSPORT_MSG_END, TRANSPORT_MSG_ERR, TRANSPORT_MSG_START,
    load_library, synth_afsk, synth_payload,
)
