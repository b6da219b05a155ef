#!/usr/bin/env python3
"""Alert command used by tests/test_samedec_gpu.py: checks the SAMEDEC_* environment that
samedec_gpu hands to its child (crates/samedec/src/spawner.rs:33-76) against one entry of
tests/golden/samedec_child_env.json, drains the audio on standard input, and prints "+OK"
exactly like the reference's sample scripts do.  usage: samedec_child_check.py <entry> [<count file>]"""
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "..", "golden", "samedec_child_env.json")) as f:
    want = json.load(f)[sys.argv[1]]
env = os.environ
n = 0
while True:                      # children MUST read or close standard input
    chunk = sys.stdin.buffer.read(65536)
    if not chunk:
        break
    n += len(chunk)
if len(sys.argv) > 2:
    with open(sys.argv[2], "a") as f:
        f.write(f"{n}\n")
bad = []
for k, v in want.get("equals", {}).items():
    if env.get(k) != v:
        bad.append(f"{k}={env.get(k)!r}, want {v!r}")
if "lifetime" in want:
    life = int(env["SAMEDEC_PURGETIME"]) - int(env["SAMEDEC_ISSUETIME"])
    if life != want["lifetime"]:
        bad.append(f"lifetime {life}, want {want['lifetime']}")
if want.get("issue_equals_purge") and env.get("SAMEDEC_ISSUETIME") != env.get("SAMEDEC_PURGETIME"):
    bad.append("SAMEDEC_ISSUETIME != SAMEDEC_PURGETIME")
for k in ("SAMEDEC_RATE", "SAMEDEC_MSG", "SAMEDEC_ORG", "SAMEDEC_ORIGINATOR", "SAMEDEC_EVT", "SAMEDEC_EVENT",
          "SAMEDEC_SIGNIFICANCE", "SAMEDEC_SIG_NUM", "SAMEDEC_LOCATIONS", "SAMEDEC_ISSUETIME", "SAMEDEC_PURGETIME",
          "SAMEDEC_IS_NATIONAL"):
    if k not in env:
        bad.append(f"{k} is not set")
if bad:
    print("-ERR " + "; ".join(bad))
    sys.exit(1)
print("+OK")
