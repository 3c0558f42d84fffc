#!/usr/bin/env python3
"""One line per bench.py JSON on stdin: tag, ms per step, verified, phases of the last step (tools/experiments/r6aj.sh)."""
import json
import sys
d = json.loads(sys.stdin.readline())
print(sys.argv[1], d["ms_per_step"], d.get("check", {}).get("verified"), dict(d.get("phase_ms_last_step", {})))
