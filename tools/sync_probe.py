#!/usr/bin/env python3
"""What one synchronous call costs besides its kernel (rio_gp_debug_stream_probe modes 20..23): a one-thread kernel that
stores a sequence number into mapped pinned memory, waited for with hipStreamSynchronize or by spinning on the word, with
the request taken from the kernel arguments or read from mapped pinned memory.  Prints microseconds per call."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd")):
    sys.path.insert(0, p)
import rio_gp
g = rio_gp.LabPlacement(1 << 16, 64)
out = {}
for mode, name in ((20, "launch + hipStreamSynchronize, request in kernel arguments"),
                   (21, "launch + spin on the pinned word, request in kernel arguments"),
                   (22, "launch + spin on the pinned word, request read from mapped pinned memory"),
                   (23, "launch + hipStreamSynchronize, request read from mapped pinned memory")):
    out[name] = round(g.stream_probe(mode, 2000) * 1000.0, 2)
print(json.dumps({"us_per_call": out}, indent=1))
g.close()
