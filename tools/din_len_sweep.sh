#!/bin/bash
# time of the (unstamped) DIN kernel vs a fixed history length: how much of a sample's cost is per-sample overhead
for l in 1 16 17 32 33 48 50; do DIN_PROBE_LEN=$l ./tools/din_probe_plain | tail -1 | sed "s/^/len $l: /"; done
