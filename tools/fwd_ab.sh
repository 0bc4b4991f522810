#!/bin/bash
# Dev tool: forward/backward call time with the window kernel off / on (same box).
for m in 0 1; do echo "MMT_FWD_WIN=$m"; MMT_FWD_WIN=$m python tools/bwd_timing.py; done
