#!/bin/bash
# W = 8 loopback share (rank 0's part of 256^3 / 8, RCCL self-exchange), all schedules: fused unpack (default) against SB_NO_FUSED_UNPACK=1
echo "== fused unpack (default)"; python tools/lb_w8_timing.py 100 8
echo "== SB_NO_FUSED_UNPACK=1"; SB_NO_FUSED_UNPACK=1 python tools/lb_w8_timing.py 100 8
