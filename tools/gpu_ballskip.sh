#!/bin/bash
# Which contact code makes the slowest ball waves slow?  Stamp builds with parts of the contact sequence compiled out
# (PP_BALL_SKIP bits: 1 paddle, 2 link shapes, 4 table + net, 8 ground).  The physics of these builds is wrong on purpose.
set -o pipefail
for mask in 0 1 2 4 3 7; do
  echo "== PP_BALL_SKIP=$mask"
  PPENV_STAMP_DEFS="-DPP_BALL_SKIP=$mask" timeout -k 10 300 python tools/gpu_stamps.py 16384 TT 2>&1 | grep "span perc\|ball FK\|last ball substep" || exit 1
done
