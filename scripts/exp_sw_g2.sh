#!/bin/bash
# GPU box: where the time of the L kernel group (R >= 26 and multi-tile shapes) goes - forward of the single-tile shapes,
# forward of the multi-tile shapes, reverse scan of the multi-tile queries - by leaving job classes out (MMGPU_SW_DEBUG_SKIP)
for skip in none multi single2 revmulti; do
  echo "== skip $skip"
  MMGPU_SW_DEBUG_SKIP=$skip python scripts/exp_sw_groups.py 2>&1 | grep -E "align kernels|done after"
done
