#!/bin/bash
# usage: bash scripts/sweep_seg.sh "seg|rays" ...   -- rank-share timing (N=1,2,4,8) under different segmentation settings
for cfg in "$@"; do
  seg=${cfg%%|*}; rays=${cfg#*|}
  echo "== RT_WF_SEG=$seg RT_WF_SEG_RAYS=$rays"
  RT_WF_SEG=$seg RT_WF_SEG_RAYS=$rays timeout -k 10 300 python scripts/rank_share.py ${WORKLOAD:-lambert_1m} ${NS:-1 2 4 8} 2>&1 | grep "^N=" | sed 's/rank 0 renders its share in //; s/(ideal.*stages/stages/'
done
