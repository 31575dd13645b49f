#!/bin/bash
# Same box, alternating processes over SEVERAL values: tools/env_ab_multi2.sh VAR "v0 v1 v2" [encoder] [reps]
var=$1; vals=$2; enc=${3:-vitl}; reps=${4:-2}
for rep in $(seq 1 $reps); do
  for v in $vals; do
    env $var=$v python3 tools/forward_once.py "$enc" "$var=$v"
  done
done
