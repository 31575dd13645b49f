#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py: tools/prof_stats.sh <tag> [bench args...]  -> gpurun_out/prof_<tag>/ (+ bench line)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inflight2 "$@" > $out/bench.json 2> $out/bench.err
echo "prof $tag rc=$?"
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/kernel_stats.csv && head -25 $out/kernel_stats.csv
