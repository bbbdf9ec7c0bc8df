#!/bin/bash
# Interleaved A/B of library variants on one box (the clock a box holds drifts by ~1 % over a minute of load: variants run
# one after the other in blocks cannot be compared at that level).  Usage: bash tools/ab6.sh ROUNDS "bench_sorted args" v1 v2 ...
# (a variant name = ab/NAME.so through LD_PRELOAD; "product" = the in-tree library).  Prints the median over the rounds.
R=$(cd "$(dirname "$0")/.." && pwd)
ROUNDS=$1; ARGS=$2; shift 2
for v in "$@"; do : > /tmp/ab6_$v.txt; done
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    if [ $v = product ]; then python3 $R/tools/bench_sorted.py --layouts 40 --reps 3 $ARGS; else LD_PRELOAD=$R/ab/$v.so python3 $R/tools/bench_sorted.py --layouts 40 --reps 3 $ARGS; fi | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['ms'])" >> /tmp/ab6_$v.txt
  done
done
for v in "$@"; do python3 -c "
import sys, statistics
x = [float(l) for l in open('/tmp/ab6_$v.txt')]
print('%-12s median %.3f  min %.3f  max %.3f  (%d rounds) %s' % ('$v', statistics.median(x), min(x), max(x), len(x), '$ARGS'))"; done
