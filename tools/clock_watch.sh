#!/bin/bash
# usage: clock_watch.sh <label> <cmd...> : samples sclk/power with rocm-smi while cmd runs
label=$1; shift
"$@" > /tmp/cw_out.txt 2>&1 &
pid=$!
sleep 1.0
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed "s/^/[$label] /"; echo
  sleep 0.7
done
tail -3 /tmp/cw_out.txt
