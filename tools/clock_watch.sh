#!/bin/bash
# Samples GPU clock / power while a command runs (DVFS investigation).  Usage: tools/clock_watch.sh <label> <cmd...>
LABEL=$1; shift
"$@" > /tmp/cw_$LABEL.out 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo
  sleep 1
done
wait $PID
echo "== $LABEL"; grep "X=" /tmp/cw_$LABEL.out
