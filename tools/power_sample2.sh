#!/bin/bash
# amd-smi / rocm-smi sampling during a long scan: tools/power_sample2.sh <out prefix> <command...>
out=$1; shift
rm -f /tmp/ps.stop
( while [ ! -e /tmp/ps.stop ]; do echo "--- $(date +%s.%N)"; amd-smi metric -p -c -u --json 2>/dev/null | tr -d '\n ' | head -c 3000; echo; sleep 0.1; done ) > $out.amdsmi 2>&1 &
a=$!
( while [ ! -e /tmp/ps.stop ]; do echo "--- $(date +%s.%N)"; rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | head -c 2000; echo; sleep 0.1; done ) > $out.rocmsmi 2>&1 &
b=$!
"$@"; rc=$?
touch /tmp/ps.stop; sleep 0.5; kill $a $b 2>/dev/null
exit $rc
