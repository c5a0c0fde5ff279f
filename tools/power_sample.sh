#!/bin/bash
# Samples socket power, sclk and mclk (rocm-smi / sysfs, whichever the unprivileged user can read) every ~50 ms while a
# command runs:  tools/power_sample.sh <out.txt> <command...>      (GPU box; evidence for the power-limit statements)
out=$1; shift
rm -f /tmp/power_sample.stop
( i=0
  while [ ! -e /tmp/power_sample.stop ]; do
    ts=$(date +%s.%N)
    hw=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
    p=""; [ -n "$hw" ] && p=$(cat $hw/power1_average 2>/dev/null || cat $hw/power1_input 2>/dev/null)
    card=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
    sclk=$(grep '\*' $card/pp_dpm_sclk 2>/dev/null | tr -d '\n')
    mclk=$(grep '\*' $card/pp_dpm_mclk 2>/dev/null | tr -d '\n')
    echo "$ts power_uW=$p sclk=[$sclk] mclk=[$mclk]"
    i=$((i+1)); sleep 0.05
  done ) > $out.sysfs 2>&1 &
spid=$!
( while [ ! -e /tmp/power_sample.stop ]; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "GPU\[|Power|sclk|mclk|Temp" ; echo "--- $(date +%s.%N)"; sleep 0.2; done ) > $out.smi 2>&1 &
mpid=$!
"$@"
rc=$?
touch /tmp/power_sample.stop; sleep 0.4; kill $spid $mpid 2>/dev/null
echo "sysfs samples: $(wc -l < $out.sysfs), rocm-smi lines: $(wc -l < $out.smi)"
head -3 $out.sysfs; tail -2 $out.sysfs; head -12 $out.smi
exit $rc
