#!/bin/bash
# which power-profile mode (if any) removes the slow start?  (root on the box; restored afterwards)
D=$(ls -d /sys/class/drm/card*/device | head -1)
echo "== pp_power_profile_mode"; cat $D/pp_power_profile_mode 2>&1 | head -30
echo "== power_dpm_force_performance_level: $(cat $D/power_dpm_force_performance_level 2>&1)"
R=${GRAFT_REPO_ROOT:-.}
run() { python $R/tools/warmup_idle.py 2>&1 | sed -n 2,3p; }
echo "== default"; run
for m in 5 4 1; do
  echo "manual" > $D/power_dpm_force_performance_level 2>/dev/null
  echo $m > $D/pp_power_profile_mode 2>/dev/null && echo "== profile mode $m: $(grep '\*' $D/pp_power_profile_mode | head -1)" && run
done
echo "auto" > $D/power_dpm_force_performance_level 2>/dev/null
for lvl in profile_peak profile_standard; do
  echo $lvl > $D/power_dpm_force_performance_level 2>/dev/null && echo "== level $lvl" && run
done
echo "auto" > $D/power_dpm_force_performance_level 2>/dev/null
rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|mclk|fclk|Power" | head
