#!/bin/bash
# round 5: config-3 kernel experiments (TUNING build of libvsgpu.so must be in the tree): parity of the new variants against the
# 16x16x64 filter, then kernel times / phase stamps at 5 M and 20 M rows.  lowp_x32 = VAR + 1.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_c3
mkdir -p $O
cd $R
V="$1"; [ -z "$V" ] && V="32770,49158,262145,262149,393217"
timeout 900 python tools/check_x32.py --vars $V > $O/check.txt 2>&1; echo "check rc=$?" >> $O/rc.txt
OPTS=$(echo $V | sed 's/\([0-9]*\)/lowp_x32=\1/g')
timeout 900 python tools/time_lowp_kernels.py --rows 5000000 --opts $OPTS > $O/time_5m.txt 2> $O/time_5m.err
timeout 900 python tools/time_lowp_kernels.py --rows 20000000 --reps 5 --opts $OPTS > $O/time_20m.txt 2> $O/time_20m.err
# diagnosis: stamps and eliminations (replies meaningless)
D="$2"; [ -z "$D" ] && D="33282"
DOPTS=$(echo $D | sed 's/\([0-9]*\)/lowp_x32=\1/g')
timeout 900 python tools/time_lowp_kernels.py --rows 5000000 --opts $DOPTS > $O/diag_5m.txt 2> $O/diag_5m.err
cat $O/rc.txt; tail -3 $O/check.txt; cat $O/time_5m.txt $O/time_20m.txt $O/diag_5m.txt; grep -v "^$" $O/diag_5m.err | tail -60
