#!/bin/bash
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sfp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sfp -o sf -- python $GRAFT_REPO_ROOT/tools/probes/style_factors_time.py 2>/dev/null | grep "^N "
F=$(find /tmp/sfp -name "*kernel_stats.csv" | head -1); grep -i "style_factors" $F | awk -F'",' '{print $1}' | cut -c1-60 > /tmp/names; grep -i "style_factors" $F | awk -F'",' '{print $2}' | awk -F, '{print "calls " $1 "  avg ns " $3 "  min " $5 "  max " $6}' | paste /tmp/names -
