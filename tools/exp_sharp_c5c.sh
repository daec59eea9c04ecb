#!/bin/bash
# is the one-wave-per-item form bound by the workgroup dispatch rate?  persistent waves / several waves per workgroup
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
r() { NMX_OVERLAP=0 timeout 300 python tools/bench_configs.py C5 2>/dev/null | grep -E "windows_per_s|\"sharp\"" | tr -d '\n'; echo; }
echo "base: $(r)"
for w in 8 16 24 32; do echo "NMX_SW_PERSISTENT=$w: $(NMX_SW_PERSISTENT=$w r)"; done
for k in 2 4; do echo "generic kernel, NMX_WAVES_PER_WG=$k: $(NMX_SW_DENSE_FIRST=0 NMX_WAVES_PER_WG=$k r)"; done
