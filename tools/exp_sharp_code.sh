#!/bin/bash
# sharp-wave kernel: one-slot specialisation on / off (code size vs instructions), serial schedule, one lease
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
c() { NMX_OVERLAP=0 timeout 300 python tools/bench_configs.py C4 C5 2>/dev/null | grep -E "windows_per_s|\"sharp\": [0-9]" | tr -d '\n'; echo; }
echo "committed build (two waves per workgroup): $(c)"
echo "   three: $(NMX_WAVES_PER_WG=3 c)"
echo "   one:   $(NMX_WAVES_PER_WG=1 c)"
export NMX_EXTRA_CXXFLAGS="-DNMX_SW_NO_ONE_SLOT"
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_build.log 2>&1 || tail -5 gpurun_out/exp_build.log
echo "two-slot code only: $(c)"
echo "   one wave per workgroup:   $(NMX_WAVES_PER_WG=1 c)"
