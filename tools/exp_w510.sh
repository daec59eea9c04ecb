#!/bin/bash
# 510-point time / oscillatory kernel (BASELINE config 5): plan re-read per phase, branch-free fill, 10 KB of LDS; 3 vs 4 waves per SIMD
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
r() { timeout 300 python tools/bench_c5_split.py 2>/dev/null | grep -E "timeosc_ms" | tr -d '\n'; echo; NMX_OVERLAP=0 timeout 300 python tools/bench_configs.py C5 2>/dev/null | grep -E "windows_per_s|\"timeosc\": [0-9]" | tr -d '\n'; echo; timeout 300 python tools/bench_configs.py C5 2>/dev/null | grep -E "windows_per_s" ; }
echo "prebuilt:"; r
timeout 600 python -m pytest tests -m gpu -q -x -k "config5 or c5 or short_windows or random_settings_highrate" 2>&1 | tail -2
export NMX_EXTRA_CXXFLAGS="-DNMX_W510_WAVES=4"
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_build.log 2>&1 || tail -5 gpurun_out/exp_build.log
echo "$NMX_EXTRA_CXXFLAGS:"; r
