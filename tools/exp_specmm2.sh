#!/bin/bash
# What binds the matrix-pipe spectrum kernel: the same launch with every lane of a wave streaming ONE window (cache-friendly,
# wrong results) and / or half the MFMAs, against the real thing.  gpurun -- 'bash tools/exp_specmm2.sh'   (rebuilds libnmx on the box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp NMX_SPECMM=1
for flags in "" "-DNMX_SMM_DEBUG_SAMEWIN" "-DNMX_SMM_DEBUG_HALFK" "-DNMX_SMM_DEBUG_SAMEWIN -DNMX_SMM_DEBUG_HALFK"; do
  echo "flags: $flags"
  touch py_neuromodulation_amd/csrc/nmx_k_specmm.h
  NMX_EXTRA_CXXFLAGS="$flags" python -c "import __graft_entry__ as g; g.build_lib(force=True)" 2>&1 | tail -2
  (cd /tmp && timeout 200 python $GRAFT_REPO_ROOT/tools/bench_scan.py --features fft 2>&1 | grep -v amdgpu | cut -c1-160)
  (cd /tmp && timeout 200 python $GRAFT_REPO_ROOT/tools/bench_scan.py 2>&1 | grep -v amdgpu | cut -c1-160)
done
