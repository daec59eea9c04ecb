#!/bin/bash
# Mode A with prebuilt variants of the library (tools/build_variants.sh -> py_neuromodulation_amd/libnmx_v<k>.so), same lease.
#   gpurun -- 'bash tools/exp_lib_variants.sh'       (the scratch copy's libnmx.so is overwritten variant by variant)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P=py_neuromodulation_amd
cp $P/libnmx.so $P/libnmx_keep.so
for v in $(ls $P/libnmx_v*.so | sort -V); do
  cp $v $P/libnmx.so
  echo "== $v"
  (cd /tmp && timeout 200 python $GRAFT_REPO_ROOT/tools/bench_scan.py --features fft 2>&1 | grep -v amdgpu | cut -c1-175)
  (cd /tmp && timeout 200 python $GRAFT_REPO_ROOT/tools/bench_scan.py 2>&1 | grep -v amdgpu | cut -c1-175)
done
cp $P/libnmx_keep.so $P/libnmx.so
