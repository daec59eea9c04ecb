#!/bin/bash
# Host-memory batches (what Stream.run hands over): hops per copy / compute chunk.  gpurun -- 'bash tools/exp_host_chunk.sh'
cd /tmp; export TMPDIR=/tmp
for c in 128 256 384 512 768 1200; do
  echo "NMX_HOST_CHUNK_WINDOWS=$c"
  NMX_HOST_CHUNK_WINDOWS=$c python $GRAFT_REPO_ROOT/tools/profile_stream.py 2>&1 | grep -E "pinned float32 in|Stream.run"
done
