#!/bin/bash
# the API-side measurements of round 6: the reference's one-window call shape, Stream.run on 2 / 10 / 30 minutes
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python tools/bench_window.py 500 > $O/${TAG}_window_latency.json 2> $O/${TAG}_window_latency.err; cat $O/${TAG}_window_latency.err | cut -c1-400
for sec in 120 600 1800; do
  timeout 900 python tools/bench_stream.py --seconds $sec > $O/${TAG}_stream_${sec}s.json 2>$O/${TAG}_stream_${sec}s.err; cat $O/${TAG}_stream_${sec}s.json; echo
done
