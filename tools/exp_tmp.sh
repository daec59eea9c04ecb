cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "normal or pipeline or stream or power or window_by_window" 2>&1 | tail -2
python tools/bench_stream.py 2>/dev/null | cut -c1-700
python tools/bench_norm.py 2>/dev/null | tail -5 | cut -c1-300
