cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 0 2; do
  if [ $k = 0 ]; then unset NMX_WAVES_PER_WG; else export NMX_WAVES_PER_WG=$k; fi
  echo "K=$k: $(timeout 300 python tools/bench_configs.py C3 2>/dev/null | tr -d '\n' | cut -c1-420)"
  echo "K=$k serial: $(NMX_OVERLAP=0 timeout 300 python tools/bench_configs.py C3 2>/dev/null | tr -d '\n' | cut -c1-420)"
done
